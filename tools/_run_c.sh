cd $GRAFT_REPO_ROOT
O=gpurun_out/xcd; mkdir -p $O
for q in 0 1; do for c in 0 1; do
  DOTSOCP_QXCD=$q DOTSOCP_XCD=$c timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/b_q${q}_c${c}.json 2>> $O/err.txt
done; done
cd /tmp && export TMPDIR=/tmp
DOTSOCP_QXCD=1 DOTSOCP_XCD=1 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>> $GRAFT_REPO_ROOT/$O/err.txt
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_fetch FETCH_SIZE | sed 's/(dotsocp::Grid[^"]*"/"/; s/(double[^"]*"/"/' | cut -c1-120 | head -8
rm -rf $O/pmc_fetch
python - <<PY
import json
for q in (0,1):
  for c in (0,1):
    d=json.load(open("gpurun_out/xcd/b_q%d_c%d.json"%(q,c))); k=d["kernel_ms"]
    print("QXCD",q,"XCD",c, round(d["value"],2), "cone", k["cone_fused_b"], "qstep", k["qstep"], "poisson", k["poisson"])
PY
