# rank-share measurements: one rank's slab of the N-way split on one GPU (loopback messages), plus the full grid
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/share
mkdir -p $O
echo skip tests
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
$B > $O/full.json 2> $O/err.txt
for n in 2 4 8; do $B --rank-share $n > $O/share$n.json 2>> $O/err.txt; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof8 -o s8 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --rank-share 8 > $O/share8_under_rocprof.json 2>> $O/err.txt
cd $GRAFT_REPO_ROOT
for f in full share2 share4 share8 share8_under_rocprof; do python - <<PY
import json
d=json.load(open("$O/$f.json"))
print("$f", round(d["ms_per_step"],4), d.get("rank_share"), {k:v for k,v in d["kernel_ms"].items() if v})
PY
done
tail -5 $O/err.txt
find $O/prof8 -name "*kernel_stats.csv"
