set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_kkt_fold.py tests/test_gpu_solver.py -x -q -k "not unfused" > gpurun_out/r02_t4.log 2>&1; tail -3 gpurun_out/r02_t4.log
for occ in 0 1; do
  DOTSOCP_KKT_OCC=$occ timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_c20_occ$occ.json 2> gpurun_out/r02_c.err
done
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02_c200.json 2>> gpurun_out/r02_c.err
cd /tmp && export TMPDIR=/tmp
DOTSOCP_KKT_OCC=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof_b -o r02b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_prof_b.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02_prof_b.err
DOTSOCP_KKT_OCC=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof_c -o r02c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_prof_c.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02_prof_c.err
cd $GRAFT_REPO_ROOT
python - <<PY
import json
for f in ("r02_c20_occ0","r02_c20_occ1","r02_c200"):
    d=json.load(open("gpurun_out/%s.json"%f)); print(f, round(d["value"],2), d["config"]["kkt_checks_in_timed_region"], d["kernel_ms"])
PY
python tools/prof_summary.py gpurun_out/r02_prof_b/r02b_results.db | cut -c1-150 | head -24
python tools/prof_summary.py gpurun_out/r02_prof_c/r02c_results.db | grep "kkt_cells\|qstep_rhs" | cut -c1-150
