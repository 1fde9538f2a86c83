set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kkt_fold.py tests/test_gpu_solver.py tests/test_gpu_palm.py tests/test_gpu_accadmm.py tests/test_gpu_slab_stress.py tests/test_gpu_multidevice.py -x -q -k "not unfused" > gpurun_out/r02_t5.log 2>&1; tail -3 gpurun_out/r02_t5.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_d20.json 2> gpurun_out/r02_d.err
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02_d200.json 2>> gpurun_out/r02_d.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof_d -o r02d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_prof_d.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02_prof_d.err
cd $GRAFT_REPO_ROOT
python - <<PY
import json
for f in ("r02_d20","r02_d200"):
    d=json.load(open("gpurun_out/%s.json"%f)); print(f, round(d["value"],2), d["config"]["kkt_checks_in_timed_region"], d["kernel_ms"])
PY
python tools/prof_summary.py gpurun_out/r02_prof_d/r02d_results.db | sed 's/(dotsocp::Grid.*)"/"/; s/(double.*)"/"/' | cut -c1-140 | head -24
