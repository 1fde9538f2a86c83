cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_palm.py tests/test_gpu_kkt_fold.py -x -q -m gpu > gpurun_out/palm.log 2>&1; tail -15 gpurun_out/palm.log
for v in 1 0 1 0; do DOTSOCP_PALM_FAST=$v timeout -k 10 300 python bench.py --no-cpu-baseline --method PALM 2>/dev/null | python tools/benchline.py palm_fast$v; done
