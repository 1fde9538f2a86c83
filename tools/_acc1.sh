cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_accadmm.py -x -q -m gpu > gpurun_out/acc1.log 2>&1; tail -5 gpurun_out/acc1.log
timeout -k 10 300 python bench.py --no-cpu-baseline --method acc-ADMM > gpurun_out/acc1_bench.json 2>gpurun_out/acc1_err.txt && python tools/benchline.py acc < gpurun_out/acc1_bench.json
