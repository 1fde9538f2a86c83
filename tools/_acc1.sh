cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_accadmm.py -x -q -m gpu > gpurun_out/acc1.log 2>&1; tail -5 gpurun_out/acc1.log
for v in 1 0; do DOTSOCP_ACC_POST=$v timeout -k 10 300 python bench.py --no-cpu-baseline --method acc-ADMM 2>/dev/null | python tools/benchline.py acc_post$v; done
