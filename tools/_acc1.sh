cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_accadmm.py tests/test_multilevel.py -x -q -m gpu > gpurun_out/acc1.log 2>&1; tail -5 gpurun_out/acc1.log
for v in 1 0 1; do DOTSOCP_KKT_FOLD=$v timeout -k 10 300 python bench.py --no-cpu-baseline --method acc-ADMM 2>/dev/null | python tools/benchline.py acc_kfold$v; done
