#!/bin/bash
# the slab suites under random stream stalls and between guard bands, then the whole GPU suite
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/s1
mkdir -p $O
SLAB="tests/test_gpu_solver.py tests/test_gpu_multidevice.py tests/test_gpu_multiprocess.py tests/test_gpu_slab_stress.py tests/test_gpu_config4.py tests/test_gpu_palm.py tests/test_gpu_accadmm.py tests/test_gpu_fullsize.py"
DOTSOCP_STRESS_STREAMS=1 timeout -k 10 1100 python -m pytest $SLAB -x -q -m gpu > $O/stress.log 2>&1
echo "stress: $(tail -1 $O/stress.log)"
DOTSOCP_CANARY=1 timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/canary.log 2>&1
echo "canary: $(tail -1 $O/canary.log)"
