#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/s2
mkdir -p $O
SLAB="tests/test_gpu_solver.py tests/test_gpu_multidevice.py tests/test_gpu_multiprocess.py tests/test_gpu_slab_stress.py tests/test_gpu_config4.py tests/test_gpu_palm.py tests/test_gpu_accadmm.py tests/test_gpu_fullsize.py"
DOTSOCP_STRESS_STREAMS=1 timeout -k 10 1100 python -m pytest $SLAB -x -q -m gpu > $O/r04_slab_suites_under_stream_stalls.log 2>&1
echo "stress: $(tail -1 $O/r04_slab_suites_under_stream_stalls.log)"
