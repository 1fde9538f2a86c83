#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/s2
mkdir -p $O
timeout -k 10 560 python -m pytest tests -x -q -m gpu > $O/r04_gpu_suite.log 2>&1
echo "plain: $(tail -1 $O/r04_gpu_suite.log)"
DOTSOCP_CANARY=1 timeout -k 10 560 python -m pytest tests -x -q -m gpu > $O/r04_gpu_suite_under_guard_bands.log 2>&1
echo "canary: $(tail -1 $O/r04_gpu_suite_under_guard_bands.log)"
