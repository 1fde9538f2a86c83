#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/p11
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_operators.py -x -q -m gpu -k "poisson or dctn" > $O/ops.log 2>&1; echo "ops: $(tail -1 $O/ops.log)"
DOTSOCP_TS_POW2=1 timeout -k 10 300 python -m pytest tests/test_gpu_operators.py tests/test_gpu_config4.py -x -q -m gpu > $O/ops2.log 2>&1; echo "ops pow2: $(tail -1 $O/ops2.log)"
DOTSOCP_TS_POW2=1 AB_REPEATS=3 python tools/ab.py DOTSOCP_TS_PIPE 1 0 > $O/ab_1024.txt 2>&1
AB_REPEATS=3 python tools/ab.py DOTSOCP_TS_POW2 0 > $O/ab_1024_dct.txt 2>&1
AB_REPEATS=3 python tools/ab.py DOTSOCP_TS_PIPE 1 0 -- --grid 1025 1025 129 > $O/ab_1025.txt 2>&1
AB_REPEATS=3 python tools/ab.py DOTSOCP_TS_PIPE 1 0 -- --grid 513 513 129 > $O/ab_513.txt 2>&1
cat $O/ab_*.txt
python tools/parity_fullsize.py 4 1025 129 > $O/parity1025.log 2>&1; tail -8 $O/parity1025.log
