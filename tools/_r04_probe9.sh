#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/p9
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_operators.py tests/test_gpu_solver.py -x -q -m gpu > $O/suite.log 2>&1
tail -6 $O/suite.log
AB_REPEATS=3 python tools/ab.py DOTSOCP_TSOLVE tridiag dct > $O/ab_1024.txt 2>&1
AB_REPEATS=3 python tools/ab.py DOTSOCP_TSOLVE tridiag dct -- --grid 1025 1025 129 > $O/ab_1025.txt 2>&1
AB_REPEATS=3 python tools/ab.py DOTSOCP_TSOLVE tridiag dct -- --grid 257 257 65 > $O/ab_257.txt 2>&1
AB_REPEATS=3 python tools/ab.py DOTSOCP_TSOLVE tridiag dct -- --grid 256 256 64 > $O/ab_256.txt 2>&1
cat $O/ab_*.txt
python tools/parity_fullsize.py 4 > $O/parity.log 2>&1; tail -9 $O/parity.log
