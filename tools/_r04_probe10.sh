#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/p10
mkdir -p $O
for W in 0 1 2; do
DOTSOCP_TS_WIDE=$W timeout -k 10 300 python -m pytest tests/test_gpu_operators.py -x -q -m gpu -k "poisson or dctn" > $O/ops_$W.log 2>&1; echo "wide=$W ops: $(tail -1 $O/ops_$W.log)"
done
DOTSOCP_TSOLVE=dct AB_REPEATS=3 python tools/ab.py DOTSOCP_TS_WIDE 0 > $O/ab_1024_dct.txt 2>&1
AB_REPEATS=3 python tools/ab.py DOTSOCP_TS_WIDE 0 1 2 > $O/ab_1024.txt 2>&1
DOTSOCP_TSOLVE=dct AB_REPEATS=3 python tools/ab.py DOTSOCP_TS_WIDE 0 -- --grid 1025 1025 129 > $O/ab_1025_dct.txt 2>&1
AB_REPEATS=3 python tools/ab.py DOTSOCP_TS_WIDE 0 1 2 -- --grid 1025 1025 129 > $O/ab_1025.txt 2>&1
AB_REPEATS=3 python tools/ab.py DOTSOCP_TS_WIDE 0 1 -- --grid 512 512 64 > $O/ab_512.txt 2>&1
cat $O/ab_*.txt
