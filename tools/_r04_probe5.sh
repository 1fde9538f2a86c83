#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/p5
mkdir -p $O
python bench.py --no-cpu-baseline --steps 100 --warmup 20 > $O/full.json 2>$O/full.err || exit 1
AB_REPEATS=3 python tools/ab.py DOTSOCP_PITCH2 1 0 -- --rank-share 8 --steps 200 --warmup 20 > $O/ab_pitch2.txt 2>&1
python bench.py --no-cpu-baseline --nslabs 8 --steps 100 > $O/nslabs8.json 2>&1
python bench.py --no-cpu-baseline --nslabs 2 --steps 100 > $O/nslabs2.json 2>&1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1
tail -5 $O/suite.log
