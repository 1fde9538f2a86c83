#!/bin/bash
# round-4 probe 1: where a rank's iteration at N = 8 spends its time (timeline), against the single slab of the same size
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/p1
mkdir -p $O
python bench.py --no-cpu-baseline --steps 100 --warmup 20 > $O/full.json
python bench.py --no-cpu-baseline --rank-share 8 > $O/share8.json
python bench.py --no-cpu-baseline --grid 1024 1024 16 > $O/single16.json
DOTSOCP_OVERLAP=0 python bench.py --no-cpu-baseline --rank-share 8 > $O/share8_nooverlap.json
DOTSOCP_SPLIT_CONE=0 python bench.py --no-cpu-baseline --rank-share 8 > $O/share8_nosplit.json
python bench.py --no-cpu-baseline --nslabs 8 --steps 100 > $O/nslabs8.json
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --rank-share 8 --steps 40 --warmup 10 > $GRAFT_REPO_ROOT/$O/share8_traced.json
cd $GRAFT_REPO_ROOT
find $O/trace -name '*.csv' | head
