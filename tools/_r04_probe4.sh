#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/p4
mkdir -p $O
python bench.py --no-cpu-baseline --steps 100 --warmup 20 > $O/full.json 2>$O/full.err || exit 1
python bench.py --no-cpu-baseline --rank-share 8 > $O/share8.json 2>$O/share8.err || exit 1
AB_REPEATS=3 python tools/ab.py DOTSOCP_OVERLAP 1 0 -- --rank-share 8 --steps 200 --warmup 20 > $O/ab_overlap.txt 2>&1
python bench.py --no-cpu-baseline --nslabs 8 --steps 100 > $O/nslabs8.json 2>&1
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --rank-share 8 --steps 40 --warmup 10 > $GRAFT_REPO_ROOT/$O/share8_traced.json 2>/dev/null)
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_multidevice.py tests/test_gpu_multiprocess.py tests/test_gpu_slab_stress.py -x -q -m gpu > $O/suite.log 2>&1
tail -5 $O/suite.log
