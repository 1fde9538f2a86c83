#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/p8
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_multidevice.py tests/test_gpu_multiprocess.py tests/test_gpu_slab_stress.py tests/test_gpu_config4.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/suite.log 2>&1
tail -4 $O/suite.log
python bench.py --no-cpu-baseline --rank-share 8 > $O/share8.json 2>$O/share8.err
python bench.py --no-cpu-baseline --rank-share 4 > $O/share4.json 2>$O/share4.err
python bench.py --no-cpu-baseline --rank-share 2 > $O/share2.json 2>$O/share2.err
python bench.py --no-cpu-baseline --nslabs 2 --steps 100 > $O/nslabs2.json 2>&1
python bench.py --no-cpu-baseline --nslabs 8 --steps 100 > $O/nslabs8.json 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/tr8 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --rank-share 8 --steps 40 --warmup 10 > $GRAFT_REPO_ROOT/$O/share8_traced.json 2>/dev/null)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/tr2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --rank-share 2 --steps 40 --warmup 10 > $GRAFT_REPO_ROOT/$O/share2_traced.json 2>/dev/null)
