#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/p6
mkdir -p $O
python bench.py --no-cpu-baseline --steps 100 --warmup 20 > $O/full.json 2>$O/full.err || exit 1
python bench.py --no-cpu-baseline --nslabs 8 --steps 100 > $O/nslabs8.json 2>&1
python bench.py --no-cpu-baseline --nslabs 2 --steps 100 > $O/nslabs2.json 2>&1
python bench.py --no-cpu-baseline --rank-share 8 > $O/share8.json 2>&1
timeout -k 10 1000 python -m pytest tests/test_gpu_solver.py tests/test_gpu_multidevice.py tests/test_gpu_slab_stress.py tests/test_gpu_palm.py tests/test_gpu_accadmm.py tests/test_gpu_config4.py tests/test_multilevel.py -x -q -m gpu > $O/suite.log 2>&1
tail -5 $O/suite.log
