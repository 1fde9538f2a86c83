#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/p12
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_multiprocess.py tests/test_gpu_solver.py::test_rccl_communicator_world1 -x -q -m gpu > $O/suite.log 2>&1
tail -3 $O/suite.log
python bench.py --no-cpu-baseline --rank-share 8 > $O/share8.json 2>$O/share8.err; tail -c 400 $O/share8.json
