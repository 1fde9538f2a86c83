#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/p7
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_palm.py tests/test_gpu_solver.py::test_messages_on_second_streams_change_nothing tests/test_gpu_multiprocess.py tests/test_analytic_gaussian.py -x -q -m gpu -s > $O/suite.log 2>&1
tail -8 $O/suite.log; grep "analytic 2-D" $O/suite.log
python bench.py --no-cpu-baseline --method PALM --steps 100 > $O/palm.json 2>&1
python bench.py --no-cpu-baseline --method PALM --nslabs 2 --steps 100 > $O/palm_ns2.json 2>&1
python bench.py --no-cpu-baseline --rank-share 8 > $O/share8.json 2>$O/share8.err
tail -c 600 $O/share8.json
