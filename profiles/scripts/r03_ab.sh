#!/bin/bash
# round 3, GPU session AB: windowed tests on the per-trajectory primitive calls, then the cost of a windowed episode at the benchmark size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_windowed.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
timeout -k 10 500 python3 tools/windowed_cost.py 200 50 > $O/windowed_cost.txt 2> $O/err.txt; echo "rc $?"; cat $O/windowed_cost.txt
