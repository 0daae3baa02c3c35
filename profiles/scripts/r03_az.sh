#!/bin/bash
# round 3, GPU session AZ: the whole GPU suite on the final tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03az; mkdir -p $O
make -C oracle -s
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu --durations=40 > $O/gpu_suite.log 2>&1; echo "suite rc $?"; grep -A45 "slowest" $O/gpu_suite.log | cut -c1-160; tail -2 $O/gpu_suite.log
