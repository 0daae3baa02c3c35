#!/bin/bash
# round 3, GPU session J: the whole GPU suite on the current tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
timeout -k 10 1150 python3 -m pytest tests -q -m gpu --durations=10 > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -30 $O/gpu_suite.log | cut -c1-300
