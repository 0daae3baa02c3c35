#!/bin/bash
# round 3, GPU session AO: the whole GPU suite on the end-of-round tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ao; mkdir -p $O
make -C oracle -s
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -4 $O/gpu_suite.log | cut -c1-300
