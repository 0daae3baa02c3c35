#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ay; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -m gpu -k "rebinning or overflow" --durations=6 -s > $O/tests.log 2>&1; echo "rc $?"; grep "float32\] state\|passed\|failed\|s call" $O/tests.log | cut -c1-300
