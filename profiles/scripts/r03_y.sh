#!/bin/bash
# round 3, GPU session Y: the API-sequence test (round-3 launch structure against everything switched off), the re-sort tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_api_sequences.py tests/test_gpu_resort.py -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc $?"; grep "re-sort\|changed their cell\|passed\|failed\|FAILED\|Error" $O/tests.log | cut -c1-400
