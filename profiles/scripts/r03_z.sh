#!/bin/bash
# round 3, GPU session Z: windowed episodes (checkpoint-every-K state frames with recompute, softmac_amd/engine/windowed.py), the API-sequence test with its log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_windowed.py tests/test_gpu_api_sequences.py -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc $?"; grep "api sequence\|passed\|failed\|FAILED\|Error\|assert" $O/tests.log | cut -c1-600
