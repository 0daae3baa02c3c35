#!/bin/bash
# round 3, GPU session AV: windowed env episodes (WindowedEnvEpisode: TaichiEnv with velocity-controlled primitives in windows of env steps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03av; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_windowed.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; grep "^E \|passed\|failed" $O/tests.log | cut -c1-400 | head -20
