#!/bin/bash
# round 2, GPU session AM: the fused-backward test with its long multi-epoch case
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02am; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_fused_backward.py -m gpu -q -x -s --durations=5 > $O/pytest_fused.log 2>&1; tail -12 $O/pytest_fused.log | cut -c1-300
