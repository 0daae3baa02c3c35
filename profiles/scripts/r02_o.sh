#!/bin/bash
# round 2, GPU session O: cloth contact tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
make -C oracle -s
timeout -k 10 900 python -m pytest tests/test_gpu_cloth.py -m gpu -q > $O/pytest_cloth.log 2>&1; tail -12 $O/pytest_cloth.log | cut -c1-400
timeout -k 10 300 python tools/cloth_debug.py > $O/debug.log 2>&1; tail -10 $O/debug.log | cut -c1-300
