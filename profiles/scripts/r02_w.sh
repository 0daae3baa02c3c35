#!/bin/bash
# round 2, GPU session W: cloth tests (golden vectors, API misuse, penalty contact tolerance)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02w; mkdir -p $O
make -C oracle -s
timeout -k 10 600 python -m pytest tests/test_gpu_cloth.py -m gpu -q > $O/pytest_cloth.log 2>&1; tail -8 $O/pytest_cloth.log | cut -c1-300
