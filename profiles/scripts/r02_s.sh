#!/bin/bash
# round 2, GPU session S: checkpoint save after g2p (is the successor dispatch what slows k_g2p?) over 5 interleaved processes; measured f32 errors of the cloth path
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
make -C oracle -s
timeout -k 10 400 python -m pytest tests/test_gpu_cloth.py -m gpu -q -s -k "substep" 2>&1 | grep "^\[cloth" > $O/cloth_errors.txt; cat $O/cloth_errors.txt | cut -c1-200
bash tools/ab_runs.sh $O 5 base=libsoftmac_hip.so late=libsoftmac_hip.so,SMAC_SAVE_AFTER_G2P=1 2>&1 | tail -3
SMAC_SAVE_AFTER_G2P=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env.py -m gpu -q -x > $O/pytest_late.log 2>&1; tail -2 $O/pytest_late.log
