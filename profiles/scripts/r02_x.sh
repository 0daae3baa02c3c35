#!/bin/bash
# round 2, GPU session X: backward pass on the checkpoint itself (SMAC_CK_MODE=3: chunk descriptors carry the checkpoint slots, no restore kernel) vs default,
# 3 interleaved processes each; parity suites under that mode; cloth fuzz in both modes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02x; mkdir -p $O
make -C oracle -s
bash tools/ab3.sh $O base=libsoftmac_hip.so ck3=libsoftmac_hip.so,SMAC_CK_MODE=3 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests/test_gpu_cloth.py -m gpu -q -x -k random > $O/pytest_fuzz.log 2>&1; tail -3 $O/pytest_fuzz.log | cut -c1-300
SMAC_CK_MODE=3 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_env.py tests/test_gpu_pour.py tests/test_gpu_cloth.py tests/test_gpu_long_rollout.py -m gpu -q -x > $O/pytest_ck3.log 2>&1; tail -3 $O/pytest_ck3.log | cut -c1-300
