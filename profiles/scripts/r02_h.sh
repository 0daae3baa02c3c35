#!/bin/bash
# round 2, GPU session H: direct checkpoint (no save/restore kernels) + register-pressure variants of the gradient kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h
make -C oracle -s
bash tools/ab3.sh $O E=libsoftmac_hip.so Ecopy=libsoftmac_hip.so,SMAC_CK_COPY=1 A=libsoftmac_hip_vA.so B=libsoftmac_hip_vB.so C=libsoftmac_hip_vC.so D=libsoftmac_hip_vD.so 2>&1 | tail -8
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_env.py tests/test_slabs.py tests/test_gpu_long_rollout.py tests/test_gpu_pour.py tests/test_losses.py -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-300
