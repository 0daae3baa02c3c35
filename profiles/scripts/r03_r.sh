#!/bin/bash
# round 3, GPU session R: wave reductions on the DPP crossbar / v_readlane instead of ds_bpermute (tile_scale, the contact chains) - parity, then A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r; mkdir -p $O
bash tools/ab3.sh $O base=libsoftmac_hip_base.so dpp=libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cloth.py tests/test_gpu_env.py tests/test_gpu_pour.py tests/test_gpu_fused_backward.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc $?"; tail -4 $O/parity.log | cut -c1-300
