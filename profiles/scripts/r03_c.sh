#!/bin/bash
# round 3, GPU session C: A/B of the backward particle kernels - LDS stash 34 -> 21 slots (e', J-1, cof(F_tmp) rebuilt from U, e, V) and
# particle rows fetched before the tile barrier - against the round's starting kernels; then parity + fused tests on the new build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
bash tools/ab3.sh $O base=libsoftmac_hip_base.so new=libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_backward.py tests/test_gpu_fullsize.py tests/test_gpu_cloth.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
