#!/bin/bash
# round 2, GPU session L: fused backward grid pass (k_reduce_grid_grad + direct contact adjoint) vs the three-kernel sequence; GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
make -C oracle -s
bash tools/ab3.sh $O fused=libsoftmac_hip.so unfused=libsoftmac_hip.so,SMAC_FUSED_GRID_BWD=0 2>&1 | tail -3
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-300
