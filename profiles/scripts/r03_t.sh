#!/bin/bash
# round 3, GPU session T: restore-ahead (k_reduce_grid_grad_ahead, two grid buffer sets; SMAC_RESTORE_AHEAD, default 1) - parity of the sweeps, A/B on one library, the whole suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_backward.py -x -q -m gpu > $O/parity.log 2>&1; rc=$?; echo "parity rc $rc"; tail -4 $O/parity.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
bash tools/ab3.sh $O off=libsoftmac_hip.so,SMAC_RESTORE_AHEAD=0 ahead=libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_fused_backward.py > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -4 $O/gpu_suite.log | cut -c1-300
