#!/bin/bash
# round 3, GPU session AP: do the two direct-checkpoint experiment modes (SMAC_CK_MODE = 1, 2; round 2, kept selectable) still pass parity after this round's changes?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ap; mkdir -p $O
for m in 1 2; do
  SMAC_CK_MODE=$m timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_pour.py -q -m gpu > $O/tests_mode$m.log 2>&1; echo "SMAC_CK_MODE=$m rc $?"; tail -6 $O/tests_mode$m.log | cut -c1-250
done
