#!/bin/bash
# round 3, GPU session S: the checkpoint save inside k_g2p's launch (SMAC_SAVE_IN_G2P, default 1) - A/B on one library, then the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
bash tools/ab3.sh $O sep=libsoftmac_hip.so,SMAC_SAVE_IN_G2P=0 merged=libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout -k 10 1050 python3 -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -4 $O/gpu_suite.log | cut -c1-300
