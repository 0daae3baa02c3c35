#!/bin/bash
# round 3, GPU session AU: the C4 slice test with its new batched-sweep leg (4M particles, 256^3: fused step, restore-ahead, save in k_g2p at that size)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03au; mkdir -p $O
make -C oracle -s
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "c4_slice" > $O/tests.log 2>&1; echo "tests rc $?"; tail -5 $O/tests.log | cut -c1-300
