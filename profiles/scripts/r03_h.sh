#!/bin/bash
# round 3, GPU session H: rebinning-invariance test with MEASURED branch flips, slab-library tests incl. the cloth variant, the C5 slice, host / GPU
# cost of the slab loops with RCCL on the kernels' stream vs on its own stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "rebinning" > $O/fullsize.log 2>&1; echo "fullsize rc $?"
grep -E "^\.?\[|passed|failed" $O/fullsize.log | cut -c1-900
timeout -k 10 600 python3 -m pytest tests/test_gpu_slab_lib.py -x -q -m gpu > $O/slab_lib.log 2>&1; echo "slab_lib rc $?"; tail -15 $O/slab_lib.log | cut -c1-400
timeout -k 10 900 python3 -m pytest tests/test_gpu_cloth.py -x -q -m gpu -s -k "c5 or full_size" > $O/cloth.log 2>&1; echo "cloth rc $?"
grep -E "^\[C5|passed|failed|Error" $O/cloth.log | cut -c1-600
timeout -k 10 900 python3 tools/exchange_overhead.py c stub rccl > $O/exchange_overhead.txt 2>&1 || { tail -20 $O/exchange_overhead.txt; exit 1; }
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/exchange_overhead.txt
