#!/bin/bash
# round 3, GPU session L: the RCCL loader change (one instance per process), the cloth env mirror's additions, the 2-rank launcher test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_slab_lib.py tests/test_bench_launch.py tests/test_gpu_cloth.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc $?"; tail -12 $O/tests.log | cut -c1-400
grep -h "Librccl path" $O/tests.log | sort | uniq -c
