#!/bin/bash
# round 3, GPU session I: device-side migration, the cloth variant under the slab loop (per-frame checkpoint invalidation), the cloth suite on the
# checkpointed backward path, rebinning test, full suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_slab_lib.py -x -q -m gpu > $O/slab_lib.log 2>&1; echo "slab_lib rc $?"; tail -25 $O/slab_lib.log | cut -c1-500
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -25 $O/gpu_suite.log | cut -c1-400
