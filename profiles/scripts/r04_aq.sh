#!/bin/bash
# round 4, session AQ: k_grid_op's halo packing as an instantiation of its own (the single-GPU kernel carries none of it) - slab tests, a parity subset, then the end-of-round pipeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04aq; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_slab_lib.py tests/test_slabs.py tests/test_gpu_parity.py -m gpu -x -q -k "slab or rank or exchange or migration or fused or batched or grip_fixture or stub or reductions or cloth_variant" > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $O/pytest.log | head -30; exit $rc; }
bash profiles/scripts/r04_af.sh
