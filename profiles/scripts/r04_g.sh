#!/bin/bash
# round 4, session G: the measured value behind every float32 bound that is not F32_TOL, collected through helpers.note
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O; rm -f $O/errs.txt
make -C oracle -s
SMAC_PRINT_ERRS=$PWD/$O/errs.txt timeout -k 10 900 python3 -m pytest tests/test_gpu_env.py tests/test_gpu_parity.py tests/test_gpu_windowed.py tests/test_gpu_slab_lib.py tests/test_gpu_pour.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log; wc -l $O/errs.txt
