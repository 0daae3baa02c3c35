#!/bin/bash
# round 4, session D: C5 as written - mixed rigid + sheet handle, two materials: new parity tests, then the cloth and contact suites they touch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
make -C oracle -s
timeout -k 10 900 python3 -m pytest tests/test_gpu_mixed.py -x -q -s -m gpu > $O/pytest_mixed.log 2>&1
echo "pytest mixed rc $?"; grep -a "mixed\|passed\|failed\|Error\|assert" $O/pytest_mixed.log | tail -20
timeout -k 10 900 python3 -m pytest tests/test_gpu_cloth.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_cloth.log 2>&1
echo "pytest cloth+parity rc $?"; tail -3 $O/pytest_cloth.log
