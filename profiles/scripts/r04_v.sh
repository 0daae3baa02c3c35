#!/bin/bash
# round 4, session V: windowed episodes - loss tape, particle actions, the earlier cases
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04v; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_windowed.py tests/test_losses.py -x -q -m gpu > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -25 $O/pytest.log
exit $rc
