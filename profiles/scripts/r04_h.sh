#!/bin/bash
# round 4, session H: the in-library slab loop and the device-side migration between TWO ranks on one GPU (IPC link, SMAC_COMM_STUB=2) + the slab suites
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
make -C oracle -s
timeout -k 10 900 python3 -m pytest tests/test_slabs.py -x -q -s -m gpu -k "two_ranks" > $O/pytest_ipc.log 2>&1
echo "pytest ipc rc $?"; grep -a "passed\|failed\|Error\|error\|two slabs\|migration" $O/pytest_ipc.log | tail -20
timeout -k 10 900 python3 -m pytest tests/test_slabs.py tests/test_gpu_slab_lib.py tests/test_gpu_env.py tests/test_gpu_windowed.py -x -q -m gpu > $O/pytest_slabs.log 2>&1
echo "pytest slabs rc $?"; tail -4 $O/pytest_slabs.log
timeout -k 10 300 python3 tools/launch_overhead.py > $O/launch_overhead.txt 2>&1; cat $O/launch_overhead.txt | tail -4
