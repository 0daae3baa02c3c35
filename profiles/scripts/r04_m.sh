#!/bin/bash
# round 4, session M: empty filed hit list -> no contact adjoint launch; suites around it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_backward.py tests/test_gpu_env.py -x -q > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -15 $O/pytest.log
exit $rc
