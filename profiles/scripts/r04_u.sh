#!/bin/bash
# round 4, session U: drift repair with the sheet in contact (the host's cloth calls of every frame on file and made again), cloth and drift suites around it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_cloth.py tests/test_gpu_mixed.py tests/test_gpu_parity.py -x -q -k "outrun or drift or cloth or mixed or env_loop or substep_with" > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -25 $O/pytest.log
exit $rc
