#!/bin/bash
# round 4, session P: windowed episode with particle actions; the in-library loop + migration among THREE ranks (IPC link)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_windowed.py tests/test_slabs.py -x -q -m gpu -s -k "particle_actions or three_ranks" > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -30 $O/pytest.log
exit $rc
