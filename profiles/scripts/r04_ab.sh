#!/bin/bash
# round 4, session AB: the whole GPU suite on the wide-tile library (default build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ab; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -15 $O/pytest.log
exit $rc
