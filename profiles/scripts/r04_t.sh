#!/bin/bash
# round 4, session T: the whole GPU suite on the final sources, then the end-of-round measurement set (as r04_r.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
make -C oracle -s
timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -4 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
sed 's/r04r/r04t/g; s/session R/session T/' profiles/scripts/r04_r.sh > /tmp/r04_t_measure.sh
bash /tmp/r04_t_measure.sh
