#!/bin/bash
# round 4, session AN: after the last source cleanup (dead host state removed) - a parity subset, then the end-of-round pipeline (profiles/scripts/r04_af.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04an; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_forward.py tests/test_gpu_api_sequences.py tests/test_gpu_resort.py -x -q > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $O/pytest.log | head -30; exit $rc; }
bash profiles/scripts/r04_af.sh
