#!/bin/bash
# round 3, GPU session N: sdf + normal taps fetched together in collide_mixed (prim_sdf_normal) - contact parity on that build, then the A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_pour.py tests/test_gpu_env.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc $?"; tail -4 $O/parity.log | cut -c1-300
bash tools/ab3.sh $O notaps=libsoftmac_hip_notaps.so taps=libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
