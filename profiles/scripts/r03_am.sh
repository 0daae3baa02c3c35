#!/bin/bash
# round 3, GPU session AM: the backward grid pass skips the blocks flagged empty in the frame's checkpoint - A/B against the commit before, parity
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03am; mkdir -p $O
bash tools/ab3.sh $O base=libsoftmac_hip_base.so skipred=libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
python3 - <<'PY'
import json,glob
for lab in ("base","skipred"):
    for f in sorted(glob.glob(f"gpurun_out/r03am/ab_{lab}_*.json")):
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels_ms']
        print(lab, round(d['value'],1), d['ms_per_step_all'], {n: round(k[n]*1e3,1) for n in ('g2p','reduce_agvout','p2g_g2p_grad','contact_grad') if n in k})
PY
timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_backward.py tests/test_gpu_api_sequences.py tests/test_gpu_windowed.py tests/test_gpu_env.py tests/test_gpu_pour.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log | cut -c1-300
