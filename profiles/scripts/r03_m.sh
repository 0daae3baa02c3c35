#!/bin/bash
# round 3, GPU session M: two-pass p2g.grad without the LDS stash (SMAC_P2GG_V2) - parity of that build, then an A/B against the shipped kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
SMAC_LIB=$PWD/softmac_amd/lib/libsoftmac_hip_v2.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_backward.py -x -q -m gpu > $O/parity_v2.log 2>&1; echo "parity rc $?"; tail -4 $O/parity_v2.log | cut -c1-300
bash tools/ab3.sh $O base=libsoftmac_hip.so v2=libsoftmac_hip_v2.so > $O/ab.txt 2>&1; cat $O/ab.txt
python3 - <<'PY'
import json,glob
for lab in ("base","v2"):
    for f in sorted(glob.glob(f"gpurun_out/r03m/ab_{lab}_*.json")):
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(lab, round(d['value'],1), d['ms_per_step_all'], {k:d['kernels_ms'].get(k) for k in ('p2g_g2p_grad','p2g_grad','g2p_grad')})
PY
