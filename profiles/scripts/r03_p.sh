#!/bin/bash
# round 3, GPU session P: p2g.grad's 27-node gather chosen per wave (LDS-only copy: ds_read_b128 instead of flat_load_dwordx4) - parity, then A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_backward.py tests/test_gpu_fuzz.py tests/test_gpu_pour.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc $?"; tail -4 $O/parity.log | cut -c1-300
bash tools/ab3.sh $O base=libsoftmac_hip_base.so new=libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
python3 - <<'PY'
import json,glob
for lab in ("base","new"):
    for f in sorted(glob.glob(f"gpurun_out/r03p/ab_{lab}_*.json")):
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels_ms']
        print(lab, round(d['value'],1), {n: round(k[n],4) for n in ('p2g_g2p_grad','p2g_grad','g2p_grad') if n in k})
PY
