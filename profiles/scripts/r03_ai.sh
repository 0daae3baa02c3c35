#!/bin/bash
# round 3, GPU session AI: slab reduction with all eight sources' loads in flight (k_grid_op, k_reduce_*) - A/B, then parity
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ai; mkdir -p $O
bash tools/ab3.sh $O base=libsoftmac_hip_base.so batched=libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
python3 - <<'PY'
import json,glob
for lab in ("base","batched"):
    for f in sorted(glob.glob(f"gpurun_out/r03ai/ab_{lab}_*.json")):
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels_ms']
        print(lab, round(d['value'],1), {n: round(k[n]*1e3,1) for n in ('grid_op','reduce_agvout','p2g','g2p','p2g_g2p_grad') if n in k})
PY
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_slab_lib.py tests/test_gpu_fused_backward.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log | cut -c1-300
