#!/bin/bash
# round 3, GPU session AG: the compiler's AMDGPU scheduling strategies (-mllvm -amdgpu-sched-strategy=...) on the whole library - A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ag; mkdir -p $O
bash tools/ab3.sh $O default=libsoftmac_hip.so maxilp=libsoftmac_hip_max-ilp.so memclause=libsoftmac_hip_max-memory-clause.so minreg=libsoftmac_hip_iterative-minreg.so > $O/ab.txt 2>&1; cat $O/ab.txt
python3 - <<'PY'
import json,glob
for lab in ("default","maxilp","memclause","minreg"):
    for f in sorted(glob.glob(f"gpurun_out/r03ag/ab_{lab}_*.json")):
        try:
            d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels_ms']
            print(lab, round(d['value'],1), {n: round(k[n]*1e3,1) for n in ('p2g','g2p','p2g_g2p_grad','contact','contact_grad','reduce_agvout','grid_op') if n in k})
        except Exception as e: print(lab,'FAILED',e)
PY
