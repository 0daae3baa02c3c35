#!/bin/bash
# round 3, GPU session AT: k_contact_hits' corrections straight to grid_v_out with global atomics instead of the per-workgroup LDS tile (phase clock: the tile,
# its flush and three barriers are 36 % of a contact wave's life) - A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03at; mkdir -p $O
bash tools/ab3.sh $O tile=libsoftmac_hip.so direct=libsoftmac_hip_cdirect.so both=libsoftmac_hip_cdirect2.so > $O/ab.txt 2>&1; cat $O/ab.txt
python3 - <<'PY'
import json,glob
for lab in ("tile","direct","both"):
    for f in sorted(glob.glob(f"gpurun_out/r03at/ab_{lab}_*.json")):
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels_ms']
        print(lab, round(d['value'],1), {n: round(k[n]*1e3,1) for n in ('contact','contact_grad','g2p') if n in k})
PY
