#!/bin/bash
# round 2, GPU session Z2: non-temporal stores in k_p2g_grad only (default now) vs plain vs + non-temporal loads of the rows k_p2g_grad reads last, 4 processes each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z2; mkdir -p $O
bash tools/ab_runs.sh $O 4 plain=libsoftmac_hip_plain.so nts=libsoftmac_hip.so ntl=libsoftmac_hip_ntl.so 2>&1 | tail -4
python3 - $O <<'PY'
import json, sys, glob
O = sys.argv[1]
for label in ("plain", "nts", "ntl"):
    acc = {}
    for f in sorted(glob.glob(f"{O}/run_{label}_*.json")):
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        for k, v in d['kernels_ms'].items():
            acc.setdefault(k, []).append(round(v * 1e3, 1))
    print(label, {k: v for k, v in acc.items() if k in ('p2g', 'g2p', 'g2p_grad', 'p2g_grad', 'grid_checkpoint', 'reduce_agvout')})
PY
