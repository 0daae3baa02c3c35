#!/bin/bash
# round 2, GPU session AP: k_p2g asked to fit 7 / 8 waves per SIMD (72 VGPRs + 32 B scratch / 64 VGPRs + 84 B scratch) vs its 6 (80 VGPRs, no scratch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ap; mkdir -p $O
bash tools/ab_runs.sh $O 3 occ6=libsoftmac_hip.so occ7=libsoftmac_hip_p2g7.so occ8=libsoftmac_hip_p2g8.so 2>&1 | tail -4
python3 - $O <<'PY'
import json, sys, glob
O = sys.argv[1]
for label in ("occ6", "occ7", "occ8"):
    acc = {}; best = []
    for f in sorted(glob.glob(f"{O}/run_{label}_*.json")):
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        best.append(min(d['ms_per_step_all']))
        for k, v in d['kernels_ms'].items():
            acc.setdefault(k, []).append(round(v * 1e3, 1))
    print(label, 'best window ms', best, {k: v for k, v in acc.items() if k in ('p2g', 'g2p', 'grid_op')})
PY
