#!/bin/bash
# round 2, GPU session AO: k_p2g at 4 / 5 workgroups per CU (LDS padding, registers unchanged) vs its 6 - what a fused k_g2p + k_p2g (113 VGPRs, 4 waves/SIMD) would give its P2G half
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ao; mkdir -p $O
bash tools/ab_runs.sh $O 3 occ6=libsoftmac_hip.so occ5=libsoftmac_hip_occ5.so occ4=libsoftmac_hip_occ4.so 2>&1 | tail -4
python3 - $O <<'PY'
import json, sys, glob
O = sys.argv[1]
for label in ("occ6", "occ5", "occ4"):
    acc = {}; best = []
    for f in sorted(glob.glob(f"{O}/run_{label}_*.json")):
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        best.append(min(d['ms_per_step_all']))
        for k, v in d['kernels_ms'].items():
            acc.setdefault(k, []).append(round(v * 1e3, 1))
    print(label, 'best window ms', best, {k: v for k, v in acc.items() if k in ('p2g', 'g2p', 'grid_op')})
PY
