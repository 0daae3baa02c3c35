#!/bin/bash
# round 2, GPU session Z3: non-temporal stores for k_p2g's F rows (ntf) and for the checkpoint records (ntck) vs default, 4 processes each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z3; mkdir -p $O
bash tools/ab_runs.sh $O 4 base=libsoftmac_hip.so ntf=libsoftmac_hip_ntf.so ntck=libsoftmac_hip_ntck.so 2>&1 | tail -4
python3 - $O <<'PY'
import json, sys, glob
O = sys.argv[1]
for label in ("base", "ntf", "ntck"):
    acc = {}; second = []
    for f in sorted(glob.glob(f"{O}/run_{label}_*.json")):
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        second.append(min(d['ms_per_step_all']))
        for k, v in d['kernels_ms'].items():
            acc.setdefault(k, []).append(round(v * 1e3, 1))
    print(label, 'best window ms', second, {k: v for k, v in acc.items() if k in ('p2g', 'g2p', 'g2p_grad', 'p2g_grad', 'grid_checkpoint', 'grid_op')})
PY
