#!/bin/bash
# round 2, GPU session Z: non-temporal stores for the frame rows written once per substep (-DSMAC_NT_STORES=1) vs default, 4 interleaved processes each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
bash tools/ab_runs.sh $O 4 base=libsoftmac_hip.so nt=libsoftmac_hip_nt.so 2>&1 | tail -3
python3 - $O <<'PY'
import json, sys, glob
O = sys.argv[1]
for label in ("base", "nt"):
    acc = {}
    for f in sorted(glob.glob(f"{O}/run_{label}_*.json")):
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        for k, v in d['kernels_ms'].items():
            acc.setdefault(k, []).append(round(v * 1e3, 1))
    print(label, {k: v for k, v in acc.items() if k in ('p2g', 'g2p', 'g2p_grad', 'p2g_grad', 'grid_checkpoint')})
PY
