#!/bin/bash
# round 2, GPU session AD: k_g2p with a wave-uniform pure-LDS gather (no flat loads) + XCD-contiguous ranges for the per-cell grid kernels, vs the previous commit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ad; mkdir -p $O
make -C oracle -s
bash tools/ab_runs.sh $O 5 prev=libsoftmac_hip_prev.so new=libsoftmac_hip.so noxcd=libsoftmac_hip.so,SMAC_XCD_GRID=0 2>&1 | tail -4
python3 - $O <<'PY'
import json, sys, glob
O = sys.argv[1]
for label in ("prev", "new", "noxcd"):
    acc = {}; best = []
    for f in sorted(glob.glob(f"{O}/run_{label}_*.json")):
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        best.append(min(d['ms_per_step_all']))
        for k, v in d['kernels_ms'].items():
            acc.setdefault(k, []).append(round(v * 1e3, 1))
    print(label, 'best window ms', best, {k: v for k, v in acc.items() if k in ('g2p', 'grid_op', 'reduce_agvout', 'grid_checkpoint', 'p2g')})
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
