#!/bin/bash
# round 2, GPU session AJ: Jacobi rotation from two rsqrt (svd), + LDS tile strides (52, 8) instead of (36, 6) (new) vs the previous commit (prev), 3 processes each; parity suites on the new library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02aj; mkdir -p $O
make -C oracle -s
bash tools/ab_runs.sh $O 3 prev=libsoftmac_hip_prev.so svd=libsoftmac_hip_svd.so new=libsoftmac_hip.so 2>&1 | tail -4
python3 - $O <<'PY'
import json, sys, glob
O = sys.argv[1]
for label in ("prev", "svd", "new"):
    acc = {}; best = []
    for f in sorted(glob.glob(f"{O}/run_{label}_*.json")):
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        best.append(min(d['ms_per_step_all']))
        for k, v in d['kernels_ms'].items():
            acc.setdefault(k, []).append(round(v * 1e3, 1))
    print(label, 'best window ms', best, {k: v for k, v in acc.items() if k in ('g2p', 'p2g_grad', 'g2p_grad', 'p2g')})
PY
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_cloth.py -m gpu -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
