#!/bin/bash
# round 2, GPU session AK: k_p2g_grad(f) + k_g2p_grad(f-1) as one launch (SMAC_FUSED_PG=1, default) vs apart (=0), 4 processes each; its own test first, then the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ak; mkdir -p $O
make -C oracle -s
timeout -k 10 300 python -m pytest tests/test_gpu_fused_backward.py -m gpu -q -x -s > $O/pytest_fused.log 2>&1; rc=$?; tail -4 $O/pytest_fused.log
if [ $rc -ne 0 ]; then echo "fused test failed"; tail -40 $O/pytest_fused.log; exit 1; fi
bash tools/ab_runs.sh $O 4 apart=libsoftmac_hip.so,SMAC_FUSED_PG=0 fused=libsoftmac_hip.so,SMAC_FUSED_PG=1 2>&1 | tail -3
python3 - $O <<'PY'
import json, sys, glob
O = sys.argv[1]
for label in ("apart", "fused"):
    acc = {}; best = []
    for f in sorted(glob.glob(f"{O}/run_{label}_*.json")):
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        best.append(min(d['ms_per_step_all']))
        for k, v in d['kernels_ms'].items():
            acc.setdefault(k, []).append(round(v * 1e3, 1))
    print(label, 'best window ms', best, {k: v for k, v in acc.items() if k in ('g2p_grad', 'p2g_grad', 'p2g_g2p_grad', 'grid_checkpoint', 'reduce_agvout')})
PY
timeout -k 10 800 python -m pytest tests -m gpu -q -x --durations=12 > $O/pytest.log 2>&1; tail -20 $O/pytest.log
