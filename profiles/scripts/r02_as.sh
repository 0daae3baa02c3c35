#!/bin/bash
# round 2, GPU session AS: k_p2g starts its SVD from the previous substep's filed V (SMAC_WARM_SVD=1, default) vs from the identity (=0); p2g.grad from the same substep's V in both; 4 processes each; whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02as; mkdir -p $O
make -C oracle -s
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "grip_fixture or one_substep or pour" > $O/pytest_quick.log 2>&1; rc=$?; tail -3 $O/pytest_quick.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "quick parity failed"; tail -40 $O/pytest_quick.log | cut -c1-300; exit 1; fi
bash tools/ab_runs.sh $O 3 cold=libsoftmac_hip.so,SMAC_WARM_SVD=0 warm=libsoftmac_hip.so,SMAC_WARM_SVD=1 nov=libsoftmac_hip.so,SMAC_KEEP_V=0 2>&1 | tail -3
python3 - $O <<'PY'
import json, sys, glob
O = sys.argv[1]
for label in ("cold", "warm", "nov"):
    acc = {}; best = []
    for f in sorted(glob.glob(f"{O}/run_{label}_*.json")):
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        best.append(min(d['ms_per_step_all']))
        for k, v in d['kernels_ms'].items():
            acc.setdefault(k, []).append(round(v * 1e3, 1))
    print(label, 'best window ms', best, {k: v for k, v in acc.items() if k in ('p2g', 'p2g_g2p_grad', 'g2p')})
PY
timeout -k 10 800 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.log 2>&1; tail -12 $O/pytest.log | cut -c1-300
