#!/bin/bash
# round 3, GPU session U: ranks kept across re-sorts (SMAC_STABLE_RANKS, default 1) - sort / reorder times on one library, the re-binning parity tests, the new ride-along test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
bash tools/ab3.sh $O fresh=libsoftmac_hip.so,SMAC_STABLE_RANKS=0 stable=libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
python3 - <<'PY'
import json,glob
for lab in ("fresh","stable"):
    for f in sorted(glob.glob(f"gpurun_out/r03u/ab_{lab}_*.json")):
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels_ms']
        print(lab, round(d['value'],1), {n: round(k[n]*1e3,1) for n in ('sort','reorder_adjoint','p2g','p2g_g2p_grad','g2p') if n in k})
PY
timeout -k 10 1000 python3 -m pytest tests/test_gpu_fused_backward.py tests/test_gpu_fullsize.py tests/test_gpu_long_rollout.py tests/test_gpu_parity.py tests/test_gpu_slab_lib.py -x -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc $?"; grep "on vs off\|passed\|failed" $O/tests.log | cut -c1-300
