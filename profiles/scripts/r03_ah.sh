#!/bin/bash
# round 3, GPU session AH: binning by the position half way through the interval (SMAC_SORT_LOOKAHEAD) - A/B on one library at two re-sort intervals
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ah; mkdir -p $O
bash tools/ab3.sh $O now=libsoftmac_hip.so half=libsoftmac_hip.so,SMAC_SORT_LOOKAHEAD=0.5 > $O/ab.txt 2>&1; cat $O/ab.txt
python3 - <<'PY'
import json,glob
for lab in ("now","half"):
    for f in sorted(glob.glob(f"gpurun_out/r03ah/ab_{lab}_*.json")):
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels_ms']
        print(lab, round(d['value'],1), {n: round(k[n]*1e3,1) for n in ('p2g','g2p','p2g_g2p_grad','sort') if n in k})
PY
SMAC_SORT_LOOKAHEAD=0.5 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_resort.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests (lookahead on) rc $?"; tail -3 $O/tests.log | cut -c1-300
