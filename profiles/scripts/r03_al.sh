#!/bin/bash
# round 3, GPU session AL: blocks without mass are not filed / read back (SMAC_CK_SKIP_EMPTY, default 1) - A/B on one library, parity, PMC traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03al; mkdir -p $O
bash tools/ab3.sh $O all=libsoftmac_hip.so,SMAC_CK_SKIP_EMPTY=0 skip=libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
python3 - <<'PY'
import json,glob
for lab in ("all","skip"):
    for f in sorted(glob.glob(f"gpurun_out/r03al/ab_{lab}_*.json")):
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels_ms']
        print(lab, round(d['value'],1), d['ms_per_step_all'], {n: round(k[n]*1e3,1) for n in ('g2p','reduce_agvout','grid_checkpoint','p2g_g2p_grad') if n in k})
PY
timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_slab_lib.py tests/test_gpu_fused_backward.py tests/test_gpu_cloth.py tests/test_gpu_api_sequences.py tests/test_gpu_windowed.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log | cut -c1-300
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $B > $O/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $B > $O/write.log 2>&1 &&
python3 tools/pmc_traffic.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/traffic_latest.json "r03 session AL" > $O/pmc_traffic.csv && cat $O/pmc_traffic.csv
