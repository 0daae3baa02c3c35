#!/bin/bash
# round 3, GPU session Q: particle rows issued before the first barrier (k_p2g: SMAC_P2G_EARLY_ROWS; p2g.grad kernels: SMAC_PGG_EARLY_ROWS) - A/B; phase clock with entry markers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
bash tools/ab3.sh $O new=libsoftmac_hip.so early=libsoftmac_hip_early.so pggearly=libsoftmac_hip_pggearly.so early2=libsoftmac_hip_early2.so > $O/ab.txt 2>&1; cat $O/ab.txt
python3 - <<'PY'
import json,glob
for lab in ("new","early","pggearly","early2"):
    for f in sorted(glob.glob(f"gpurun_out/r03q/ab_{lab}_*.json")):
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels_ms']
        print(lab, round(d['value'],1), {n: round(k[n],4) for n in ('p2g','p2g_g2p_grad','p2g_grad','g2p_grad') if n in k})
PY
SMAC_LIB=$PWD/softmac_amd/lib/libsoftmac_hip_phase.so SMAC_PHASE_DUMP=$PWD/$O/phase.txt timeout -k 10 300 python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 2 > $O/bench_phase.json 2> $O/bench_phase.err
python3 tools/phase_clock.py $O/phase.txt | tee $O/phase_report.txt
