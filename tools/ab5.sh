#!/bin/bash
# Round-5 A/B: interleaved runs of several builds of libsoftmac_hip.so under the driver's bench command (--steps 20 --warmup 5), every run printed with the
# per-kernel times (us per launch) that matter this round.   tools/ab5.sh out_dir rounds label=lib.so[,ENV=V...] ...
O=$1; R=$2; shift 2
mkdir -p $O
for round in $(seq 1 $R); do
  for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}; lib=${rest%%,*}; envs=""; flags=""
    if [[ "$rest" == *,* ]]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
    # (an entry BENCH_FLAGS=--sort-interval:80 adds bench.py flags, ':' for the blank)
    for e in $envs; do if [[ "$e" == BENCH_FLAGS=* ]]; then flags=$(echo "${e#BENCH_FLAGS=}" | tr ':' ' '); fi; done
    env SMAC_LIB=$PWD/softmac_amd/lib/$lib $envs timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop $flags > $O/run_${label}_$round.json 2> $O/run_${label}_$round.err || echo "$label round $round FAILED (see $O/run_${label}_$round.err)"
  done
done
python3 - "$O" "$@" <<'PY'
import json, sys, glob
O = sys.argv[1]
keys = ("g2p_p2g", "grid_op", "contact", "reduce_agvout", "contact_grad", "p2g_g2p_grad")
for spec in sys.argv[2:]:
    label = spec.split('=')[0]
    rows = []
    for f in sorted(glob.glob(f"{O}/run_{label}_*.json")):
        try:
            d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        except Exception:
            print(label, "FAILED", f); continue
        rows.append((d['value'], d['device_ms_per_step'] * 1e3, [d['kernels_ms'].get(k, 0) * 1e3 for k in keys]))
    if not rows:
        continue
    mean = sum(r[0] for r in rows) / len(rows)
    print(f"{label:12s} substeps/s {[round(r[0], 1) for r in rows]} mean {mean:7.1f}   device us/pair {[round(r[1], 1) for r in rows]}")
    for i, k in enumerate(keys):
        print(f"{'':12s}   {k:14s} us/launch {[round(r[2][i], 1) for r in rows]}")
PY
