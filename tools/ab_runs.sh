#!/bin/bash
# like ab3.sh, but prints EVERY run (value and the per-kernel times that are bimodal per process) - for variants whose point is to be
# insensitive to the fast/slow mode of profiles/HISTORY.md 7.   tools/ab_runs.sh out_dir rounds label=lib.so[,ENV=V...] ...
O=$1; R=$2; shift 2
mkdir -p $O
for round in $(seq 1 $R); do
  for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}; lib=${rest%%,*}; envs=""
    if [[ "$rest" == *,* ]]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
    env SMAC_LIB=$PWD/softmac_amd/lib/$lib $envs timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-f64 --no-cloth --repeats 2 > $O/run_${label}_$round.json 2> $O/run_${label}_$round.err
  done
done
python3 - "$O" "$@" <<'PY'
import json, sys, glob
O = sys.argv[1]
for spec in sys.argv[2:]:
    label = spec.split('=')[0]
    vals, g2p, g2pg = [], [], []
    for f in sorted(glob.glob(f"{O}/run_{label}_*.json")):
        try:
            d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        except Exception:
            print(label, "FAILED", f); continue
        vals.append(round(d['value'], 1)); g2p.append(round(d['kernels_ms']['g2p'] * 1e3, 1)); g2pg.append(round(d['kernels_ms']['g2p_grad'] * 1e3, 1))
    print(f"{label:10s} substeps/s {vals}  mean {sum(vals)/max(len(vals),1):7.1f}   g2p us {g2p}   g2p_grad us {g2pg}")
PY
