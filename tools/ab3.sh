#!/bin/bash
# A/B of several builds of libsoftmac_hip.so, 3 interleaved rounds (the particle kernels are bimodal PER PROCESS, profiles/HISTORY.md 7): prints per
# variant the best substeps/s and the per-kernel minima over its runs.   tools/ab3.sh out_dir label=lib.so[,ENV=V...] ...
O=$1; shift
mkdir -p $O
for round in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}; lib=${rest%%,*}; envs=""
    if [[ "$rest" == *,* ]]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
    env SMAC_LIB=$PWD/softmac_amd/lib/$lib $envs timeout -k 10 300 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 2 > $O/ab_${label}_$round.json 2> $O/ab_${label}_$round.err
  done
done
python3 - "$O" "$@" <<'PY'
import json, sys, glob
O = sys.argv[1]
keys = ('p2g','grid_op','contact','g2p','grid_checkpoint','g2p_grad','reduce_agvout','contact_grad','grid_op_grad','p2g_grad','clear_grid')
for spec in sys.argv[2:]:
    label = spec.split('=')[0]
    best, kmin = 0.0, {}
    for f in sorted(glob.glob(f"{O}/ab_{label}_*.json")):
        try:
            d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        except Exception as e:
            print(label, "FAILED", f); continue
        best = max(best, d['value'])
        for k, v in d['kernels_ms'].items():
            kmin[k] = min(kmin.get(k, 1e9), v)
    print(f"{label:10s} best {best:7.1f}/s  min-sum {sum(kmin.get(k,0) for k in keys)*1e3:6.1f} us ", {k: round(kmin[k]*1e3,1) for k in keys if k in kmin})
PY
