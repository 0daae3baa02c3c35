#!/bin/bash
# round 4, session J: do the phases of the fused backward kernel overlap better with (1, 2) wave priorities by phase or (3) a staggered first round?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
for round in 1 2; do
  for v in base exp1 exp2 exp3; do
    lib=libsoftmac_hip.so; [ $v != base ] && lib=libsoftmac_hip_$v.so
    SMAC_LIB=$PWD/softmac_amd/lib/$lib timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 3 > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err
    python3 -c "
import json;d=json.loads([l for l in open('$O/bench_${v}_$round.json') if l.startswith('{')][-1]); print('$v round $round', round(d['value'],1), d['ms_per_step_all'], {k: round(x*1e3,1) for k,x in d['kernels_ms_per_step'].items() if k in ('p2g','g2p','p2g_g2p_grad','reduce_agvout')})"
  done
done
