#!/bin/bash
# round 4, session L: where do the 35 us go when the contact adjoint rides in the fused backward launch?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
for v in "0 0" "1 0" "1 1" "1 2" "1 3"; do
  set -- $v
  SMAC_CONTACT_RIDE=$1 SMAC_RIDE_DEBUG=$2 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 3 > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err || exit 1
  python3 -c "
import json;d=json.loads([l for l in open('$O/bench_$1_$2.json') if l.startswith('{')][-1]); print('ride $1 debug $2', round(d['value'],1), 'bwd', round(d['bwd_only']['ms_per_step']*1e3,1), {k: round(x*1e3,1) for k,x in d['kernels_ms_per_step'].items() if k in ('contact_grad','p2g_g2p_grad','reduce_agvout')})"
done
