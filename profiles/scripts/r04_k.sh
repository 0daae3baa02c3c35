#!/bin/bash
# round 4, session K: the contact adjoint inside the fused backward launch (SMAC_CONTACT_RIDE) - parity first, then A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_backward.py tests/test_gpu_parity.py -x -q -s -k "ride_along or batched or fused" > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -25 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for round in 1 2; do
  for v in 0 1; do
    SMAC_CONTACT_RIDE=$v timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 3 > $O/bench_ride${v}_$round.json 2> $O/bench_ride${v}_$round.err || exit 1
    python3 -c "
import json;d=json.loads([l for l in open('$O/bench_ride${v}_$round.json') if l.startswith('{')][-1]); print('ride $v round $round', round(d['value'],1), d['ms_per_step_all'], 'bwd', d['bwd_only']['ms_per_step'], {k: round(x*1e3,1) for k,x in d['kernels_ms_per_step'].items() if k in ('contact_grad','p2g_g2p_grad','reduce_agvout')})"
  done
done
