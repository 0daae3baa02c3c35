#!/bin/bash
# round 4, session Q (and S: with the composed map): the backward sweep crosses a re-sort through a map inside the kernels (SMAC_AN_MAP) instead of a gather pass - parity, then A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_fused_backward.py tests/test_gpu_parity.py tests/test_gpu_resort.py tests/test_gpu_long_rollout.py -x -q > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -8 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for round in 1 2; do
  for v in 0 1; do
    SMAC_AN_MAP=$v timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 3 > $O/bench_map${v}_$round.json 2> $O/bench_map${v}_$round.err || exit 1
    python3 -c "
import json;d=json.loads([l for l in open('$O/bench_map${v}_$round.json') if l.startswith('{')][-1]); print('map $v round $round', round(d['value'],1), d['ms_per_step_all'], 'bwd', round(d['bwd_only']['ms_per_step']*1e3,1), {k: round(x*1e3,1) for k,x in d['kernels_ms_per_step'].items() if k in ('reorder_adjoint','p2g_g2p_grad','p2g_grad','g2p_grad')})"
  done
done
