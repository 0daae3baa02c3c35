#!/bin/bash
# round 4, session AC: fused forward step (k_g2p_p2g) - self-comparison + oracle parity, then SMAC_FUSED_FWD = 0 / 1 on one library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ac; mkdir -p $O
SMAC_PRINT_ERRS=$PWD/$O/errs.txt timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_forward.py tests/test_gpu_parity.py tests/test_gpu_env.py -x -q > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $O/pytest.log | head -30; exit $rc; }
for round in 1 2; do
  for v in 0 1; do
    SMAC_FUSED_FWD=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_fwd${v}_$round.json 2> $O/bench_fwd${v}_$round.err || { tail -5 $O/bench_fwd${v}_$round.err; exit 1; }
    python3 -c "
import json;d=json.loads([l for l in open('$O/bench_fwd${v}_$round.json') if l.startswith('{')][-1]); print('fused_fwd $v round $round', round(d['value'],1), d['ms_per_step_all'], 'fwd', round(d['fwd_only']['ms_per_step']*1e3,1), 'bwd', round(d['bwd_only']['ms_per_step']*1e3,1), {k: round(x*1e3,1) for k,x in d['kernels_ms_per_step'].items()}, 'launch', round(d['roofline']['avg_launch_ms']*1e3,1))"
  done
done
