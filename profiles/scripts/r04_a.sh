#!/bin/bash
# round 4, session A: (1) parity of the early-exit / chunk-wave build at 4 and 3 waves per chunk, (2) A/B of the chunk size on the benchmark
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
make -C oracle -s
for cw in 4 3; do
  SMAC_CHUNK_WAVES=$cw timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_backward.py tests/test_gpu_resort.py -x -q -m gpu > $O/pytest_cw$cw.log 2>&1
  echo "pytest cw=$cw rc $?"; tail -3 $O/pytest_cw$cw.log
done
for round in 1 2; do
  for cw in 4 3 2; do
    SMAC_CHUNK_WAVES=$cw timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 3 > $O/bench_cw${cw}_$round.json 2> $O/bench_cw${cw}_$round.err
    python3 -c "
import json;d=json.loads([l for l in open('$O/bench_cw${cw}_$round.json') if l.startswith('{')][-1]); print('cw $cw round $round', round(d['value'],1), d['ms_per_step_all'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_step'].items()})"
  done
done
