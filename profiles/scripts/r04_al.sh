#!/bin/bash
# round 4, session AL: the drift error of S-grip 4M / 256^3 under the new defaults - which switch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04al; mkdir -p $O
for v in "1 40" "0 40" "1 20" "1 32"; do
  set -- $v
  SMAC_FUSED_FWD=$1 timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 --particles 4194304 --grid 256 --sort-interval $2 --repeats 4 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  echo "fused_fwd $1 interval $2 rc $?"; tail -1 $O/bench_$1_$2.err | cut -c1-200
  python3 -c "
import json;d=json.loads([l for l in open('$O/bench_$1_$2.json') if l.startswith('{')][-1]); print(round(d['value'],1), d['ms_per_step_all'], d['config']['resorts_in_windows'])" 2>/dev/null
done
