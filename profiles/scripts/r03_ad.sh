#!/bin/bash
# round 3, GPU session AD: get_action_grad of a whole episode in one call per primitive (TaichiEnv.backward): env tests, env_loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ad; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_env.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log | cut -c1-300
for v in 1 1; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth > $O/bench.json 2> $O/bench.err
  python3 -c "
import json;d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
e=d['env_loop']; print('value', round(d['value'],1), 'env_loop', round(e['value'],1), e.get('ms_per_step_all'), 'ratio', round(e['value']/d['value'],3))"
done
