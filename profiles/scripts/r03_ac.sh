#!/bin/bash
# round 3, GPU session AC: forward_kinematics / its adjoint riding in k_g2p's / the reduction's launch (SMAC_FK_RIDE, default 1): env tests, then env_loop with it off and on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ac; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_env.py tests/test_gpu_api_sequences.py tests/test_gpu_fused_backward.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log | cut -c1-300
for v in 0 1 0 1; do
  SMAC_FK_RIDE=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth > $O/bench_fk$v.json 2> $O/bench_fk$v.err
  python3 -c "
import json;d=json.loads([l for l in open('$O/bench_fk$v.json') if l.startswith('{')][-1])
e=d['env_loop']; print('SMAC_FK_RIDE=$v value', round(d['value'],1), 'env_loop', round(e['value'],1), e.get('ms_per_step_all'), 'ratio', round(e['value']/d['value'],3))"
done
