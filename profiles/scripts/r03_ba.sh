#!/bin/bash
# round 3, GPU session BA: an adjoint frame crosses a re-sort by one gather through the sort's own destination map (no k_invert / k_compose) - the tests that cross re-sorts, a bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ba; mkdir -p $O
timeout -k 10 700 python3 -m pytest tests/test_gpu_fused_backward.py tests/test_gpu_parity.py tests/test_gpu_long_rollout.py tests/test_gpu_api_sequences.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log | cut -c1-300
for i in 1 2 3; do timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_$i.json 2> $O/bench_$i.err; python3 -c "
import json;d=json.loads([l for l in open('$O/bench_$i.json') if l.startswith('{')][-1]); print(round(d['value'],1), d['ms_per_step_all'], round(d['bwd_only']['ms_per_step'],4))"; done
