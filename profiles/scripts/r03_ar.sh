#!/bin/bash
# round 3, GPU session AR: bench.py --gpus 2 with the in-library runner where it CANNOT come up (both ranks on the one GPU: RCCL refuses duplicate devices) -
# every rank must take the Python loop instead and the line must say so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ar; mkdir -p $O
SMAC_FORCE_DEVICE=0 timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 4 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 1 > $O/bench_n2_fallback.json 2> $O/bench_n2_fallback.err; echo "rc $?"
python3 -c "
import json;d=json.loads([l for l in open('$O/bench_n2_fallback.json') if l.startswith('{')][-1]); print(round(d['value'],1), d['n_gpus'], d.get('slab_runner'))"; grep -i "bench.py:\|error\|duplicate" $O/bench_n2_fallback.err | cut -c1-400 | head -8
