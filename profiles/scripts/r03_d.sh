#!/bin/bash
# round 3, GPU session D: bench line with the fwd_only / bwd_only / env_loop sub-records, then the whole GPU suite on the committed kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "import json;d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]);print(d['value'],d['ms_per_step_all']);print(d.get('fwd_only'));print(d.get('bwd_only'));print(d.get('env_loop'))"
timeout -k 10 1500 python3 -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1 || { tail -40 $O/gpu_suite.log; exit 1; }
tail -3 $O/gpu_suite.log
