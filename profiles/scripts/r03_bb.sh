#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bb; mkdir -p $O
make -C oracle -s
s=$(date +%s); timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc $? in $(( $(date +%s) - s )) s"
python3 -c "
import json;d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1]); print(d['value'], d['steps'], d['warmup'], d['ms_per_step_all'], d['roofline']['frac'], d['env_loop']['value'], d['cpu_baseline']['value'])"
