#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bc; mkdir -p $O
for cfg in "4194304 256" "262144 64"; do set -- $cfg
  timeout -k 10 400 python3 bench.py --particles $1 --grid $2 --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_$1.json 2> $O/bench_$1.err; echo "rc $?"
  python3 -c "
import json;d=json.loads([l for l in open('$O/bench_$1.json') if l.startswith('{')][-1]); print($1, $2, round(d['value'],1), d['ms_per_step_all'], round(d['roofline']['frac'],3), round(d['roofline_substep']['frac'],3), d['config']['touched_cells'], d['config']['contact_particles'])"
done
