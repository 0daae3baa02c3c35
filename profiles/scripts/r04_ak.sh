#!/bin/bash
# round 4, session AK: S-grip at other sizes on the final kernels (bench.py --steps 20 --warmup 5, f32, fwd + bwd): particles, grid, substeps/s, windows, roofline fracs.
# 4M / 256^3: ONE window (frames 5 .. 25, interval 20) - the synthetic scene keeps dt = 1e-4 at half the cell size and blows up from frame 22 on (vmax 15 -> 1,976 m/s:
# tools/drift_probe.py, the same on the round-3 kernels): not an episode to advance through
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ak; mkdir -p $O
for cfg in "4194304 256 --repeats 1 --sort-interval 20" "262144 64" "65536 64"; do
  set -- $cfg
  timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 --particles $1 --grid $2 $3 $4 $5 $6 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err || { tail -5 $O/bench_$1_$2.err; exit 1; }
  python3 -c "
import json;d=json.loads([l for l in open('$O/bench_$1_$2.json') if l.startswith('{')][-1]); print($1, $2, round(d['value'],1), d['ms_per_step_all'], round(d['roofline']['frac'],3), round(d['roofline_substep']['frac'],3), d['config']['touched_cells'], d['config']['contact_particles'])"
done
