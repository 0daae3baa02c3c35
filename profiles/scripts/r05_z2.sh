#!/bin/bash
# round 5, final GPU session part 2: the driver-style bench line and the default-flags line of the headline workload, the C4 / C5 lines (S-pour 4M / 256^3, S-mixed 16M / 256^3).
# Run AFTER part 1's traffic_latest.json has been copied to profiles/ (bench.py quotes roofline.traffic from it when the kernel-source hash matches).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z2; mkdir -p $O
make -C oracle -s
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; python3 -c "
import json;d=json.loads([l for l in open('$O/bench_driver_style.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step_all'], d['roofline'], d['roofline_substep']['frac'])
for k in ('fwd_only','bwd_only','env_loop','f64','cloth','cpu_baseline'): print(k, str(d.get(k))[:300])"
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json;d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1]); print('default flags:', d['value'], d['steps'], d['warmup'], d['roofline']['frac'], d['cpu_baseline'])"
timeout -k 10 500 python bench.py --workload s-pour --steps 20 --warmup 5 > $O/bench_c4_s_pour.json 2> $O/bench_c4_s_pour.err && python3 -c "
import json;d=json.loads([l for l in open('$O/bench_c4_s_pour.json') if l.startswith('{')][-1]); print('s-pour', d['value'], d['ms_per_step_all'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_substep']['frac'], d.get('drift_repairs'), str(d.get('cpu_baseline'))[:300])" &&
timeout -k 10 500 python bench.py --workload s-pour --steps 20 --warmup 25 --no-cpu-baseline > $O/bench_c4_s_pour_warmup25.json 2> $O/bench_c4_s_pour_warmup25.err && python3 -c "
import json;d=json.loads([l for l in open('$O/bench_c4_s_pour_warmup25.json') if l.startswith('{')][-1]); print('s-pour warmup 25', d['value'], d['ms_per_step_all'], d.get('drift_repairs'))" &&
timeout -k 10 900 python bench.py --workload s-mixed > $O/bench_c5_s_mixed.json 2> $O/bench_c5_s_mixed.err && python3 -c "
import json;d=json.loads([l for l in open('$O/bench_c5_s_mixed.json') if l.startswith('{')][-1]); print('s-mixed', d['value'], d['ms_per_step_all'], d['roofline']['kernel'], d['roofline']['frac'], d['drift_repairs'], str(d.get('cpu_baseline'))[:300])"
echo "done rc=$?"; tail -3 $O/*.err 2>/dev/null | cut -c1-300
