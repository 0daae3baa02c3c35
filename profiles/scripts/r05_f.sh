#!/bin/bash
# round 5, GPU session F: window-parity test with offender diagnostics, the C4 / C5 bench lines (S-pour 4M / 256^3, S-mixed 16M / 256^3), the 120-frame stability probe of
# S-grip at 4M / 256^3 with the Courant-scaled dt, the driver-style bench line of the headline workload
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
make -C oracle -s
timeout -k 10 300 python -m pytest tests/test_gpu_window_parity.py -q -x -s -p no:cacheprovider -k headline > $O/tests_window.txt 2>&1; echo "window rc=$?"
timeout -k 10 300 python tools/drift_probe.py 4194304 256 40 > $O/drift_probe_4m.txt 2>&1 && tail -4 $O/drift_probe_4m.txt &&
timeout -k 10 500 python bench.py --workload s-pour --steps 20 --warmup 5 > $O/bench_c4_s_pour.json 2> $O/bench_c4_s_pour.err && python3 -c "
import json;d=json.loads([l for l in open('$O/bench_c4_s_pour.json') if l.startswith('{')][-1]); print('s-pour', d['value'], d['ms_per_step_all'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_substep']['frac'], str(d.get('cpu_baseline'))[:300])" &&
timeout -k 10 900 python bench.py --workload s-mixed > $O/bench_c5_s_mixed.json 2> $O/bench_c5_s_mixed.err && python3 -c "
import json;d=json.loads([l for l in open('$O/bench_c5_s_mixed.json') if l.startswith('{')][-1]); print('s-mixed', d['value'], d['ms_per_step_all'], d['roofline']['kernel'], d['roofline']['frac'], d['drift_repairs'], str(d.get('cpu_baseline'))[:300])" &&
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err && python3 -c "
import json;d=json.loads([l for l in open('$O/bench_driver_style.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step_all'], d['roofline'], d['roofline_substep']['frac'])
for k in ('fwd_only','bwd_only','env_loop','f64','cloth','cpu_baseline'): print(k, str(d.get(k))[:300])"
echo "done rc=$?"; tail -3 $O/*.err 2>/dev/null | cut -c1-300
