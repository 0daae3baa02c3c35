#!/bin/bash
# round 3, GPU session G: rebinning-invariance test with explained exceedances, cloth broad phase (1M equivalence + the 16M / 256^3 C5 slice),
# bench with env_loop after binning-at-reset, host cost of the slab loops
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "rebinning" > $O/fullsize.log 2>&1; echo "fullsize rc $?"
grep -E "^\[|unexplained|passed|failed" $O/fullsize.log | cut -c1-900
timeout -k 10 900 python3 -m pytest tests/test_gpu_cloth.py -x -q -m gpu -s > $O/cloth.log 2>&1; echo "cloth rc $?"
grep -E "^\[C5|passed|failed|Error" $O/cloth.log | cut -c1-600
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "import json;d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]);print(d['value'],d['ms_per_step_all'],d['kernels_ms']);print(d.get('env_loop'))"
timeout -k 10 600 python3 tools/exchange_overhead.py > $O/exchange_overhead.txt 2>&1 || { tail -20 $O/exchange_overhead.txt; exit 1; }
cat $O/exchange_overhead.txt
