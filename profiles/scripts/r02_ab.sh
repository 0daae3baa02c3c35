#!/bin/bash
# round 2, GPU session AB: contact-exchange shortcut - slab test against the single-domain run, bench --gpus 2 over gloo (strong) with and without it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ab; mkdir -p $O
make -C oracle -s
timeout -k 10 900 python -m pytest tests/test_slabs.py -m gpu -q > $O/pytest_slabs.log 2>&1; tail -4 $O/pytest_slabs.log | cut -c1-300
SMAC_DIST_BACKEND=gloo SMAC_FORCE_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 4 --no-cpu-baseline --no-f64 --no-cloth --repeats 1 > $O/bench_n2_strong.json 2> $O/bench_n2_strong.err; cut -c1-260 $O/bench_n2_strong.json; tail -1 $O/bench_n2_strong.err | cut -c1-200
SMAC_SLAB_ALL_EXCHANGES=1 SMAC_DIST_BACKEND=gloo SMAC_FORCE_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 4 --no-cpu-baseline --no-f64 --no-cloth --repeats 1 > $O/bench_n2_strong_all.json 2> $O/bench_n2_strong_all.err; cut -c1-260 $O/bench_n2_strong_all.json
