#!/bin/bash
# round 2, GPU session AT: the library as committed at the end of the round - smoke, whole GPU suite, driver-style bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02at; mkdir -p $O
make -C oracle -s
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-200
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; tail -9 $O/pytest.log | cut -c1-300
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err && cut -c1-400 $O/bench_driver_style.json
