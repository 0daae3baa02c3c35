#!/bin/bash
# round 2, final GPU session: whole GPU suite + smoke + driver-style bench line on the final code
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-200
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; cut -c1-300 $O/bench_driver_style.json
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
