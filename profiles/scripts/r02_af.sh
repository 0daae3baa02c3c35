#!/bin/bash
# round 2, GPU session AF: whole GPU suite on the final code (k_g2p LDS gather, three-tier clamp tolerance), driver-style bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02af; mkdir -p $O
make -C oracle -s
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-300
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; cut -c1-300 $O/bench_driver_style.json
