#!/bin/bash
# round 2, GPU session T: cloth tests after the hit-list filter and the tightened tolerances; cloth timing; whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
make -C oracle -s
timeout -k 10 400 python -m pytest tests/test_gpu_cloth.py -m gpu -q -s 2>&1 | grep "cloth\|passed\|failed\|Error" > $O/cloth.txt; cat $O/cloth.txt | cut -c1-200
timeout -k 10 300 python tools/bench_cloth.py > $O/bench_cloth.json 2> $O/bench_cloth.err; cat $O/bench_cloth.json | cut -c1-900
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
