#!/bin/bash
# round 2, GPU session V: cloth tests incl. the penalty contact (collision_type 1); host overhead of the Python-driven phase loop (world 1, no exchange)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02v; mkdir -p $O
make -C oracle -s
timeout -k 10 500 python -m pytest tests/test_gpu_cloth.py -m gpu -q -s 2>&1 | grep "cloth\|passed\|failed\|Error\|assert" > $O/cloth.txt; cat $O/cloth.txt | cut -c1-220
timeout -k 10 300 python tools/phase_overhead.py > $O/phase_overhead.txt 2>&1; tail -4 $O/phase_overhead.txt
