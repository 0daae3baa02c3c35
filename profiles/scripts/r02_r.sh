#!/bin/bash
# round 2, GPU session R: full-size cloth test + cloth timing; per-chunk vs persistent pipelined k_g2p over 5 interleaved processes each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
make -C oracle -s
timeout -k 10 400 python -m pytest tests/test_gpu_cloth.py -m gpu -q -k full_size > $O/pytest_cloth_full.log 2>&1; tail -4 $O/pytest_cloth_full.log | cut -c1-300
timeout -k 10 300 python tools/bench_cloth.py > $O/bench_cloth.json 2> $O/bench_cloth.err; cat $O/bench_cloth.json | cut -c1-900
bash tools/ab_runs.sh $O 5 base=libsoftmac_hip.so pipe96=libsoftmac_hip.so,SMAC_G2P_PIPE=96 pipe160=libsoftmac_hip.so,SMAC_G2P_PIPE=160 2>&1 | tail -4
