#!/bin/bash
# round 3, GPU session AQ: the multi-rank bench path after this round's launch changes - 2 ranks over gloo sharing the one GPU (Python SlabRunner: the
# in-library RCCL loop cannot put two ranks on one device), strong and weak; the slab-decomposed result against the single-GPU one is tests/test_slabs.py + test_gpu_slab_lib.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aq; mkdir -p $O
SMAC_DIST_BACKEND=gloo SMAC_FORCE_DEVICE=0 timeout -k 10 400 python3 bench.py --gpus 2 --slab-runner python --steps 10 --warmup 4 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 1 > $O/bench_n2_strong.json 2> $O/bench_n2_strong.err; echo "strong rc $?"; cut -c1-600 $O/bench_n2_strong.json; tail -2 $O/bench_n2_strong.err | cut -c1-300
SMAC_DIST_BACKEND=gloo SMAC_FORCE_DEVICE=0 timeout -k 10 400 python3 bench.py --gpus 2 --slab-runner python --scaling weak --steps 10 --warmup 4 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 1 > $O/bench_n2_weak.json 2> $O/bench_n2_weak.err; echo "weak rc $?"; cut -c1-400 $O/bench_n2_weak.json; tail -2 $O/bench_n2_weak.err | cut -c1-300
