#!/bin/bash
# round 2, GPU session U: the multi-rank bench path (2 ranks over gloo sharing the one GPU: strong and weak), occupancy variants of the two gather kernels
# over 4 interleaved processes each, cloth timing at the C5-shaped slice (2M particles, 256^3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u; mkdir -p $O
make -C oracle -s
SMAC_DIST_BACKEND=gloo SMAC_FORCE_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 4 --no-cpu-baseline --no-f64 --repeats 1 > $O/bench_n2_strong.json 2> $O/bench_n2_strong.err; cut -c1-700 $O/bench_n2_strong.json; tail -2 $O/bench_n2_strong.err | cut -c1-300
SMAC_DIST_BACKEND=gloo SMAC_FORCE_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --scaling weak --steps 10 --warmup 4 --no-cpu-baseline --no-f64 --repeats 1 > $O/bench_n2_weak.json 2> $O/bench_n2_weak.err; cut -c1-400 $O/bench_n2_weak.json; tail -2 $O/bench_n2_weak.err | cut -c1-300
timeout -k 10 300 python tools/bench_cloth.py --particles 2097152 --grid 256 > $O/bench_cloth_c5slice.json 2> $O/bench_cloth_c5slice.err; cat $O/bench_cloth_c5slice.json | cut -c1-900; tail -2 $O/bench_cloth_c5slice.err | cut -c1-300
bash tools/ab_runs.sh $O 4 base=libsoftmac_hip.so g2pg4=libsoftmac_hip_g2pg4.so g2p5=libsoftmac_hip_g2p5.so 2>&1 | tail -4
