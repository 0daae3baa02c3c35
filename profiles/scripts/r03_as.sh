#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03as; mkdir -p $O
SMAC_LIB=$PWD/softmac_amd/lib/libsoftmac_hip_phase.so SMAC_PHASE_DUMP=$PWD/$O/phase.txt timeout -k 10 300 python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 2 > $O/bench_phase.json 2> $O/bench_phase.err
python3 tools/phase_clock.py $O/phase.txt | tee $O/phase_report.txt
