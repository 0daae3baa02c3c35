#!/bin/bash
# round 3, GPU session V: sphere pre-test in the contact band test (A/B), the ride-along test, rocprofv3 kernel stats of the bench command (where the re-sort's time goes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
bash tools/ab3.sh $O nosphere=libsoftmac_hip_nosphere.so sphere=libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_fused_backward.py tests/test_gpu_parity.py tests/test_gpu_pour.py -x -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc $?"; grep "SMAC_\|passed\|failed" $O/tests.log | cut -c1-400
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 2 > $O/bench_prof.json 2> $O/bench_prof.err
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -40 $O/kernel_stats.csv | cut -c1-200
