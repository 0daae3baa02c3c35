#!/bin/bash
# round 3, GPU session W: ranks kept across re-sorts + restore-ahead: the sweep / re-binning tests, then rocprofv3 kernel stats of the bench command (where the re-sort's time goes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_backward.py tests/test_gpu_parity.py tests/test_gpu_pour.py tests/test_gpu_fullsize.py tests/test_gpu_long_rollout.py tests/test_gpu_slab_lib.py -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc $?"; grep "SMAC_\|passed\|failed\|FAILED" $O/tests.log | cut -c1-400
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 2 > $O/bench_prof.json 2> $O/bench_prof.err
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -45 $O/kernel_stats.csv | cut -c1-220
