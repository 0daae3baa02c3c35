#!/bin/bash
# round 2, GPU session Y: final state - whole GPU suite, driver-style bench line (with f64, cloth and CPU sub-records), rocprofv3 stats + PMC passes of the final code
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02y; mkdir -p $O
make -C oracle -s
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; cut -c1-600 $O/bench_driver_style.json; tail -2 $O/bench_driver_style.err | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-f64 --no-cloth --repeats 1 > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f64 --no-cloth --repeats 1 > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f64 --no-cloth --repeats 1 > $O/write.log 2>&1
find $O -name "*_kernel_stats.csv" | head -2; find $O -name "*counter_collection.csv" | head -4
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
