#!/bin/bash
# round 3, GPU session F: binning at reset + single-sweep TaichiEnv.backward (env_loop), S-grip from the reference's palm cache and the voxelised
# finger.obj, tiered tolerances in the slab / rebinning tests (their measured errors are printed), host cost of the slab loops
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_slabs.py tests/test_gpu_env.py tests/test_gpu_parity.py tests/test_gpu_slab_lib.py tests/test_bench_launch.py -x -q -m gpu -s > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
grep -E "^\[|passed|failed" $O/tests.log | cut -c1-1200
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "rebinning" > $O/fullsize.log 2>&1 || { tail -60 $O/fullsize.log; exit 1; }
grep -E "^\[|passed|failed" $O/fullsize.log | cut -c1-400
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "import json;d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]);print(d['value'],d['ms_per_step_all'],d['kernels_ms']);print(d.get('env_loop'))"
timeout -k 10 600 python3 tools/exchange_overhead.py > $O/exchange_overhead.txt 2>&1 || { tail -20 $O/exchange_overhead.txt; exit 1; }
cat $O/exchange_overhead.txt
