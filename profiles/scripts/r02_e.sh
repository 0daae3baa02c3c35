#!/bin/bash
# round 2, GPU session E: ctype probes, new tests (long rollout, pour scene C1, equilibrium pin, launcher), layout microbenchmark
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
make -C oracle -s
timeout -k 10 300 python tools/prec_probe.py --precision float32 --only ctype --out $O/prec_ctype.json > $O/prec_ctype.log 2>&1; cat $O/prec_ctype.log | cut -c1-700
./tools/microbench/frame_layout > $O/frame_layout.txt 2>&1; cat $O/frame_layout.txt
timeout -k 10 1000 python -m pytest tests/test_gpu_long_rollout.py tests/test_gpu_pour.py tests/test_equilibrium.py tests/test_bench_launch.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py "tests/test_gpu_fullsize.py::test_fullsize_rebinning_invariance" -m gpu -q -s > $O/pytest.log 2>&1; grep -E "passed|failed|^FAILED|^\[float|^\[long|AssertionError:|Error" $O/pytest.log | cut -c1-400 | head -60
