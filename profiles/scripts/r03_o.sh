#!/bin/bash
# round 3, GPU session O: (1) the overflow test's errors on the build before / after the combined SDF taps, (2) phase clock of the particle kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
for lib in libsoftmac_hip_notaps.so libsoftmac_hip.so; do
  SMAC_LIB=$PWD/softmac_amd/lib/$lib SMAC_PRINT_ERRS=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "overflow" > $O/overflow_$lib.log 2>&1
  echo "$lib rc $?"; grep "ERRS" $O/overflow_$lib.log | cut -c1-400
done
SMAC_LIB=$PWD/softmac_amd/lib/libsoftmac_hip_phase.so SMAC_PHASE_DUMP=$PWD/$O/phase.txt timeout -k 10 300 python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 2 > $O/bench_phase.json 2> $O/bench_phase.err
python3 tools/phase_clock.py $O/phase.txt | tee $O/phase_report.txt
