#!/bin/bash
# round 4, session AA: the new window scheme of bench.py (advancing windows, seed from HBM inside the window, mean of all windows) on the wide-tile library
# and on the 6^3-tile build of the same sources (SMAC_WIDE_TILE=0), intervals 20 / 40 / 64; the device-seed test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04aa; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_api_sequences.py -x -q -k "seed_from_device" > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { tail -30 $O/pytest.log; exit $rc; }
for round in 1 2; do
  for v in narrow wide; do
    lib=libsoftmac_hip.so; [ $v != wide ] && lib=libsoftmac_hip_$v.so
    for iv in 20 40 64; do
      [ $round = 2 ] && [ $iv = 64 ] && continue
      SMAC_LIB=$PWD/softmac_amd/lib/$lib timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --sort-interval $iv --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_${v}_${iv}_$round.json 2> $O/bench_${v}_${iv}_$round.err || { tail -5 $O/bench_${v}_${iv}_$round.err; exit 1; }
      python3 -c "
import json;d=json.loads([l for l in open('$O/bench_${v}_${iv}_$round.json') if l.startswith('{')][-1]); print('$v interval $iv round $round', round(d['value'],1), d['ms_per_step_all'], 'fwd', round(d['fwd_only']['ms_per_step']*1e3,1), 'bwd', round(d['bwd_only']['ms_per_step']*1e3,1), {k: round(x*1e3,1) for k,x in d['kernels_ms_per_step'].items() if k in ('p2g','g2p','p2g_g2p_grad','sort','g2p_grad','p2g_grad')}, 'launch', round(d['roofline']['avg_launch_ms']*1e3,1), d['config']['resorts_in_windows'])"
    done
  done
done
