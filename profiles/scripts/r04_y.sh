#!/bin/bash
# round 4, session Y: wide tiles with the fixed costs cut (both staged records of a thread in flight together, 16-byte zeroing, shell enumerated) - wide2 -
# against the first wide build (wide1) and the 6^3 tiles (base)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_fused_backward.py tests/test_gpu_resort.py tests/test_gpu_parity.py -x -q -k "fused or ride or resort or rebin or batched or grip_fixture or drift" > $O/pytest.log 2>&1
rc=$?; echo "pytest (wide2) rc $rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for round in 1 2; do
  for v in base wide1 wide2; do
    lib=libsoftmac_hip.so; [ $v != wide2 ] && lib=libsoftmac_hip_$v.so
    for cfg in "20 20" "40 40"; do
      set -- $cfg
      SMAC_LIB=$PWD/softmac_amd/lib/$lib timeout -k 10 200 python3 bench.py --steps $1 --sort-interval $2 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 3 > $O/bench_${v}_$1_$round.json 2> $O/bench_${v}_$1_$round.err || exit 1
      python3 -c "
import json;d=json.loads([l for l in open('$O/bench_${v}_$1_$round.json') if l.startswith('{')][-1]); print('$v steps $1 round $round', round(d['value'],1), d['ms_per_step_all'], 'fwd', round(d['fwd_only']['ms_per_step']*1e3,1), 'bwd', round(d['bwd_only']['ms_per_step']*1e3,1), {k: round(x*1e3,1) for k,x in d['kernels_ms_per_step'].items() if k in ('p2g','g2p','p2g_g2p_grad','sort')}, 'launch', round(d['roofline']['avg_launch_ms']*1e3,1))"
    done
  done
done
