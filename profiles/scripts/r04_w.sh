#!/bin/bash
# round 4, session W: the fused backward kernel at a TRUE 4 workgroups per CU - 21-slot stash (33.6 KB of LDS) and 128 VGPRs (148 B of scratch) - never measured together before
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; mkdir -p $O
SMAC_LIB=$PWD/softmac_amd/lib/libsoftmac_hip_s21o4.so timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_backward.py tests/test_gpu_parity.py -x -q -k "batched or fused or materials or grip_fixture" > $O/pytest.log 2>&1
rc=$?; echo "pytest (s21o4) rc $rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for round in 1 2; do
  for v in base s21 s21o4; do
    lib=libsoftmac_hip.so; [ $v != base ] && lib=libsoftmac_hip_$v.so
    SMAC_LIB=$PWD/softmac_amd/lib/$lib timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 3 > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || exit 1
    python3 -c "
import json;d=json.loads([l for l in open('$O/bench_${v}_$round.json') if l.startswith('{')][-1]); print('$v round $round', round(d['value'],1), d['ms_per_step_all'], 'bwd', round(d['bwd_only']['ms_per_step']*1e3,1), {k: round(x*1e3,1) for k,x in d['kernels_ms_per_step'].items() if k in ('p2g_g2p_grad','p2g_grad','g2p_grad')}, 'launch', round(d['roofline']['avg_launch_ms']*1e3,1))"
  done
done
