#!/bin/bash
# round 4, session AD: variants of the fused forward kernel - base (121 VGPRs, 4 waves / SIMD), occ5 (launch bound 5: 96 VGPRs + 96 B scratch),
# roll5 (G2P gather with x-planes as a real loop: 96 VGPRs, no scratch, 5 waves / SIMD); and the re-sort interval 48 on base
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ad; mkdir -p $O
SMAC_LIB=$PWD/softmac_amd/lib/libsoftmac_hip_roll5.so timeout -k 10 600 python3 -m pytest tests/test_gpu_fused_forward.py tests/test_gpu_parity.py -x -q -k "fused or batched or grip_fixture" > $O/pytest.log 2>&1
rc=$?; echo "pytest (roll5) rc $rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $O/pytest.log | head -30; exit $rc; }
for round in 1 2; do
  for v in base occ5 roll5 base48; do
    lib=libsoftmac_hip.so; extra=""
    [ $v = occ5 ] && lib=libsoftmac_hip_occ5.so
    [ $v = roll5 ] && lib=libsoftmac_hip_roll5.so
    [ $v = base48 ] && extra="--sort-interval 48"
    [ $v = base48 ] && [ $round = 2 ] && continue
    SMAC_LIB=$PWD/softmac_amd/lib/$lib timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 $extra --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || { tail -5 $O/bench_${v}_$round.err; exit 1; }
    python3 -c "
import json;d=json.loads([l for l in open('$O/bench_${v}_$round.json') if l.startswith('{')][-1]); print('$v round $round', round(d['value'],1), 'fwd', round(d['fwd_only']['ms_per_step']*1e3,1), 'bwd', round(d['bwd_only']['ms_per_step']*1e3,1), {k: round(x*1e3,1) for k,x in d['kernels_ms'].items() if k in ('g2p_p2g','g2p','p2g','grid_op','sort','p2g_g2p_grad')})"
  done
done
