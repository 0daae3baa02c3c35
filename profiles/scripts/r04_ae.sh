#!/bin/bash
# round 4, session AE: grid_v_mixed no longer stored / filed / restored on the whole-substep path (recomputed from {m,p} by the contact kernels), contact
# kernels' grids fitted to the known hit counts (SMAC_CONTACT_FIT) - parity, then A/B against the previous commit's library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ae; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_forward.py tests/test_gpu_fused_backward.py tests/test_gpu_mixed.py tests/test_gpu_slab_lib.py tests/test_slabs.py -m gpu -x -q > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $O/pytest.log | head -30; exit $rc; }
for round in 1 2; do
  for v in prev new nofit; do
    lib=libsoftmac_hip.so; fit=1
    [ $v = prev ] && lib=libsoftmac_hip_prev.so
    [ $v = nofit ] && fit=0
    SMAC_CONTACT_FIT=$fit SMAC_LIB=$PWD/softmac_amd/lib/$lib timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || { tail -5 $O/bench_${v}_$round.err; exit 1; }
    python3 -c "
import json;d=json.loads([l for l in open('$O/bench_${v}_$round.json') if l.startswith('{')][-1]); print('$v round $round', round(d['value'],1), 'fwd', round(d['fwd_only']['ms_per_step']*1e3,1), 'bwd', round(d['bwd_only']['ms_per_step']*1e3,1), {k: round(x*1e3,1) for k,x in d['kernels_ms'].items() if k in ('g2p_p2g','grid_op','contact','contact_grad','reduce_agvout','p2g_g2p_grad','grid_checkpoint')})"
  done
done
