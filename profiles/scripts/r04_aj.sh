#!/bin/bash
# round 4, session AJ: hit entries carry the particle's position, the contact adjoint takes the list's length by value - parity, then A/B against the previous commit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04aj; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py tests/test_gpu_env.py tests/test_gpu_fused_backward.py -x -q > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $O/pytest.log | head -30; exit $rc; }
for round in 1 2 3; do
  for v in prev new; do
    lib=libsoftmac_hip.so; [ $v = prev ] && lib=libsoftmac_hip_prev.so
    SMAC_LIB=$PWD/softmac_amd/lib/$lib timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || { tail -5 $O/bench_${v}_$round.err; exit 1; }
    python3 -c "
import json;d=json.loads([l for l in open('$O/bench_${v}_$round.json') if l.startswith('{')][-1]); print('$v round $round', round(d['value'],1), {k: round(x*1e3,1) for k,x in d['kernels_ms'].items() if k in ('contact','contact_grad','g2p_p2g','p2g_g2p_grad')})"
  done
done
