#!/bin/bash
# round 4, session AH: the re-sort's host round trips - one write to pinned memory instead of four hipMemcpyAsync to pageable memory, lists and per-block
# tables emitted before the synchronisation, destination map written where the epoch keeps it - parity of everything that re-sorts, A/B against the previous commit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ah; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_resort.py tests/test_gpu_api_sequences.py tests/test_gpu_fused_forward.py tests/test_gpu_long_rollout.py tests/test_gpu_windowed.py -x -q > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $O/pytest.log | head -30; exit $rc; }
for round in 1 2 3; do
  for v in prev new; do
    lib=libsoftmac_hip.so; [ $v = prev ] && lib=libsoftmac_hip_prev.so
    SMAC_LIB=$PWD/softmac_amd/lib/$lib timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || { tail -5 $O/bench_${v}_$round.err; exit 1; }
    python3 -c "
import json;d=json.loads([l for l in open('$O/bench_${v}_$round.json') if l.startswith('{')][-1]); print('$v round $round', round(d['value'],1), d['ms_per_step_all'], 'sort', round(d['kernels_ms']['sort']*1e3,1), 'env_loop', round(d['env_loop']['value'],1))"
  done
done
