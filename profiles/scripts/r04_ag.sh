#!/bin/bash
# round 4, session AG: what k_contact_hits waits for - timing-only builds: x1 = without its global atomics (corrections dropped), x2 = also without collide_mixed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ag; mkdir -p $O
for v in base x1 x2; do
  lib=libsoftmac_hip.so; [ $v != base ] && lib=libsoftmac_hip_$v.so
  SMAC_LIB=$PWD/softmac_amd/lib/$lib timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  python3 -c "
import json;d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); print('$v', round(d['value'],1), {k: round(x*1e3,1) for k,x in d['kernels_ms'].items() if k in ('contact','contact_grad','grid_op','g2p_p2g')})"
done
