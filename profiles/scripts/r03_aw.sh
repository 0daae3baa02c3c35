#!/bin/bash
# round 3, GPU session AW: small scenes (the reference's own demo sizes: 5,000 ... 16,384 particles, 64^3) - is the batched loop GPU- or host-bound there?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aw; mkdir -p $O
for n in 5000 16384 65536; do
  timeout -k 10 200 python3 bench.py --workload s-elastic --particles $n --grid 64 --steps 200 --warmup 40 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop --repeats 3 > $O/bench_$n.json 2> $O/bench_$n.err
  python3 -c "
import json;d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1])
k=d['kernels_ms']; print($n, 'substeps/s', round(d['value']), 'wall ms/pair', round(d['ms_per_step'],4), 'device ms/pair', round(d['device_ms_per_step'],4), 'sum of kernel avgs us', round(1e3*sum(v for n_,v in k.items() if n_ not in ('sort','reorder_adjoint','p2g_grad','g2p_grad')),1), {a: round(b*1e3,1) for a,b in k.items()})"
done
