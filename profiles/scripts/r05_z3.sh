#!/bin/bash
# round 5, final GPU session part 3: `bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, 127.0.0.1), two ranks on the ONE GPU of the box:
# (a) over the IPC test transport (SMAC_COMM_STUB=2): the in-library slab loop between two processes - a functional record of the N > 1 path on the final sources, not a scaling number;
# (b) without the stub: RCCL refuses two ranks on one device, every rank agrees on the fallback and the line says so.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z3; mkdir -p $O
L="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline"
SMAC_FORCE_DEVICE=0 SMAC_COMM_STUB=2 timeout -k 10 500 $L > $O/bench_n2_lib_ipc.json 2> $O/bench_n2_lib_ipc.err; echo "ipc rc=$?"
python3 -c "
import json;d=json.loads([l for l in open('$O/bench_n2_lib_ipc.json') if l.startswith('{')][-1]); print(d['value'], d['n_gpus'], d['scaling'], d['config'].get('transport'), d['config'].get('slab_runner'), d['ms_per_step_all'])"
SMAC_FORCE_DEVICE=0 timeout -k 10 500 ${L/29541/29542} > $O/bench_n2_fallback.json 2> $O/bench_n2_fallback.err; echo "fallback rc=$?"
python3 -c "
import json;d=json.loads([l for l in open('$O/bench_n2_fallback.json') if l.startswith('{')][-1]); print(d['value'], d['n_gpus'], d['config'].get('transport'), d['config'].get('slab_runner'), str(d['config'].get('fallback'))[:200])"
for e in $O/*.err; do tail -n 2 $e | cut -c1-300; done
