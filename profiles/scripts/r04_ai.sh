#!/bin/bash
# round 4, session AI: bench.py --gpus 2 on ONE GPU with the new window scheme - the in-library slab loop over the IPC test transport without and with a
# migration inside the timed windows, and the Python SlabRunner over gloo (functional records, not scaling numbers)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ai; mkdir -p $O
for mig in 0 8; do
  SMAC_FORCE_DEVICE=0 SMAC_COMM_STUB=2 timeout -k 10 500 python3 bench.py --gpus 2 --slab-runner lib --steps 16 --warmup 4 --repeats 2 --sort-interval 16 --migrate-every $mig --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_n2_ipc_mig$mig.json 2> $O/bench_n2_ipc_mig$mig.err
  echo "bench lib mig=$mig rc $?"; tail -2 $O/bench_n2_ipc_mig$mig.err
  python3 -c "
import json;d=json.loads([l for l in open('$O/bench_n2_ipc_mig$mig.json') if l.startswith('{')][-1]); print({k:d.get(k) for k in ('value','n_gpus','scaling','transport','migrations_in_window','particles_migrated','slab_runner','ms_per_step_all')})"
done
SMAC_FORCE_DEVICE=0 SMAC_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --slab-runner python --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n2_python_gloo.json 2> $O/bench_n2_python_gloo.err
echo "bench python runner rc $?"; tail -2 $O/bench_n2_python_gloo.err
python3 -c "
import json;d=json.loads([l for l in open('$O/bench_n2_python_gloo.json') if l.startswith('{')][-1]); print({k:d.get(k) for k in ('value','n_gpus','scaling','slab_runner','ms_per_step_all','repeats')}, d['config']['resorts_in_windows'], d['config']['particles_per_gpu'])"
