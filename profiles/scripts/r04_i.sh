#!/bin/bash
# round 4, session I: bench.py --gpus 2 on ONE GPU over the IPC test transport, with and without a migration inside the timed window (functional evidence,
# not a scaling number), + the drift-repair test with particle actions
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
make -C oracle -s
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "drift_repair or outrun" > $O/pytest_drift.log 2>&1
echo "pytest drift rc $?"; tail -3 $O/pytest_drift.log
for mig in 0 8; do
  SMAC_FORCE_DEVICE=0 SMAC_COMM_STUB=2 timeout -k 10 500 python3 bench.py --gpus 2 --slab-runner lib --steps 16 --warmup 4 --repeats 2 --migrate-every $mig --no-cpu-baseline --no-f64 --no-cloth --no-env-loop > $O/bench_n2_ipc_mig$mig.json 2> $O/bench_n2_ipc_mig$mig.err
  echo "bench mig=$mig rc $?"; tail -2 $O/bench_n2_ipc_mig$mig.err
  python3 -c "
import json;d=json.loads([l for l in open('$O/bench_n2_ipc_mig$mig.json') if l.startswith('{')][-1]); print({k:d.get(k) for k in ('value','n_gpus','scaling','transport','migrations_in_window','particles_migrated','slab_runner','ms_per_step_all')})"
done
