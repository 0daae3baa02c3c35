#!/bin/bash
# round 3, GPU session E: the in-library RCCL slab path on one GPU (world-1 self exchange), the env loop after its host-side diet, the slab + env suites
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_slab_lib.py -x -q -m gpu > $O/slab_lib.log 2>&1 || { tail -60 $O/slab_lib.log; exit 1; }
tail -3 $O/slab_lib.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_env.py tests/test_slabs.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "import json;d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]);print(d['value'],d['ms_per_step_all']);print(d.get('env_loop'))"
