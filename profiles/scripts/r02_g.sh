#!/bin/bash
# round 2, GPU session G: timing after the register / integer-position work, migration tests on the HIP engine, whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
make -C oracle -s
run() { local label=$1 lib=$2; shift 2
  env SMAC_LIB=$PWD/softmac_amd/lib/$lib "$@" timeout -k 10 300 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-f64 --repeats 3 2>$O/err_$label.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$label', round(d['value'],1), 'ms/step', d['ms_per_step_all'], {a:k.get(a) for a in ('p2g','grid_op','contact','g2p','grid_checkpoint','g2p_grad','reduce_agvout','contact_grad','grid_op_grad','p2g_grad','sort')})"
}
run base libsoftmac_hip.so
run base_again libsoftmac_hip.so
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; grep -E "passed|failed|^FAILED|^\[float|^\[long|AssertionError:|Error" $O/pytest.log | cut -c1-400 | head -40
