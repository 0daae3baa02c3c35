#!/bin/bash
# round 2, GPU session F: perf A/B (pipelined persistent g2p, occupancy variants, precise-FP variant), layout microbenchmark, ctype1 re-test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
make -C oracle -s
./tools/microbench/frame_layout > $O/frame_layout.txt 2>&1; cat $O/frame_layout.txt
run() { # label, lib, env...
  local label=$1 lib=$2; shift 2
  env SMAC_LIB=$PWD/softmac_amd/lib/$lib "$@" timeout -k 10 300 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-f64 --repeats 3 2>$O/err_$label.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$label', round(d['value'],1), 'ms/step', d['ms_per_step_all'], {a:k.get(a) for a in ('p2g','grid_op','contact','g2p','grid_checkpoint','g2p_grad','reduce_agvout','contact_grad','grid_op_grad','p2g_grad','sort')})"
}
run base libsoftmac_hip.so
run pipe128 libsoftmac_hip.so SMAC_G2P_PIPE=128
run pipe96 libsoftmac_hip.so SMAC_G2P_PIPE=96
run pipe64 libsoftmac_hip.so SMAC_G2P_PIPE=64
run p2gg2 libsoftmac_hip_p2gg2.so
run pfp libsoftmac_hip_pfp.so
run base2 libsoftmac_hip.so
SMAC_LIB=$PWD/softmac_amd/lib/libsoftmac_hip_pfp.so timeout -k 10 300 python tools/prec_probe.py --precision float32 --only grip2k --out $O/prec_pfp.json > $O/prec_pfp.log 2>&1; grep -v "^ " $O/prec_pfp.log | cut -c1-420
timeout -k 10 300 python tools/prec_probe.py --precision float32 --only grip2k --out $O/prec_base.json > $O/prec_base.log 2>&1; grep -v "^ " $O/prec_base.log | cut -c1-420
SMAC_G2P_PIPE=96 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_pipe.log 2>&1; tail -3 $O/pytest_pipe.log
