#!/bin/bash
# round 2, GPU session B: fixed-point positions + f64 contact chain: parity suite, precision probes, A/B timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
make -C oracle -s
timeout -k 10 300 python tools/prec_probe.py --precision float32 --out $O/prec_f32_default.json > $O/prec_f32_default.log 2>&1
SMAC_LIB=$PWD/softmac_amd/lib/libsoftmac_hip_cf64.so timeout -k 10 300 python tools/prec_probe.py --precision float32 --out $O/prec_f32_cf64.json > $O/prec_f32_cf64.log 2>&1
timeout -k 10 300 python tools/prec_probe.py --precision float64 --out $O/prec_f64.json > $O/prec_f64.log 2>&1
grep -v "^ " $O/prec_f32_default.log; grep -v "^ " $O/prec_f32_cf64.log; grep -v "^ " $O/prec_f64.log
bash tools/ab.sh softmac_amd/lib/libsoftmac_hip.so softmac_amd/lib/libsoftmac_hip_cf64.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
