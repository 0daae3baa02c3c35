#!/bin/bash
# round 2, GPU session D: precise-FP constitutive code + f64 tiles for sparse chunks: probes (fixed scenes + failing fuzz cases), full suite, A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
make -C oracle -s
timeout -k 10 300 python tools/prec_probe.py --precision float32 --out $O/prec_f32.json > $O/prec_f32.log 2>&1
grep -v "^ " $O/prec_f32.log
timeout -k 10 300 python tools/prec_probe.py --precision float32 --fuzz 3,5,13,15,17,27 --out $O/prec_fuzz.json > $O/prec_fuzz.log 2>&1
cat $O/prec_fuzz.log
bash tools/ab.sh softmac_amd/lib/libsoftmac_hip.so softmac_amd/lib/libsoftmac_hip_p2g5.so softmac_amd/lib/libsoftmac_hip.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; grep -E "passed|failed|^FAILED|^\[float|AssertionError:" $O/pytest.log | cut -c1-400 | head -60
