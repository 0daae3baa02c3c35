#!/bin/bash
# round 2, GPU session C: tightened f32 tolerances over the whole GPU suite; fast-math vs none on the precision probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
make -C oracle -s
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; grep -E "passed|failed|^FAILED|^\[float|Error" $O/pytest.log | head -60
SMAC_LIB=$PWD/softmac_amd/lib/libsoftmac_hip_nofast.so timeout -k 10 300 python tools/prec_probe.py --precision float32 --out $O/prec_f32_nofast.json > $O/prec_f32_nofast.log 2>&1
grep -v "^ " $O/prec_f32_nofast.log
bash tools/ab.sh softmac_amd/lib/libsoftmac_hip.so softmac_amd/lib/libsoftmac_hip_nofast.so > $O/ab.txt 2>&1; cat $O/ab.txt
