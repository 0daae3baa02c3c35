#!/bin/bash
# A/B several builds of libsoftmac_hip.so in one GPU session: tools/ab.sh lib1.so lib2.so ...
for lib in "$@"; do
  SMAC_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 64 --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$lib', round(d['value'],1), 'dev_ms', round(d['device_ms_per_step'],4), {a:k[a] for a in ('p2g','g2p','g2p_grad','p2g_grad','contact','contact_grad','grid_checkpoint','sort')})"
done
