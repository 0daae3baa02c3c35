#!/bin/bash
# libsoftmac_hip_prev.so = the library of the last commit (git HEAD), for an A/B against the working tree (SMAC_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/softmac_amd/csrc $T/include
for f in $(git -C $ROOT ls-tree --name-only HEAD softmac_amd/csrc/); do git -C $ROOT show HEAD:$f > $T/$f; done
git -C $ROOT show HEAD:include/softmac_hip.h > $T/include/softmac_hip.h
(cd $T && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=fast -fno-slp-vectorize -ffast-math -fno-finite-math-only -Wno-unused-value -shared -fPIC -o $ROOT/softmac_amd/lib/libsoftmac_hip_prev.so softmac_amd/csrc/softmac_hip.hip)
rm -rf $T
