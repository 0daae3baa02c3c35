#!/bin/bash
# round 2, GPU session I: checkpoint modes, k_p2g_grad variants, AoSoA-1024 layout; parity suite on the tiled layout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i
make -C oracle -s
bash tools/ab3.sh $O base=libsoftmac_hip.so ck0=libsoftmac_hip.so,SMAC_CK_MODE=0 ck1=libsoftmac_hip.so,SMAC_CK_MODE=1 ns=libsoftmac_hip_ns.so ns4=libsoftmac_hip_ns4.so t1024=libsoftmac_hip_t1024.so 2>&1 | tail -8
SMAC_LIB=$PWD/softmac_amd/lib/libsoftmac_hip_t1024.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_env.py tests/test_slabs.py tests/test_losses.py tests/test_gpu_long_rollout.py -m gpu -q -x > $O/pytest_t1024.log 2>&1; tail -4 $O/pytest_t1024.log | cut -c1-300
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_slabs.py tests/test_gpu_pour.py "tests/test_gpu_fullsize.py::test_c4_slice_4m_particles_256_grid_vs_cpu_port" -m gpu -q -x > $O/pytest_base.log 2>&1; tail -4 $O/pytest_base.log | cut -c1-300
