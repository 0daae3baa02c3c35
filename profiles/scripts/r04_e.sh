#!/bin/bash
# round 4, session E: voxeliser with trimesh's sign rule (door cache), import-level drop-in, losses / env after the ScalarField change
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
make -C oracle -s
timeout -k 10 900 python3 -m pytest tests/test_voxelize.py tests/test_gpu_pour.py tests/test_compat_imports.py tests/test_gpu_env.py tests/test_losses.py tests/test_gpu_windowed.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -15 $O/pytest.log
