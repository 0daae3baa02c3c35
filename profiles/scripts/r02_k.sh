#!/bin/bash
# round 2, GPU session K: stream-overlap microbenchmark, the two GPU tests that had not run / failed in J
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k; mkdir -p $O
make -C oracle -s
hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_overlap.hip -o /tmp/stream_overlap && timeout -k 10 120 /tmp/stream_overlap > $O/stream_overlap.txt 2>&1; cat $O/stream_overlap.txt
timeout -k 10 900 python -m pytest tests/test_slabs.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-300
