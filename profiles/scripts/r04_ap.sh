#!/bin/bash
# round 4, session AP: the fused forward launch inside the library's slab loop (its pieces: G2P of substep f + P2G of substep f + 1 cross no exchange) -
# the slab tests (self loop, two ranks over IPC, migration), then tools/exchange_overhead.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ap; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_slab_lib.py tests/test_slabs.py tests/test_bench_launch.py -m gpu -x -q > $O/pytest.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $O/pytest.log | head -30; exit $rc; }
timeout -k 10 600 python3 tools/exchange_overhead.py c stub rccl > $O/exchange.txt 2> $O/exchange.err; grep -v "exchanges per pair" $O/exchange.txt | tail -8
SMAC_FUSED_FWD=0 timeout -k 10 600 python3 tools/exchange_overhead.py stub > $O/exchange_nofuse.txt 2>> $O/exchange.err; grep -v "exchanges per pair" $O/exchange_nofuse.txt | tail -4
