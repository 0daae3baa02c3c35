#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
make -C oracle -s
timeout -k 10 300 python tools/cloth_debug.py > $O/debug.log 2>&1; tail -12 $O/debug.log | cut -c1-300
