#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bd; mkdir -p $O
timeout -k 10 600 python3 tools/windowed_cost.py 600 50 > $O/windowed_cost_600.txt 2> $O/err.txt; echo "rc $?"; cat $O/windowed_cost_600.txt; tail -2 $O/err.txt | cut -c1-200
