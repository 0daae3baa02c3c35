#!/bin/bash
# round 3, GPU session AA: cost of a windowed episode against the resident one at the benchmark size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aa; mkdir -p $O
timeout -k 10 500 python3 tools/windowed_cost.py 200 50 > $O/windowed_cost.txt 2> $O/err.txt; echo "rc $?"; cat $O/windowed_cost.txt; tail -3 $O/err.txt | cut -c1-300
