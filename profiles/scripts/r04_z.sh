#!/bin/bash
# round 4, session Z: first pass over fresh frames vs a repeat of the same frames (tools/window_probe.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O
timeout -k 10 300 python3 tools/window_probe.py --steps 20 --sort-interval 40 > $O/probe_20_40.txt 2> $O/probe.err || { tail -5 $O/probe.err; exit 1; }
cat $O/probe_20_40.txt
timeout -k 10 300 python3 tools/window_probe.py --steps 20 --sort-interval 20 > $O/probe_20_20.txt 2>> $O/probe.err || { tail -5 $O/probe.err; exit 1; }
cat $O/probe_20_20.txt
