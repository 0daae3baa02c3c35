#!/bin/bash
# round 2, GPU session AN: a 200-substep episode of the benchmark scene, fused backward step vs apart (tools/long_episode.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02an; mkdir -p $O
timeout -k 10 500 python tools/long_episode.py 200 > $O/long_episode.txt 2>&1; tail -6 $O/long_episode.txt | cut -c1-500
