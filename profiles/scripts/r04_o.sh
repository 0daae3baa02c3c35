#!/bin/bash
# round 4, session O: the whole GPU suite as the driver runs it, then the driver-style bench line on the frozen sources (roofline.traffic filled from profiles/traffic_latest.json)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O
make -C oracle -s
timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -6 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; python3 -c "
import json;d=json.loads([l for l in open('$O/bench_driver_style.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step_all'], d['roofline'])"
