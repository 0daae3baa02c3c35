#!/bin/bash
# round 2, GPU session AQ: SQ counters of the final kernels (the counter set of profiles/scripts/r02_j.sh, which this hardware collects in one pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02aq; mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -o sq -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f64 --no-cloth --repeats 1 > $O/sq.log 2>&1
python3 tools/sq_summary.py $O/sq/sq_counter_collection.csv > $O/sq_summary.csv && head -12 $O/sq_summary.csv | cut -c1-220
