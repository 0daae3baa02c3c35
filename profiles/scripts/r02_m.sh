#!/bin/bash
# round 2, GPU session M: which counters exist (for the fast/slow mode study of k_g2p), + a first pair of TCC passes in 4 processes each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1 || rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -c . $O/avail.txt
for i in 1 2 3 4; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/tccA_$i -o t -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f64 --repeats 1 > $O/tccA_$i.log 2>&1 || echo "pass A $i failed"
done
for i in 1 2 3 4; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_WRITEBACK_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $O/tccB_$i -o t -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f64 --repeats 1 > $O/tccB_$i.log 2>&1 || echo "pass B $i failed"
done
find $O -name "*counter_collection.csv" | wc -l
