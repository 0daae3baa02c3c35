#!/bin/bash
# round 5, final GPU session part 1 (end state: contact kernels with loads asked for together, filed tangents, launch width from the host-visible hit counts, flush with a node's
# words in neighbouring lanes): whole GPU suite with durations, smoke, PMC traffic passes (refreshes profiles/traffic_latest.json with the kernel-source hash),
# rocprofv3 --kernel-trace --stats of the driver's bench command, SQ counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z1; mkdir -p $O
make -C oracle -s
timeout -k 10 800 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -22 $O/gpu_suite.txt | cut -c1-200
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -3 $O/smoke.log | cut -c1-200
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $B --repeats 2 > $O/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $B --repeats 2 > $O/write.log 2>&1 &&
python3 tools/pmc_traffic.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/traffic_latest.json "r05 session Z1" > $O/pmc_traffic.csv && cat $O/pmc_traffic.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B > $O/stats.log 2>&1; find $O/stats -name "*kernel_stats.csv" | head -1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -o sq -- $B --repeats 2 > $O/sq.log 2>&1
python3 tools/sq_summary.py $(find $O/sq -name "*counter_collection.csv" | head -1) > $O/sq_summary.csv; head -8 $O/sq_summary.csv | cut -c1-220
