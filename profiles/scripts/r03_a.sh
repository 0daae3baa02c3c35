#!/bin/bash
# round 3, GPU session A: (1) VALU issue-rate microbenchmark, (2) which SQ counters this rocprofv3 knows,
# (3) baseline bench line of the round's starting code, (4) SQ busy / VALU-cycle counters on the bench's kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_issue tools/microbench/valu_issue.hip 2> $O/valu_build.log || exit 1
timeout -k 10 120 /tmp/valu_issue > $O/valu_issue.txt 2>&1 || { echo "valu_issue failed"; tail -5 $O/valu_issue.txt; exit 1; }
cat $O/valu_issue.txt
timeout -k 10 60 rocprofv3 -L > $O/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/avail.txt | sort -u | tr '\n' ' ' > $O/sq_counters.txt; wc -w $O/sq_counters.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f64 --no-cloth > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "import json;d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]);print(d['value'],d['ms_per_step_all'],d['kernels_ms'])"
for set in "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VALU SQ_INSTS_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '+')
  if timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f64 --no-cloth --repeats 1 > $O/pmc_$tag.log 2>&1; then
    echo "set [$set] ok"
  else
    echo "set [$set] FAILED"; tail -3 $O/pmc_$tag.log
  fi
done
ls $O
