#!/bin/bash
# round 2, GPU session N: cloth contact tests; fast/slow-mode study of k_g2p with TLB / L2 / write-path counters over several processes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
make -C oracle -s
timeout -k 10 900 python -m pytest tests/test_gpu_cloth.py -m gpu -q -x > $O/pytest_cloth.log 2>&1; tail -5 $O/pytest_cloth.log | cut -c1-400
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/tlb_$i -o t -- python3 tools/mode_pmc.py > $O/tlb_$i.log 2>&1 || echo "tlb $i failed"
  grep g2p_us $O/tlb_$i.log
done
for i in 1 2 3 4 5 6; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum --output-format csv -d $O/tcc_$i -o t -- python3 tools/mode_pmc.py > $O/tcc_$i.log 2>&1 || echo "tcc $i failed"
  grep g2p_us $O/tcc_$i.log
done
for i in 1 2 3 4 5 6; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum --output-format csv -d $O/wr_$i -o t -- python3 tools/mode_pmc.py > $O/wr_$i.log 2>&1 || echo "wr $i failed"
  grep g2p_us $O/wr_$i.log
done
