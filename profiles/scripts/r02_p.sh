#!/bin/bash
# round 2, GPU session P: fast/slow-mode study of k_g2p - TLB, L2 and write-path counters over several processes (each counter set is first tried
# on a trivial program: a set the hardware cannot collect makes rocprofv3 abort and hang until its timeout)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 tools/microbench/xcc_probe.hip -o /tmp/xcc_probe 2> /dev/null
declare -A SETS
SETS[tlb]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_PENDING_STALL_CYCLES_sum"
SETS[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum"
SETS[wr]="TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum"
for name in tlb tcc wr; do
  if timeout -k 5 40 rocprofv3 --kernel-trace --pmc ${SETS[$name]} --output-format csv -d $O/probe_$name -o t -- /tmp/xcc_probe > $O/probe_$name.log 2>&1; then
    echo "set $name: ok"
    for i in 1 2 3 4 5 6; do
      timeout -k 5 110 rocprofv3 --kernel-trace --pmc ${SETS[$name]} --output-format csv -d $O/${name}_$i -o t -- python3 tools/mode_pmc.py > $O/${name}_$i.log 2>&1 || echo "$name $i failed"
      grep g2p_us $O/${name}_$i.log
    done
  else
    echo "set $name: not collectable"
  fi
done
