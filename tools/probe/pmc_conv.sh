#!/bin/bash
# PMC passes over the conv kernels (one counter group per run, as the pool requires).
set -u
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_conv
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 /root/repo/tools/probe/conv_once.py > $OUT/p$i.log 2>&1
  echo "pass $i ($grp) exit $?"
done
