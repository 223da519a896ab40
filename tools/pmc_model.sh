#!/bin/bash
# SQ counter passes over the fused model kernel (one group per run; stops at the first failure).
set -u
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_model
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- \
      python3 /root/repo/bench.py --workload model32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
  rc=$?
  echo "pass $i exit $rc"
  if [ $rc -ne 0 ]; then exit $rc; fi
done
