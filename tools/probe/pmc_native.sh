#!/bin/bash
# SQ counter pass over one native probe binary: pmc_native.sh <variant> <N> [counters...]
set -u
cd /tmp && export TMPDIR=/tmp
v=$1; n=$2; shift 2
ctrs="${*:-SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS}"
OUT=/root/repo/gpurun_out/pmc_native_$v
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 100 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT -- /root/repo/tools/probe/variants/$v $n > $OUT/run.log 2>&1
echo "exit $?"
