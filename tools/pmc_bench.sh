#!/bin/bash
# HBM-traffic PMC passes over the bench workloads: one counter group per run (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950), no trace domains besides --kernel-trace.  Stops at
# the first pass that fails.  Summarise with tools/pmc_summarise.py.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/${PMC_TAG:-pmc_bench}
mkdir -p $OUT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 170 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$grp -- \
      python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$grp.log 2>&1
  rc=$?
  echo "pass $i ($grp) exit $rc"
  if [ $rc -ne 0 ]; then exit $rc; fi
done
