#!/bin/bash
# HBM-traffic PMC passes over ONE bench workload: tools/pmc_one.sh <workload> <tag>
set -u
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${2:-pmc_one}
mkdir -p $OUT
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -s KILL 170 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$grp -- \
      python3 $R/bench.py --workload $1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$grp.log 2>&1
  rc=$?
  echo "pass ($grp) exit $rc"
  if [ $rc -ne 0 ]; then exit $rc; fi
done
