#!/bin/bash
# rocprofv3 kernel stats of ONE bench workload (run on the GPU box from the repo root):
#   tools/profile_one.sh <workload> <tag> [steps]  ->  gpurun_out/prof_<tag>/
set -u
wl=$1; tag=$2; steps=${3:-50}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$tag -- \
    python3 $ROOT/bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline > $ROOT/gpurun_out/prof_$tag.log 2>&1
echo "rocprof exit $?"
tail -1 $ROOT/gpurun_out/prof_$tag.log | cut -c1-600
