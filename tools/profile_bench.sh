#!/bin/bash
# Bench line + rocprofv3 kernel stats of the same command (run on the GPU box from the repo root):
#   tools/profile_bench.sh <tag>   ->  gpurun_out/bench_<tag>.json, gpurun_out/prof_<tag>/
set -u
tag=${1:-run}
ROOT=$(pwd)
timeout -k 10 700 python3 -u bench.py --workload all > gpurun_out/bench_$tag.log 2>&1 || { echo "bench failed"; tail -5 gpurun_out/bench_$tag.log; exit 1; }
tail -1 gpurun_out/bench_$tag.log > gpurun_out/bench_$tag.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$tag -- \
    python3 $ROOT/bench.py --workload all --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/gpurun_out/prof_$tag.log 2>&1
echo "rocprof exit $?"
