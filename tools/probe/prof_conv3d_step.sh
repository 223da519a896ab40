#!/bin/bash
# rocprofv3 kernel stats of the Conv3D model training step (tools/probe/conv3d_step.py)
set -u
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_c3d -- python3 $ROOT/tools/probe/conv3d_step.py > $ROOT/gpurun_out/prof_c3d.log 2>&1
echo "exit $?"
