#!/bin/bash
# build_native.sh <out-name> [extra hipcc flags...] -- builds tools/probe/variants/<out-name>
set -e
cd "$(dirname "$0")/../.."
out=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iweather_model_amd/csrc "$@" \
    -x hip tools/probe/conv32_native.cpp weather_model_amd/csrc/conv3d.hip weather_model_amd/csrc/common.hip \
    -o tools/probe/variants/$out 2>&1 | grep -v "warning\|loop not unrolled\|^ *[0-9]* |\|^ *|\|generated" || true
ls -la tools/probe/variants/$out
