#!/bin/bash
# Timing-only variants of the blocked solve (run here, no GPU): tools/probe/variants/libwm_sb_<tag>.so
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/probe/variants
for tag in base ABLATE NO_MM NO_SUBST NO_FACTOR; do
  flag=""; [ $tag != base ] && flag="-DSB_ABLATE -DSB_$tag"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iweather_model_amd/csrc $flag \
     -c weather_model_amd/csrc/solve_blocked.hip -o /tmp/sb_$tag.o
  objs=$(ls weather_model_amd/build/*.o | grep -v solve_blocked)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/variants/libwm_sb_$tag.so $objs /tmp/sb_$tag.o
done
ls -la tools/probe/variants/libwm_sb_*
