#!/bin/bash
# mobileposer_amd/libmp_<NAME>.so: the whole library compiled with extra flags (A/B builds, used through MP_LIB_PATH; the
# binding skips the build-id check for an overridden path).   bash tools/debug/build_variant.sh tanhpoly "-DMP_TANH_POLY=1"
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; FLAGS=$2
mkdir -p /tmp/mpv_$NAME
for src in $ROOT/mobileposer_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $src -o /tmp/mpv_$NAME/$(basename ${src%.hip}).o \
     -I $ROOT/include -Xclang -target-feature -Xclang -packed-fp32-ops $FLAGS 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/mobileposer_amd/libmp_$NAME.so /tmp/mpv_$NAME/*.o
echo built $ROOT/mobileposer_amd/libmp_$NAME.so
