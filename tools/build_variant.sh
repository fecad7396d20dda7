#!/usr/bin/env bash
# Developer A/B helper: build libesctp1rt.so from a source tree (default: this one) with extra
# compiler flags into build/variants/<name>.so; time it with ESC_LIB_PATH=build/variants/<name>.so.
#   tools/build_variant.sh <name> [src-root] [extra hipcc flags...]
set -e
NAME=$1; SRC=${2:-.}; shift; shift || true
HERE=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $HERE/build/variants
cd $SRC
PKG=esctp1raytracer_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize \
  -Iinclude -I$PKG/host -I$PKG/csrc -Wall -Wno-unused-function "$@" -shared \
  -o $HERE/build/variants/$NAME.so $PKG/csrc/rt_kernels.hip $PKG/csrc/rt_capi.cpp $PKG/csrc/rt_multi.cpp \
  $PKG/host/host_core.cpp $PKG/host/obj_loader.cpp $PKG/host/synth.cpp $PKG/host/accel_build.cpp -ldl
echo built build/variants/$NAME.so
