#!/bin/bash
# usage: tools/build_variant.sh NAME [-DFLAG ...]   -> build/variants/librrt_NAME.so (experiment builds; RRT_LIB selects one)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -fPIC -shared \
  -I $R/include -I $R/reversible-raytracer_amd/csrc "$@" \
  $R/reversible-raytracer_amd/csrc/rrt_kernels.hip -o $R/build/variants/librrt_$name.so
echo build/variants/librrt_$name.so
