#!/bin/bash
# diagnostic: the whole library rebuilt with extra compiler flags into tools/ubench/librls_<name>.so
# usage: tools/build_variant.sh <name> [flags...]
set -e
name=$1; shift
cd "$(dirname "$0")/../regularizedleastsquares.jl_amd/csrc"
B=$(mktemp -d)
for f in api comm f64 gemv normal gramk small skinny setup kaczmarz blas1 prox pgm svt tv nested solvers; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -Wno-unused-value -ffp-contract=fast -fno-slp-vectorize "$@" -c $f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/librls_$name.so $B/*.o -ldl
