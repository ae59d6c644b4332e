#!/bin/bash
# build_variant.sh <name> <trace_kernels source> [extra hipcc flags]: a second libptamd build under ab/ for
# A/B timing on the GPU box (select it with PTAMD_LIB=ab/<name>.so); API_SRC=<file> replaces api.hip as well
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
mkdir -p "$ROOT/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -I"$ROOT/path-tracer_amd/csrc" -I"$ROOT/include" "$@" \
    -shared -o "$ROOT/ab/$name.so" "${API_SRC:-$ROOT/path-tracer_amd/csrc/api.hip}" "$ROOT/path-tracer_amd/csrc/build_kernels.hip" "$src"
echo "built ab/$name.so"
