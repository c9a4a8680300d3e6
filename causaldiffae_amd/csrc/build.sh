#!/bin/bash
# Build libcdae.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libcdae.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="${EXTRA_HIPCC_FLAGS:-} --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -I../../include"
mkdir -p build
pids=()
$HIPCC $FLAGS -c igemm.hip -o build/igemm.o & pids+=($!)
$HIPCC $FLAGS -c api.hip -o build/api.o & pids+=($!)
$HIPCC $FLAGS -c norm.hip -o build/norm.o & pids+=($!)
$HIPCC $FLAGS -ffp-contract=off -c elementwise.hip -o build/elementwise.o & pids+=($!)
$HIPCC $FLAGS -c prof.hip -o build/prof.o & pids+=($!)
$HIPCC $FLAGS -c attention.hip -o build/attention.o & pids+=($!)
$HIPCC $FLAGS -c wgrad.hip -o build/wgrad.o & pids+=($!)
$HIPCC $FLAGS -c stem.hip -o build/stem.o & pids+=($!)
$HIPCC $FLAGS -c convwin.hip -o build/convwin.o & pids+=($!)
$HIPCC $FLAGS -c skipgn.hip -o build/skipgn.o & pids+=($!)
$HIPCC $FLAGS -c head.hip -o build/head.o & pids+=($!)
for pid in "${pids[@]}"; do wait "$pid" || { echo "build.sh: a compile step failed" >&2; exit 1; }; done     # a bare `wait` would hide failures
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT build/igemm.o build/api.o build/norm.o build/elementwise.o build/prof.o build/attention.o build/wgrad.o build/stem.o build/convwin.o build/skipgn.o build/head.o
echo "built $(realpath $OUT)"
