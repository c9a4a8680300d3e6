#!/bin/bash
# Build libcdae.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# A translation unit is recompiled when it, a shared header or the flag set is newer than its object (FORCE=1: everything).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libcdae.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="${EXTRA_HIPCC_FLAGS:-} --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-inline-asm -I../../include"
UNITS="igemm planes api norm elementwise prof attention attn16 rows16 wg16 wgrad stem convwin skipgn head"
mkdir -p build
# (the flag record is written only after every compile step has succeeded: a failed build after a flag change must not leave
#  objects of the old flag set looking current)
if [ "$(cat build/.flags 2>/dev/null || true)" != "$FLAGS" ]; then FORCE=1; rm -f build/.flags; fi
pids=()
for u in $UNITS; do
    obj=build/$u.o
    if [ "${FORCE:-0}" = 1 ] || [ ! -f $obj ] || [ $u.hip -nt $obj ] || [ cdae_internal.h -nt $obj ] || [ gemm_common.h -nt $obj ] || [ ../../include/cdae.h -nt $obj ]; then
        extra=""
        [ $u = elementwise ] && extra="-ffp-contract=off"       # sampler updates round like the reference's separate ATen ops
        $HIPCC $FLAGS $extra -c $u.hip -o $obj & pids+=($!)
    fi
done
for pid in "${pids[@]:-}"; do [ -z "$pid" ] || wait "$pid" || { echo "build.sh: a compile step failed" >&2; exit 1; }; done     # a bare `wait` would hide failures
echo "$FLAGS" > build/.flags
objs=""
for u in $UNITS; do objs="$objs build/$u.o"; done
# build/ holds exactly the current units' objects: anything else (retired objects, -save-temps dumps) would ride to every GPU lease
for f in build/*; do
    keep=0; for u in $UNITS; do [ "$f" = "build/$u.o" ] && keep=1; done
    [ $keep = 1 ] || rm -rf "$f"
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $objs
echo "built $(realpath $OUT)"
