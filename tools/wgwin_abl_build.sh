#!/bin/bash
# Dev tool: libcdae.so variants with wgwin_kernel ablations (-DWG_ABL=<bits>, wgrad.hip) as gpurun_ab_lib<k>.so for tools/ab_lib.sh
# (timing only: ablated kernels compute wrong results).   usage: bash tools/wgwin_abl_build.sh 0 1 2 4 8
set -euo pipefail
cd "$(dirname "$0")/../causaldiffae_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-inline-asm -I../../include"
objs=""; for u in igemm planes api norm elementwise prof attention attn16 rows16 wg16 stem convwin skipgn head; do objs="$objs build/$u.o"; done
k=0
for a in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DWG_ABL=$a -c wgrad.hip -o /tmp/wgrad_abl_$a.o &
done
wait
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_ab_lib$k.so $objs /tmp/wgrad_abl_$a.o
  echo "gpurun_ab_lib$k.so = WG_ABL=$a"; k=$((k+1))
done
