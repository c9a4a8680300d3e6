#!/bin/bash
# Dev tool: same-box A/B of two builds of libcdae.so (gpurun_ab_libA.so / gpurun_ab_libB.so at the repo root) on the conv shape timings
for V in A B A B; do
  cp gpurun_ab_lib$V.so causaldiffae_amd/libcdae.so
  echo "== lib $V"; timeout 120 python3 tools/prof_shapes.py --time ${AB_ARGS:-} 2>&1 | grep -E "128->128 @64|256->128 @64|256->256 @32|384->384 @16|512->512 @ 8"
done
