#!/bin/bash
# Dev tool: same-box A/B of two builds of libcdae.so (gpurun_ab_libA.so / gpurun_ab_libB.so at the repo root).
#   AB_CMD='python3 tools/train_step.py 20 32' bash tools/ab_lib.sh        (default: the conv shape timings, AB_ARGS=--res for residual convs)
CMD=${AB_CMD:-"python3 tools/prof_shapes.py --time ${AB_ARGS:-}"}
for V in A B A B; do
  cp gpurun_ab_lib$V.so causaldiffae_amd/libcdae.so
  echo "== lib $V"; timeout 300 $CMD 2>&1 | grep -v amdgpu | grep -E "${AB_GREP:-128->128 @64|256->128 @64|256->256 @32|384->384 @16|512->512 @ 8|value}" | cut -c1-200
done
