#!/bin/bash
# Dev tool: same-box comparison of several builds of libcdae.so (gpurun_ab_lib?.so at the repo root), two rounds each.
#   AB_CMD='python3 tools/train_step.py 20 32' bash tools/ab_lib.sh        (default: the conv shape timings, AB_ARGS=--res for residual convs)
CMD=${AB_CMD:-"python3 tools/prof_shapes.py --time ${AB_ARGS:-}"}
cp causaldiffae_amd/libcdae.so /tmp/libcdae_keep.so
for R in 1 2; do for L in gpurun_ab_lib?.so; do
  cp $L causaldiffae_amd/libcdae.so
  echo "== $L"; timeout 300 $CMD 2>&1 | grep -v amdgpu | grep -E "${AB_GREP:-128->128 @64|256->128 @64|256->256 @32|384->384 @16|512->512 @ 8|value}" | cut -c1-6000
done; done
cp /tmp/libcdae_keep.so causaldiffae_amd/libcdae.so
