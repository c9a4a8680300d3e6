#!/bin/bash
cp causaldiffae_amd/libcdae.so /tmp/keep.so; cp gpurun_ab_libD.so causaldiffae_amd/libcdae.so
for A in "" "--gm"; do for D in 0 16 4 0; do
  echo "== planes '$A' CDAE_PS_DBG=$D"
  CDAE_PS_DBG=$D timeout 120 python3 tools/prof_shapes.py --time $A 2>&1 | grep -E "128->128 @64|256->256 @32|384->384 @16"
done; done
cp /tmp/keep.so causaldiffae_amd/libcdae.so
