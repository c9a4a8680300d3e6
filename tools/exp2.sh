#!/bin/bash
cd "$(dirname "$0")/.."
cp causaldiffae_amd/libcdae.so /tmp/libcdae_orig.so
for v in 1 2 3; do
  cp causaldiffae_amd/libcdae_v$v.so causaldiffae_amd/libcdae.so
  echo "== variant $v"
  CDAE_CONVWIN=1 CDAE_CONVWIN_MINTILES=1 timeout 120 python tools/dbg_cw.py 8 128 128 64 2>&1 | grep "max err"
done
cp /tmp/libcdae_orig.so causaldiffae_amd/libcdae.so
