#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
cp causaldiffae_amd/libcdae.so /tmp/new.so
for v in new old new old; do
  cp /tmp/new.so causaldiffae_amd/libcdae.so; [ $v = old ] && cp causaldiffae_amd/libcdae_old.so causaldiffae_amd/libcdae.so
  echo "== $v"
  timeout 300 python tools/train_step.py 20 2>&1 | grep -v amdgpu | tail -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
cp /tmp/new.so causaldiffae_amd/libcdae.so
} > gpurun_out/exp1.log 2>&1
tail -10 gpurun_out/exp1.log
