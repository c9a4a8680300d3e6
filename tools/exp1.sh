#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for dbg in 0 2048 0 2048; do
CDAE_PS_DBG=$dbg timeout 120 python tools/ps_ablate.py 2>&1 | grep -v "^$\|amdgpu.ids" | head -3
done
} > gpurun_out/exp1.log 2>&1
tail -14 gpurun_out/exp1.log
