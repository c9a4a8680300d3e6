#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
CDAE_CONVWIN=1 CDAE_CONVWIN_MINTILES=1 timeout 120 python tools/dbg_cw.py 8 128 128 64 2>&1 | grep "max err"
CDAE_CONVWIN=1 CDAE_CONVWIN_MINTILES=1 timeout 120 python tools/dbg_cw.py 3 64 96 32 2>&1 | grep "max err"
for dbg in 0 4 256; do
    CDAE_CONVWIN=1 CDAE_CONVWIN_MINTILES=256 CDAE_PS_DBG=$dbg timeout 120 python tools/ps_ablate.py 2>&1 | grep -v "^$\|amdgpu.ids"
done
CDAE_CONVWIN=1 CDAE_KPACK=0 CDAE_CONVWIN_MINTILES=256 timeout 120 python tools/ps_ablate.py 2>&1 | grep -v "^$\|amdgpu.ids"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3 or upconv or presplit or fused_groupnorm" 2>&1 | tail -5
timeout 600 python bench.py --no-train --no-cpu-baseline 2>&1 | tail -2
CDAE_CONVWIN=0 timeout 600 python bench.py --no-train --no-cpu-baseline 2>&1 | tail -2
} > gpurun_out/exp1.log 2>&1
tail -60 gpurun_out/exp1.log
