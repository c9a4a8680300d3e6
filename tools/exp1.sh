#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
CDAE_CONVWIN=1 CDAE_CONVWIN_MINTILES=1 timeout 120 python tools/dbg_cw.py 8 128 128 64 2>&1 | grep "max err"
CDAE_CONVWIN=1 CDAE_CONVWIN_MINTILES=1 timeout 120 python tools/dbg_cw.py 16 256 256 8 2>&1 | grep "max err"
CDAE_CONVWIN=1 CDAE_CONVWIN_MINTILES=1 timeout 120 python tools/dbg_cw.py 3 64 96 32 2>&1 | grep "max err"
for dbg in 0 4 256 260 512; do
    CDAE_CONVWIN=1 CDAE_CONVWIN_MINTILES=256 CDAE_PS_DBG=$dbg timeout 120 python tools/ps_ablate.py 2>&1 | grep -v "^$\|amdgpu.ids"
done
} > gpurun_out/exp1.log 2>&1
tail -100 gpurun_out/exp1.log
