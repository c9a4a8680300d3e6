#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for dp in 1 0 1 0; do
echo "dephase $dp"
CDAE_CONVWIN_DEPHASE=$dp timeout 120 python tools/ps_ablate.py 2>&1 | grep -v "^$\|amdgpu.ids" | head -3
done
CDAE_CONVWIN_DEPHASE=1 timeout 600 python bench.py --no-train --no-cpu-baseline --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dephase 1', d['ms_per_step'], d['value'])"
CDAE_CONVWIN_DEPHASE=0 timeout 600 python bench.py --no-train --no-cpu-baseline --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dephase 0', d['ms_per_step'], d['value'])"
CDAE_CONVWIN_DEPHASE=1 timeout 600 python bench.py --no-train --no-cpu-baseline --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dephase 1', d['ms_per_step'], d['value'])"
} > gpurun_out/exp1.log 2>&1
tail -30 gpurun_out/exp1.log
