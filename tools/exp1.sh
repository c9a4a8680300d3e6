#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
{
timeout 600 python bench.py --no-train --no-cpu-baseline --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio on ', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
CDAE_PS_DBG=64 timeout 600 python bench.py --no-train --no-cpu-baseline --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio off', d['ms_per_step'], d['value'])"
timeout 600 python bench.py --no-train --no-cpu-baseline --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio on ', d['ms_per_step'], d['value'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2b -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-train --no-fp32 > $R/gpurun_out/prof_r2b.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_r2b/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms (6 steps incl. warm-up and probe)", tot/1e6)
for r in rows[:24]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f}us  {r['Name'][:110]}")
PY
rm -f gpurun_out/prof_r2b/*kernel_trace.csv
} > gpurun_out/exp1.log 2>&1
tail -40 gpurun_out/exp1.log
