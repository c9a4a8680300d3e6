#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tr -o t -- python3 $R/tools/train_step.py > $R/gpurun_out/prof_tr.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_tr/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
calls=sum(int(r['Calls']) for r in rows)
print("total kernel ms", tot/1e6, "calls", calls)
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f}us  {r['Name'][:120]}")
PY
tail -3 gpurun_out/prof_tr.log
rm -f gpurun_out/prof_tr/*kernel_trace.csv
