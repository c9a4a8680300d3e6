#!/bin/bash
# batch-16 P64 DDIM step (the reference evaluation script's default batch): per-shape label table + kernel-trace timeline and stats
TAG=${1:-r06}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/b16
mkdir -p $O
BATCH=16 TOP=70 python3 tools/ddim_shapes.py > $O/ddim_shapes_b16.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 4 --warmup 1 --regions 1 --no-graph --no-cpu-baseline --no-train --no-fp32 --no-extra --batch 16 > $O/trace.log 2>&1
cd $R
python3 tools/step_timeline.py $O/trace > $O/${TAG}_ddim_step_timeline_b16.txt 2>&1
cp $O/trace/*kernel_stats.csv $O/${TAG}_bench_ddim_p64_b16_kernel_stats.csv
rm -rf $O/trace
head -40 $O/${TAG}_ddim_step_timeline_b16.txt
