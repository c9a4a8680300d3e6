#!/bin/bash
# Dev tool: kernel-trace stats of a few C64 training steps (batch 32) -> gpurun_out/train_trace/kernel_stats.csv
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/train_trace; mkdir -p $O; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 $R/tools/train_step.py 5 32 > $O/log.txt 2>&1
cp $O/t/*kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/t
tail -1 $O/log.txt | cut -c1-200
