#!/bin/bash
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/exp2; mkdir -p $O; cd /tmp
for SK in 1 0; do
  export CDAE_CONVWIN_SPLITK=$SK
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$SK -o t -- python3 $R/tools/train_step.py 3 32 > $O/log$SK.txt 2>&1
  cp $O/t$SK/*kernel_stats.csv $O/stats_sk$SK.csv; rm -rf $O/t$SK
  grep -h "splitk_reduce\|convwin_kernel\|wg_reduce" $O/stats_sk$SK.csv | cut -d, -f1-4 | cut -c1-150
  tail -2 $O/log$SK.txt
done
