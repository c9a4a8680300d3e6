#!/bin/bash
TAG=${1:-r06}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/m32q
mkdir -p $O
cd /tmp
export CDAE_WGRAD_STREAM=0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/tools/train_step_m32.py 3 1 > $O/trace.log 2>&1
python3 $R/tools/step_timeline.py $O/trace --end adamw_ema > $O/${TAG}_m32_b256_mixed16_timeline_serial.txt 2>&1
cp $O/trace/*kernel_stats.csv $O/${TAG}_train_m32_b256_mixed16_kernel_stats_serial.csv
rm -rf $O/trace
head -70 $O/${TAG}_m32_b256_mixed16_timeline_serial.txt
