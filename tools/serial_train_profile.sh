#!/bin/bash
# Serial (side stream OFF) kernel trace of the C64 batch-32 training step: per-kernel durations that are not inflated by
# concurrent weight-gradient launches, + the per-shape label table and the wall time of both policies.
#   gpurun --timeout 900 -- 'bash tools/serial_train_profile.sh r05 v1'
set -u
TAG=${1:-r05}; VER=${2:-v1}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/serial_$VER
mkdir -p $O
cd /tmp
export CDAE_WGRAD_STREAM=0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/tools/train_step.py 3 32 > $O/trace.log 2>&1
cd $R
cp $O/trace/*kernel_stats.csv $O/${TAG}_train_c64_b32_kernel_stats_${VER}_serial.csv 2>/dev/null
python3 tools/step_timeline.py $O/trace --end adamw_ema > $O/${TAG}_train_step_timeline_${VER}_serial.txt 2>&1
python3 tools/step_timeline.py $O/trace --end adamw_ema --list > $O/${TAG}_train_step_list_${VER}_serial.txt 2>&1
rm -rf $O/trace
timeout 300 python3 tools/train_shapes.py > $O/train_shapes_serial.txt 2>&1
timeout 300 python3 tools/exp_train.py > $O/exp_serial.txt 2>&1
unset CDAE_WGRAD_STREAM
timeout 300 python3 tools/exp_train.py > $O/exp_side.txt 2>&1
cat $O/exp_serial.txt $O/exp_side.txt
head -40 $O/${TAG}_train_step_timeline_${VER}_serial.txt
