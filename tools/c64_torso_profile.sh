#!/bin/bash
# Serial (side stream OFF) kernel trace of the C64 batch-32 training step on the 16-bit torso: is that leg bound by the GPU's kernel time or by the host's launch path?
#   gpurun --timeout 900 -- 'bash tools/c64_torso_profile.sh r06'
TAG=${1:-r06}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/c64torso
mkdir -p $O
cd /tmp
export CDAE_WGRAD_STREAM=0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/tools/exp_train.py > $O/trace.log 2>&1
cd $R
python3 tools/step_timeline.py $O/trace --end adamw_ema > $O/${TAG}_c64_b32_mixed16_timeline_serial.txt 2>&1
rm -rf $O/trace
head -30 $O/${TAG}_c64_b32_mixed16_timeline_serial.txt
