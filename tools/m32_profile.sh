#!/bin/bash
# BASELINE config [1] (M32, batch 256) training step, mixed16 and f16x3: serial kernel trace + per-shape label table.
#   gpurun --timeout 900 -- 'bash tools/m32_profile.sh v0'
set -u
VER=${1:-v0}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/m32_$VER
mkdir -p $O
cd /tmp
export CDAE_WGRAD_STREAM=0
for FP in 1 0; do
  export FP16=$FP
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$FP -o t -- python3 $R/tools/train_step_m32.py 3 $FP > $O/trace$FP.log 2>&1
  python3 $R/tools/step_timeline.py $O/trace$FP --end adamw_ema > $O/m32_b256_timeline_fp16_${FP}_serial.txt 2>&1
  rm -rf $O/trace$FP
done
cd $R
MODEL=m32 BATCH=256 FP16=1 TOP=60 timeout 300 python3 tools/train_shapes.py > $O/m32_shapes_fp16.txt 2>&1
head -60 $O/m32_b256_timeline_fp16_1_serial.txt
