#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/m32q
mkdir -p $O
python3 tools/train_step_m32.py 20 1 2>&1 | tail -1
CDAE_WGRAD_STREAM=0 MODEL=m32 BATCH=256 FP16=1 TOP=60 timeout 300 python3 tools/train_shapes.py > $O/m32_shapes_fp16.txt 2>&1
grep -E "gn_bwd|gn_apply|fam6|total" $O/m32_shapes_fp16.txt | head -34
