#!/bin/bash
# torso16 tests + per-shape label table of the M32 b256 mixed16 step
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/m32q
mkdir -p $O
python3 -m pytest tests/test_gpu_torso16.py -q 2>&1 | tail -40
CDAE_WGRAD_STREAM=0 MODEL=m32 BATCH=256 FP16=1 TOP=70 timeout 300 python3 tools/train_shapes.py > $O/m32_shapes_fp16.txt 2>&1
head -75 $O/m32_shapes_fp16.txt
