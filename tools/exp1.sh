#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_model.py -q -m gpu -k "full_model or use_checkpoint or guided or p_sample_loop or mixed16_m32 or encoder_golden or trajectory" > gpurun_out/t_new.log 2>&1
tail -40 gpurun_out/t_new.log
