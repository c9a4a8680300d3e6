#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "race_free or lds_out_of_range" > gpurun_out/t_race.log 2>&1
tail -8 gpurun_out/t_race.log
