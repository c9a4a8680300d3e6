#!/bin/bash
timeout 200 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "window_conv or presplit or conv3x3 or upconv" 2>&1 | tail -3
timeout 120 python3 tools/prof_shapes.py --time 2>&1 | grep -v amdgpu
CDAE_PS_DBG=268 timeout 120 python3 tools/prof_shapes.py --time 2>&1 | grep -E "128->128 @64|256->256 @32"
