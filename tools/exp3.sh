#!/bin/bash
timeout 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "group_major or group_norm" 2>&1 | tail -1
for R in 1 2; do for G in 0 1; do
  echo "== CDAE_PLANES_GM=$G"; CDAE_PLANES_GM=$G timeout 300 python3 bench.py --no-cpu-baseline --no-train --no-fp32 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['family_ms_per_step']['groupnorm'])"
done; done
