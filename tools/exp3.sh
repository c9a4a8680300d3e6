#!/bin/bash
for R in 1 2 3; do for G in 0 1; do
  echo "== CDAE_PLANES_GM=$G"; CDAE_PLANES_GM=$G timeout 300 python3 bench.py --no-cpu-baseline --no-train --no-fp32 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
done; done
