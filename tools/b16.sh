#!/bin/bash
# Dev: the batch-16 DDIM step (SURVEY 8d config 3, N = 16) under an environment setting: graph-replay ms per step from bench.py
for v in "$@"; do
  echo "== $v"; env $v python3 bench.py --steps 20 --warmup 5 --regions 3 --no-train --no-fp32 --no-extra --no-cpu-baseline --batch 16 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('batch 16: %.3f ms per step, %.1f image-steps/s' % (d['ms_per_step'], d['value']))"
done
