#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1
tail -5 gpurun_out/t_all.log
timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d['value'], d['train'])"
