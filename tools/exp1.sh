#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1200 python bench.py ) > gpurun_out/bench_full.log 2>&1
tail -4 gpurun_out/bench_full.log
