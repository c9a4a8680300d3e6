#!/bin/bash
# Round-end evidence in one gpurun call: PMC + kernel stats of the DDIM step and of the training step, per-shape tables.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/final_profiles.sh r02 v2'
TAG=${1:-r06}; VER=${2:-v1}
bash tools/profile_round.sh $TAG $VER 2>&1 | tail -40
bash tools/profile_train.sh $TAG $VER 2>&1 | tail -25
bash tools/prof_shapes.sh $TAG 2>&1 | tail -45
mkdir -p gpurun_out/micro
timeout 200 python3 tools/prof_skip.py > gpurun_out/micro/skip.txt 2>&1; grep -v amdgpu gpurun_out/micro/skip.txt
timeout 200 python3 tools/prof_head.py > gpurun_out/micro/head.txt 2>&1; grep -v amdgpu gpurun_out/micro/head.txt
bash tools/serial_train_profile.sh $TAG $VER 2>&1 | tail -6          # side stream OFF: per-kernel durations that concurrency does not inflate
bash tools/m32_quick.sh $TAG 2>&1 | tail -12                                # BASELINE config [1] on the 16-bit torso: serial kernel stats + timeline
bash tools/b16_profile.sh $TAG 2>&1 | tail -12                              # batch-16 DDIM step: per-shape table, kernel stats, timeline
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/micro/bench_line.json 2> gpurun_out/micro/bench.err; tail -c 600 gpurun_out/micro/bench_line.json
