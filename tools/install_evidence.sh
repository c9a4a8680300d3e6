#!/bin/bash
# Dev tool (runs in the build container, after `gpurun -- 'bash tools/final_profiles.sh r04 vN'`): copy what that call merged into
# gpurun_out/ to profiles/ under this round's names, renaming the versioned files of the previous evidence set.
#   bash tools/install_evidence.sh r04 v7 v6        (tag, new version, version being replaced)
set -e
TAG=${1:?tag}; V=${2:?new version}; P=${3:?previous version}
cd "$(dirname "$0")/.."
for f in ${TAG}_bench_ddim_p64_b128_kernel_stats_${P}_f16x3.csv ${TAG}_bench_ddim_p64_b128_kernel_stats_${P}_fp32.csv ${TAG}_ddim_step_timeline_${P}_f16x3.txt \
         ${TAG}_train_c64_b32_kernel_stats_${P}.csv ${TAG}_train_step_timeline_${P}.txt ${TAG}_bench_line_${P}.json; do
    [ -f profiles/$f ] && git mv profiles/$f profiles/${f/$P/$V}
done
cp gpurun_out/prof_$V/${TAG}_*.csv gpurun_out/prof_$V/${TAG}_*.json gpurun_out/prof_$V/${TAG}_*timeline*.txt profiles/
cp gpurun_out/proftrain_$V/${TAG}_*.csv gpurun_out/proftrain_$V/${TAG}_*.json gpurun_out/proftrain_$V/${TAG}_*timeline*.txt profiles/
cp gpurun_out/shapes/${TAG}_conv_shapes.md profiles/
( grep -v amdgpu gpurun_out/shapes/timing.txt; echo; echo "with a residual operand (the second conv of every ResBlock):"; grep -v amdgpu gpurun_out/shapes/timing_res.txt ) \
    > profiles/${TAG}_conv_shapes_hip_event_timing.txt
grep -v amdgpu gpurun_out/micro/skip.txt > profiles/${TAG}_skipgn_shapes_hip_event_timing.txt
grep -v amdgpu gpurun_out/micro/head.txt > profiles/${TAG}_head_micro.txt
cp gpurun_out/micro/bench_line.json profiles/${TAG}_bench_line_$V.json
python3 - "$TAG" "$V" <<'E'
import json, sys
d = json.load(open(f"profiles/{sys.argv[1]}_bench_line_{sys.argv[2]}.json"))
r, t = d["roofline"], d["train"]
print("ddim", round(d["value"], 1), round(d["ms_per_step"], 3), "convwin frac", round(r["frac"], 4), round(r["avg_launch_us"], 1), "whole-step TF", round(d["model_tflops"], 1))
print("fp32 ddim", round(d["other_precision"]["ms_per_step"], 2), round(d["other_precision"]["roofline"]["frac"], 3))
print("train", round(t["value"], 2), round(t["ms_per_step"], 3), round(t["frac_of_roof"], 4), "host ms", round(t["host_cpu_ms_per_step"], 1),
      "| world-8 policy", round(t["world8_policy"]["ms_per_step"], 2), round(t["world8_policy"]["host_cpu_over_step"], 3), "cores needed at world 8", round(t["host_cores_needed_at_world8"], 1))
print("fp32 train", round(t["fp32_mode"]["ms_per_step"], 2), round(t["fp32_mode"]["frac_of_roof"], 3))
c = t["config1_m32_b256"]
print("m32 b256", round(c["f16x3"]["ms_per_step"], 2), round(c["mixed16"]["ms_per_step"], 2), round(c["mixed16"]["frac_of_roof"], 4), "world-8 host", round(c["mixed16"]["world8_policy"]["host_cpu_over_step"], 3))
print("c64 torso", round(t["mixed16_torso"]["ms_per_step"], 2), round(t["mixed16_torso"]["frac_of_roof"], 4), "world-8 host", round(t["mixed16_torso"]["world8_policy"]["host_cpu_over_step"], 3))
print("box", d["box"], "bytes of the line", len(json.dumps(d)))
print("guided", round(d["guided_w2"]["ms_per_step"], 2), "batch16", round(d["batch16"]["ms_per_step"], 3), "public loop", round(d["public_ddim_sample_loop"]["default_ms_per_step"], 2),
      round(d["public_ddim_sample_loop"]["eager_ms_per_step"], 2))
print("cpu", round(d["cpu_baseline"]["value"], 2), round(d["cpu_baseline"]["train"]["value"], 3), "gpu/cpu", round(d["gpu_over_cpu"], 1))
E
