#!/bin/bash
# Dev tool: the model-level GPU tests with each fused path of the Python layer switched off in turn (the predecessor paths must stay
# green; the tests that assert WHICH kernel ran — packed weight planes, benchmark dispatch — are left out: they fail by design here).
# The toggles are constants of causaldiffae_amd/ops.py (PATH_TOGGLES), flipped by the test session (`--paths-off`), not by the environment.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/switch_matrix.sh'
for SW in skipgn_v2 skip_gn stream_gemm head_conv planes_gm linear_gn fused_attn fused_attn_train kpack presplit train_presplit train_rbnode \
          train_gnparts train_emball train_cat weight_bank wscale wgrad_stream s2_dgrad_ps dgrad_stream wgrad_group lwgrad_group; do
  echo "== $SW off"
  timeout 900 python3 -m pytest tests/test_gpu_model.py -x -q --paths-off $SW -k "(unet_forward or ddim_p64 or p_sample_loop or guided or full_model or trainloop) and not packed_weight and not benchmark_dispatch" 2>&1 | tail -2
done
# the 16-bit torso with its own toggles and with the packed planes off (round-5 advisor finding: pointers16 needs kpack)
for SW in torso16 down16 im2col16 kpack lwgrad_group; do
  echo "== torso tests, $SW off"
  timeout 600 python3 -m pytest tests/test_gpu_torso16.py tests/test_gpu_model.py -x -q --paths-off $SW -k "(full_model_step_on_the_16bit or curve_reference or m32_batch256 or odd_sized) and not packed_weight" 2>&1 | tail -2
done
echo "== smoke"; timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench 2 ranks (gloo on one GPU)"; timeout 600 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-extra 2>&1 | tail -1 | cut -c1-400
