#!/bin/bash
# Dev tool: the model-level GPU tests with each new kernel switched off in turn (the fallbacks must stay green).
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/switch_matrix.sh'
for SW in CDAE_SKIPGN_V2 CDAE_STREAM_GEMM CDAE_HEAD_CONV CDAE_SPLITK_REDUCE4 CDAE_GN_PARTS_FUSED CDAE_UPCONV_FUSED CDAE_PLANES_GM CDAE_GN_APPLY_GM CDAE_GN_BWD_STREAM CDAE_CONVWIN CDAE_LINEAR_GN CDAE_FUSED_ATTN_TRAIN CDAE_GN_BWD_PARAM_ROW CDAE_KS_ROUNDS CDAE_ATTN_BWD_FUSED; do
  echo "== $SW=0"
  env $SW=0 timeout 600 python3 -m pytest tests/test_gpu_model.py -x -q -k "unet_forward or ddim_p64 or p_sample_loop or guided or full_model" 2>&1 | tail -2
done
for KV in CDAE_WG_KS_CEIL=1 CDAE_GN_CHUNK_CAP=2048 CDAE_CONVWIN_MINCHUNK=2; do
  echo "== $KV"
  env $KV timeout 600 python3 -m pytest tests/test_gpu_model.py -x -q -k "unet_forward or ddim_p64 or full_model" 2>&1 | tail -2
done
echo "== smoke"; timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench 2 ranks (gloo on one GPU)"; CDAE_DIST_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
