#!/bin/bash
# Dev tool: the model-level GPU tests with each fast path of the Python layer switched off in turn (the fallbacks must stay green;
# the tests that assert WHICH kernel ran — packed weight planes, benchmark dispatch — are left out: they fail by design here).
# These are the module-level `CDAE_*` dev switches of causaldiffae_amd/ops.py, read once at import; the LIBRARY reads no dispatch
# switch from the environment (thresholds: cdae_tune_set; compile-time experiments: a -DCW_DEV=1 build).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/switch_matrix.sh'
for SW in CDAE_SKIPGN_V2 CDAE_SKIP_GN CDAE_STREAM_GEMM CDAE_HEAD_CONV CDAE_PLANES_GM CDAE_LINEAR_GN CDAE_FUSED_ATTN CDAE_FUSED_ATTN_TRAIN CDAE_KPACK \
          CDAE_PRESPLIT CDAE_TRAIN_PRESPLIT CDAE_TRAIN_RBNODE CDAE_TRAIN_GNPARTS CDAE_TRAIN_EMBALL CDAE_TRAIN_CAT CDAE_WEIGHT_BANK \
          CDAE_WSCALE CDAE_WGRAD_STREAM CDAE_S2_DGRAD_PS CDAE_DGRAD_STREAM; do
  echo "== $SW=0"
  env $SW=0 timeout 900 python3 -m pytest tests/test_gpu_model.py -x -q -k "(unet_forward or ddim_p64 or p_sample_loop or guided or full_model or trainloop) and not packed_weight and not benchmark_dispatch" 2>&1 | tail -2
done
echo "== smoke"; timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench 2 ranks (gloo on one GPU)"; CDAE_DIST_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-extra 2>&1 | tail -1 | cut -c1-400
