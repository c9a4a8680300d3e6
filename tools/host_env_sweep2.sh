#!/bin/bash
# Dev: the C64 batch-32 training step (tools/exp_train.py: ms per step, process CPU per step, CPU per thread) under runtime settings
for e in "" "DEBUG_CLR_MAX_BATCH_SIZE=10000" "DEBUG_CLR_MAX_BATCH_SIZE=10000 OFF=stream_links" "OFF=stream_links" "DEBUG_CLR_MAX_BATCH_SIZE=10000 CDAE_WGRAD_STREAM=0" "CDAE_WGRAD_STREAM=0" \
         "DEBUG_CLR_MAX_BATCH_SIZE=100000 ROC_AQL_QUEUE_SIZE=1024"; do
  echo "== ${e:-default}"
  env $e STEPS=30 REGIONS=2 timeout 200 python3 tools/exp_train.py 2>&1 | tail -2
done
