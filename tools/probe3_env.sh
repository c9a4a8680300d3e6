#!/bin/bash
# Dev: tools/two_stream_probe3.py (modes one / dep1 / val1 only) under ROCm runtime settings: what stops rocr's AsyncEventsLoop thread from spinning?
for e in "" "ROC_SIGNAL_POOL_SIZE=4096" "ROC_SIGNAL_POOL_SIZE=32" "ROC_AQL_QUEUE_SIZE=65536" "ROC_AQL_QUEUE_SIZE=1024" "DEBUG_CLR_MAX_BATCH_SIZE=1" "DEBUG_CLR_MAX_BATCH_SIZE=10000" \
         "GPU_MAX_COMMAND_BUFFERS=64" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "AMD_SERIALIZE_KERNEL=0" "GPU_FORCE_QUEUE_PROFILING=0" "HIP_LAUNCH_BLOCKING=0" "DEBUG_HIP_KERNARG_COPY_OPT=0" "HSA_KERNARG_POOL_SIZE=8388608"; do
  echo "== ${e:-default}"
  env $e MODES=one,dep1,val1 timeout 100 python3 tools/two_stream_probe3.py 2>&1 | grep wall
done
