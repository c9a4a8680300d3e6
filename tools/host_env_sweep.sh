#!/bin/bash
# Dev: CPU time per thread of the C64 batch-32 training step under ROCm runtime environment settings (one process each).
#   gpurun --timeout 900 -- 'bash tools/host_env_sweep.sh'
for e in "" "ROC_CPU_WAIT_FOR_SIGNAL=0" "ROC_ACTIVE_WAIT_TIMEOUT=0" "ROC_ACTIVE_WAIT_TIMEOUT=50" "AMD_DIRECT_DISPATCH=0" "GPU_MAX_HW_QUEUES=2" "DEBUG_HIP_DYNAMIC_QUEUES=0" \
         "HSA_ENABLE_INTERRUPT=0" "ROC_SYSTEM_SCOPE_SIGNAL=0" "DEBUG_HIP_BLOCK_SYNC=1" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_CLR_BATCH_CPU_SYNC_SIZE=64"; do
  echo "== ${e:-default}"
  env $e timeout 200 python3 tools/host_threads.py 2>&1 | tail -1
done
