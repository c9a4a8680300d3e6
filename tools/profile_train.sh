#!/bin/bash
# Training-step profile on the GPU box: kernel stats + PMC passes (run through gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/profile_train.sh r01 v10'
set -u
TAG=${1:-r05}; VER=${2:-v1}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/proftrain_$VER
mkdir -p $O
cd /tmp
CMD="python3 $R/tools/train_step.py 3 32"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -o w -- $CMD > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o s -- $CMD > $O/sq.log 2>&1
cd $R
for K in wgwin_kernel 'convwin_kernel<true' 'convwin_kernel<false'; do
  N=$(echo $K | tr -c 'a-z_' '_' | cut -c1-24)
  python3 tools/pmc_summary.py --fetch $O/fetch --write $O/write --sq $O/sq --trace $O/trace --kernels "$K" \
      --label "f16x3 / bf16x3, C64 training step, batch 32" --out $O/${TAG}_train_pmc_${N}.json >> $O/summary.log 2>&1
done
cp $O/trace/*kernel_stats.csv $O/${TAG}_train_c64_b32_kernel_stats_${VER}.csv 2>/dev/null
python3 tools/step_timeline.py $O/trace --end adamw_ema > $O/${TAG}_train_step_timeline_${VER}.txt 2>&1
tail -60 $O/summary.log
rm -rf $O/fetch $O/write $O/sq $O/trace/*kernel_trace.csv 2>/dev/null
du -sh $O
