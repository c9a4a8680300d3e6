#!/bin/bash
# PMC passes of the BASELINE config [1] training step on the 16-bit torso (side stream off): HBM traffic / MFMA busy / LDS conflicts of the
# streaming kernels (rows16, wg16) and of the window conv's bf16-row form.    gpurun --timeout 1500 -- 'bash tools/profile_m32.sh r05'
set -u
TAG=${1:-r05}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/profm32
mkdir -p $O
cd /tmp
export CDAE_WGRAD_STREAM=0
CMD="python3 $R/tools/train_step_m32.py 3 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -o w -- $CMD > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o s -- $CMD > $O/sq.log 2>&1
cd $R
for K in rows16_ring_kernel rows16_reg_kernel wg16_kernel 'convwin_kernel<true, 9, 2, 4, true, true>' 'wgwin_kernel<1' attn16_fwd_kernel attn16_bwd_kernel gn_bwd_partial_kernel gn_bwd_dx_stream_kernel; do
  N=$(echo $K | tr -c 'a-z0-9_' '_' | cut -c1-28)
  python3 tools/pmc_summary.py --fetch $O/fetch --write $O/write --sq $O/sq --trace $O/trace --kernels "$K" \
      --label "BASELINE config [1] training step on the 16-bit torso, batch 256" --out $O/${TAG}_m32_pmc_${N}.json >> $O/summary.log 2>&1
done
tail -150 $O/summary.log
rm -rf $O/fetch $O/write $O/sq $O/trace
