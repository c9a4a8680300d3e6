#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root): kernel stats + three PMC passes of eager DDIM steps.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02 v1'
# Every rocprofv3 call runs under `timeout` (a pass that aborts inside the profiler otherwise hangs until gpurun's own limit).
set -u
TAG=${1:-r05}; VER=${2:-v1}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof_$VER
mkdir -p $O
cd /tmp
CMD="python3 $R/bench.py --steps 4 --warmup 1 --regions 1 --no-graph --no-cpu-baseline --no-train --no-fp32 --no-extra"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -o w -- $CMD > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o s -- $CMD > $O/sq.log 2>&1
cd $R
# the 9-tap window conv (forward 3x3) and the 4-tap sub-pixel up-conv are different kernels with different algorithmic bytes: one summary each
python3 tools/pmc_summary.py --fetch $O/fetch --write $O/write --sq $O/sq --trace $O/trace --kernels 'convwin_kernel<false, 9' \
    --label "f16x3, eager P64 DDIM steps, batch 128: the 9-tap instantiations (256 x 128 and 256 x 96 tiles)" --out $O/${TAG}_convwin9_pmc_summary_f16x3.json > $O/summary.log 2>&1
python3 tools/pmc_summary.py --fetch $O/fetch --write $O/write --sq $O/sq --trace $O/trace --kernels 'convwin_kernel<false, 4' \
    --label "f16x3, eager P64 DDIM steps, batch 128: the 4-tap sub-pixel up-conv instantiation" --out $O/${TAG}_convwin4_pmc_summary_f16x3.json >> $O/summary.log 2>&1
python3 tools/pmc_summary.py --fetch $O/fetch --write $O/write --sq $O/sq --trace $O/trace --kernels 'ps_kernel|pswin_kernel|igemm_kernel' \
    --label "f16x3, eager P64 DDIM steps, batch 128: every contraction that is not the window conv kernel" --out $O/${TAG}_igemm_pmc_summary_f16x3.json >> $O/summary.log 2>&1
python3 tools/pmc_summary.py --fetch $O/fetch --write $O/write --sq $O/sq --trace $O/trace --kernels 'skipgn_kernel' \
    --label "f16x3, eager P64 DDIM steps, batch 128: ResBlock entry sweeps (1x1 skip conv + GroupNorm planes) and the streaming 1x1 GEMMs" --out $O/${TAG}_skipgn_pmc_summary_f16x3.json >> $O/summary.log 2>&1
python3 tools/pmc_summary.py --fetch $O/fetch --write $O/write --sq $O/sq --trace $O/trace --kernels 'gn_apply' \
    --label "eager P64 DDIM steps, batch 128: GroupNorm apply -> f16 planes" --out $O/${TAG}_gn_apply_pmc_summary.json >> $O/summary.log 2>&1
cp $O/trace/*kernel_stats.csv $O/${TAG}_bench_ddim_p64_b128_kernel_stats_${VER}_f16x3.csv 2>/dev/null
python3 tools/step_timeline.py $O/trace > $O/${TAG}_ddim_step_timeline_${VER}_f16x3.txt 2>&1
rm -rf $O/fetch $O/write $O/sq $O/trace
# ---- the same four passes in the IEEE-fp32 product mode (bench.py's `other_precision` leg): igemm_kernel on v_mfma_f32_32x32x2_f32
cd /tmp
CMD32="$CMD --precision fp32"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $CMD32 > $O/trace32.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $CMD32 > $O/fetch32.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -o w -- $CMD32 > $O/write32.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o s -- $CMD32 > $O/sq32.log 2>&1
cd $R
python3 tools/pmc_summary.py --fetch $O/fetch --write $O/write --sq $O/sq --trace $O/trace --kernels 'igemm_kernel' \
    --label "IEEE fp32 products, eager P64 DDIM steps, batch 128" --out $O/${TAG}_igemm_pmc_summary_fp32.json >> $O/summary.log 2>&1
cp $O/trace/*kernel_stats.csv $O/${TAG}_bench_ddim_p64_b128_kernel_stats_${VER}_fp32.csv 2>/dev/null
tail -80 $O/summary.log
rm -rf $O/fetch $O/write $O/sq $O/trace
du -sh $O
