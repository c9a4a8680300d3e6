#!/bin/bash
# PMC passes of one conv shape: bash tools/prof_conv.sh TAG Cin Cout res   (env selects the kernel variant)
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pc_$TAG
mkdir -p $O
cd /tmp
CMD="python3 $R/tools/prof_conv.py $*"
timeout 150 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq -o s -- $CMD > $O/sq.log 2>&1
if [ "${PROF_MORE:-0}" = "1" ]; then
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq2 -o s -- $CMD > $O/sq2.log 2>&1
timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/f -o s -- $CMD > $O/f.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/w -o s -- $CMD > $O/w.log 2>&1
fi
cd $R
python3 tools/pmc_fold.py 'pswin|convwin' $O/trace $O/sq $O/sq2 $O/f $O/w > $O/summary.txt 2>&1
cat $O/summary.txt
rm -rf $O/trace $O/sq $O/sq2 $O/f $O/w
