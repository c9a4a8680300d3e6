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
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq -o s -- $CMD > $O/sq.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d $O/tcp -o s -- $CMD > $O/tcp.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/sq2 -o s -- $CMD > $O/sq2.log 2>&1
cd $R
python3 tools/pmc_fold.py 'pswin|convwin' $O/trace $O/sq $O/tcp $O/sq2 > $O/summary.txt 2>&1
cat $O/summary.txt
rm -rf $O/trace $O/sq $O/tcp $O/sq2
