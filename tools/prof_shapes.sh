#!/bin/bash
# per-shape profile of the stride-1 conv3x3 launches: bash tools/prof_shapes.sh  (through gpurun, from the repo root)
set -u
TAG=${1:-r04}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/shapes
rm -rf $O; mkdir -p $O
python3 tools/prof_shapes.py --time --gm --gn > $O/timing.txt 2>&1
python3 tools/prof_shapes.py --time --gm --gn --res > $O/timing_res.txt 2>&1
cd /tmp
CMD="python3 $R/tools/prof_shapes.py --gm --gn"        # group-major planes + epilogue GroupNorm sums: how the model calls these convs
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/sq -o s -- $CMD > $O/sq.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -o w -- $CMD > $O/write.log 2>&1
cd $R
python3 tools/prof_shapes_fold.py $O > $O/${TAG}_conv_shapes.md 2> $O/fold.err
cat $O/timing.txt $O/timing_res.txt | grep -v amdgpu; cat $O/${TAG}_conv_shapes.md; cat $O/fold.err | tail -3
rm -rf $O/trace $O/sq $O/fetch $O/write
