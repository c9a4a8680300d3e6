#!/bin/bash
# Dev tool: GPU test suite, one bench line and the kernel-trace stats of eager DDIM steps in one gpurun call.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/quick_trace.sh v2'
set -u
VER=${1:-v2}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/quick_$VER
mkdir -p $O
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.log
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 5 --warmup 1 --no-graph --no-cpu-baseline --no-train --no-fp32 > $O/trace.log 2>&1
cd $R
cp $O/trace/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/trace
head -30 $O/kernel_stats.csv | cut -c1-200
