#!/bin/bash
# Dev tool: which thread of a training process burns a core?  Starts the C64 batch-32 training loop (tools/exp_train.py), waits until it is
# in its timed regions, lists CPU ticks per thread over 5 s and dumps the native stacks of all threads (HIP / HSA / libc symbols only).
#   gpurun --timeout 600 -- 'bash tools/spin_bt.sh [ENV=VALUE ...]'
O=gpurun_out/spin_bt; mkdir -p $O; rm -f $O/*
env "$@" PTRACE_ANY=1 STEPS=1500 REGIONS=1 python3 tools/exp_train.py > $O/out.txt 2> $O/err.txt &
JOB=$!
sleep 45
pid=$JOB
for t in /proc/$pid/task/*; do echo "$(basename $t) $(cat $t/comm) $(cut -d' ' -f14,15 $t/stat)"; done > $O/t0.txt
sleep 5
for t in /proc/$pid/task/*; do echo "$(basename $t) $(cat $t/comm) $(cut -d' ' -f14,15 $t/stat)"; done > $O/t1.txt
join <(sort $O/t0.txt | awk '{print $1, $2, $3+$4}') <(sort $O/t1.txt | awk '{print $1, $3+$4}') | awk '{d=$4-$3; if (d>5) print "tid", $1, $2, "ticks in 5 s:", d}' | tee $O/busy.txt
timeout 200 /opt/rocm/bin/rocgdb -iex "set auto-solib-add off" -p $pid -batch -ex "set pagination off" -ex "sharedlibrary amdhip64" -ex "sharedlibrary hsa-runtime" \
    -ex "sharedlibrary libc.so" -ex "sharedlibrary libhsakmt" -ex "thread apply all bt 14" > $O/bt.txt 2>&1
kill $JOB 2>/dev/null
wait $JOB 2>/dev/null
grep -n "LWP\|^#" $O/bt.txt | cut -c1-220 | head -${HEAD:-220}
