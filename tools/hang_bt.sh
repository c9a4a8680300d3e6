#!/bin/bash
# Dev tool: start an N-rank gloo bench on one GPU, wait, and dump the native stacks of every rank that is still running
# (used to find the 4-rank stall of the training leg with the weight-gradient side stream off).
#   gpurun --timeout 400 -- 'bash tools/hang_bt.sh 4 CDAE_WGRAD_STREAM=0'
N=${1:-4}; shift
O=gpurun_out/hang_bt; mkdir -p $O
env "$@" CDAE_WATCHDOG_S=260 CDAE_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29574 \
    bench.py --gpus $N --steps 2 --warmup 1 --regions 1 --batch 8 --train-batch 8 --train-steps 4 --no-cpu-baseline --no-fp32 --no-extra > $O/out.txt 2> $O/err.txt &
JOB=$!
sleep 75
for pid in $(pgrep -P $JOB); do
    echo "== pid $pid $(tr '\0' ' ' < /proc/$pid/cmdline | cut -c1-80)" >> $O/bt.txt
    for t in /proc/$pid/task/*; do echo "$(cat $t/comm) $(grep State $t/status | cut -f2) $(cut -d' ' -f14,15 $t/stat)"; done >> $O/threads_$pid.txt
    # (symbols of the HIP / HSA runtimes and libc only: reading libtorch's takes minutes)
    timeout 150 /opt/rocm/bin/rocgdb -iex "set auto-solib-add off" -p $pid -batch -ex "set pagination off" -ex "sharedlibrary amdhip64" -ex "sharedlibrary hsa-runtime" \
        -ex "sharedlibrary libc.so" -ex "sharedlibrary libc10" -ex "thread apply all bt 16" >> $O/bt.txt 2>&1
    break        # one rank is enough (and detaching lets it continue)
done
sleep 5
kill $JOB 2>/dev/null
wait $JOB 2>/dev/null
grep -c . $O/bt.txt
