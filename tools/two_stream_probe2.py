"""Dev: cross-stream ordering by STREAM MEMORY OPERATIONS (hipStreamWriteValue32 on the producer stream, hipStreamWaitValue32 on the
consumer stream: executed by the command processor, no host thread) against hipEvent record / wait — per-thread CPU as in
two_stream_probe.py.  FLAGS=<n>: hipSetDeviceFlags(n) before the context exists (2 = yield, 4 = blocking sync)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = ctypes.CDLL("libamdhip64.so")
if os.environ.get("FLAGS"):
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(int(os.environ["FLAGS"])))
import torch
import bench

dev = torch.device("cuda:0")
x = torch.zeros(1 << 22, device=dev)
y = torch.zeros(1 << 22, device=dev)
s2 = torch.cuda.Stream()
main = torch.cuda.current_stream()
hip.hipExtMallocWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
sig, sig2 = ctypes.c_void_p(), ctypes.c_void_p()
rc = hip.hipExtMallocWithFlags(ctypes.byref(sig), 8, 0x2)
rc2 = hip.hipExtMallocWithFlags(ctypes.byref(sig2), 8, 0x2)
print("signal memory rc", rc, rc2, hex(sig.value or 0), hex(sig2.value or 0))
if not sig.value:          # plain device memory instead
    t_ = torch.zeros(64, dtype=torch.int32, device=dev)
    sig, sig2 = ctypes.c_void_p(t_.data_ptr()), ctypes.c_void_p(t_.data_ptr() + 128)
    print("falling back to plain device memory for the flag words")
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipMemset(sig, 0, 8); hip.hipMemset(sig2, 0, 8)
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
seq = [0]


def run(mode, n=20000):
    torch.cuda.synchronize()
    t0, w0 = bench.thread_cpu(), time.perf_counter()
    for i in range(n):
        x.add_(1.0)
        if mode == "dep":
            s2.wait_stream(main)
        elif mode == "val":
            seq[0] += 1
            a = hip.hipStreamWriteValue32(main.cuda_stream, sig, seq[0], 0)
            b = hip.hipStreamWaitValue32(s2.cuda_stream, sig, seq[0], 0, 0xFFFFFFFF)
            assert a == 0 and b == 0, (a, b)
        with torch.cuda.stream(s2):
            y.add_(1.0)
        if mode == "dep":
            main.wait_stream(s2)
        elif mode == "val":
            a = hip.hipStreamWriteValue32(s2.cuda_stream, sig2, seq[0], 0)
            b = hip.hipStreamWaitValue32(main.cuda_stream, sig2, seq[0], 0, 0xFFFFFFFF)
            assert a == 0 and b == 0, (a, b)
        if i % 256 == 255:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    w = time.perf_counter() - w0
    t1 = bench.thread_cpu()
    per = sorted(((name, (cpu - t0.get(tid, ("", 0.0))[1]) / w) for tid, (name, cpu) in t1.items()), key=lambda kv: -kv[1])
    print(f"{mode:6s} wall {w:.2f} s  x={x[0].item():.0f} y={y[0].item():.0f} threads (cores busy):", [(n_, round(v, 2)) for n_, v in per if v > 0.02])


for m in ("dep", "val", "dep", "val"):
    run(m)
