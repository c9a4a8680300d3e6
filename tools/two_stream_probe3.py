"""Dev: what keeps rocr's AsyncEventsLoop thread busy?  GPU-BOUND loops (each kernel ~60 us, so the launch queue is full and every
dependency is pending when it is enqueued); CPU of every thread but the launcher, in cores.  Modes:
  one      one stream, kernels only
  ev       one stream + an event recorded per kernel (what record_stream / the run-ahead throttle do)
  two      two streams alternating, no dependencies
  dep      two streams, wait_stream both ways per pair (hipEventRecord + hipStreamWaitEvent)
  dep1     two streams, side waits for main only (fork), joined once at the end
  val      two streams, stream order links both ways (hipStreamWriteValue32 + hipStreamWaitValue32)
  val1     links, fork only"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
_seq = {}


def check(rc):
    assert rc == 0, rc


def link_create(ref):
    rc = hip.hipExtMallocWithFlags(ref, 8, 0x2)          # hipMallocSignalMemory
    hip.hipMemset(ref._obj, 0, 8)
    return rc


def link_order(link, prod, cons):
    v = _seq[link.value] = _seq.get(link.value, 0) + 1
    return hip.hipStreamWriteValue32(prod, link, v, 0) or hip.hipStreamWaitValue32(cons, link, v, 0, 0xFFFFFFFF)

dev = torch.device("cuda:0")
x = torch.zeros(1 << 25, device=dev)
y = torch.zeros(1 << 25, device=dev)
s2 = torch.cuda.Stream()
main = torch.cuda.current_stream()
fork, join = ctypes.c_void_p(), ctypes.c_void_p()
check(link_create(ctypes.byref(fork)))
check(link_create(ctypes.byref(join)))
me = int(open("/proc/thread-self/stat").read().split()[0])


def run(mode, n=4000):
    torch.cuda.synchronize()
    t0, w0 = bench.thread_cpu(), time.perf_counter()
    evs = []
    for i in range(n):
        x.add_(1.0)
        if mode == "one":
            y.add_(1.0)
        elif mode == "ev":
            y.add_(1.0)
            e = torch.cuda.Event(); e.record(); evs.append(e)
            if len(evs) > 64:
                evs.pop(0).query()
        else:
            if mode in ("dep", "dep1"):
                s2.wait_stream(main)
            elif mode in ("val", "val1"):
                check(link_order(fork, main.cuda_stream, s2.cuda_stream))
            with torch.cuda.stream(s2):
                y.add_(1.0)
            if mode == "dep":
                main.wait_stream(s2)
            elif mode == "val":
                check(link_order(join, s2.cuda_stream, main.cuda_stream))
    main.wait_stream(s2)
    torch.cuda.synchronize()
    w = time.perf_counter() - w0
    t1 = bench.thread_cpu()
    per = sorted(((name, tid, (cpu - t0.get(tid, ("", 0.0))[1]) / w) for tid, (name, cpu) in t1.items()), key=lambda kv: -kv[2])
    print(f"{mode:5s} wall {w:.2f} s ({1e6 * w / n:.0f} us / iteration)  launcher {sum(v for _, t, v in per if t == me):.2f}  others:",
          [(n_, round(v, 2)) for n_, t, v in per if v > 0.02 and t != me], flush=True)


for m in (os.environ["MODES"].split(",") if os.environ.get("MODES") else ("one", "ev", "two", "dep", "dep1", "val", "val1", "one")):
    run(m)
