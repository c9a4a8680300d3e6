"""Dev: which Python-level operation makes which HIP runtime calls — run one candidate 2000 times under `rocprofv3 --hip-trace --stats`
and read the call counts (hipGetDeviceCount at ~6 us a call showed up 740 times per training step).
    WHICH=empty rocprofv3 --hip-trace --stats --output-format csv -d out -o t -- python3 tools/hip_api_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import _lib
dev = torch.device("cuda:0")
x = torch.zeros(16, device=dev, requires_grad=True)
torch.cuda.synchronize()
which = os.environ.get("WHICH", "none")
N = 2000


class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        return a * 1.0

    @staticmethod
    def backward(ctx, g):
        return g


for _ in range(N):
    if which == "current_device":
        torch.cuda.current_device()
    elif which == "raw_stream":
        _lib.stream()
    elif which == "empty":
        torch.empty((4,), dtype=torch.float32, device=dev)
    elif which == "empty_strided":
        torch.empty_strided((2, 2), (2, 1), dtype=torch.float32, device=dev)
    elif which == "data_ptr":
        x.data_ptr()
    elif which == "autograd":
        F.apply(x).sum().backward()
    elif which == "ctypes":
        _lib.lib.cdae_version()
    elif which == "device_guard":
        with torch.cuda.device(0):
            pass
torch.cuda.synchronize()
print("done", which)
