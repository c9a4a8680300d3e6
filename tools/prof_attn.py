"""Dev tool: HIP-event timings of the QKV attention at the sampling shapes (fused forward, batch 128) and at the training shapes
(batch 32: the three-launch forward that keeps the probabilities, the fused forward, and the five-launch backward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causaldiffae_amd import ops


def timed(f, n=10):
    for _ in range(3):      # (the first two backward calls of a shape allocate: 160 ms and 13 ms)
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (B, T, heads, ch) in [(128, 256, 4, 96), (128, 64, 4, 128), (256, 256, 4, 64), (256, 64, 4, 64)]:
    qkv = torch.randn(B, T, 3 * heads * ch, device="cuda:0")
    with torch.no_grad():
        us = timed(lambda: ops.qkv_attention(qkv, heads))
    print(f"sampling  B={B} T={T} heads={heads} ch={ch}: {us:.1f} us")
for (B, T, heads, ch, cnt) in [(32, 256, 4, 96, 5), (32, 64, 4, 128, 6)]:
    qkv = torch.randn(B, T, 3 * heads * ch, device="cuda:0", requires_grad=True)
    with torch.no_grad():
        fused = timed(lambda: ops.qkv_attention(qkv, heads))
    fwd = timed(lambda: ops.qkv_attention(qkv, heads))
    out = ops.qkv_attention(qkv, heads)
    g = torch.randn_like(out)
    bwd = timed(lambda: out.backward(g, retain_graph=True))
    print(f"training  B={B} T={T} heads={heads} ch={ch} x{cnt}/step: forward {fwd:.1f} us (fused, no probabilities: {fused:.1f}), backward {bwd:.1f} us")
