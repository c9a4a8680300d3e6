import os, sys
sys.path.insert(0, "/root/repo")
import torch
from causaldiffae_amd import ops
for (B, T, heads, ch) in [(128, 256, 4, 96), (128, 64, 4, 128), (256, 256, 4, 64), (256, 64, 4, 64)]:
    qkv = torch.randn(B, T, 3 * heads * ch, device="cuda:0")
    with torch.no_grad():
        ops.qkv_attention(qkv, heads); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.qkv_attention(qkv, heads)
        e1.record(); torch.cuda.synchronize()
    print(f"attention B={B} T={T} heads={heads} ch={ch}: value {e0.elapsed_time(e1)*100:.1f} us")
