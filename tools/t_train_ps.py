import os, sys, torch
sys.path.insert(0, "/root/repo")
from causaldiffae_amd import ops
torch.manual_seed(0)
dev = "cuda:0"
def run(N, C, Cout, H, ss_on, res_on, fused):
    os.environ["X"] = "1"
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(N, C, H, H, device=dev, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_()
    gamma = (1 + 0.1 * torch.randn(C, device=dev, generator=g)).requires_grad_()
    beta = (0.1 * torch.randn(C, device=dev, generator=g)).requires_grad_()
    ss = (0.2 * torch.randn(N, 2 * C, device=dev, generator=g)).requires_grad_() if ss_on else None
    w = (torch.randn(Cout, C, 3, 3, device=dev, generator=g) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last).requires_grad_()
    b = (0.1 * torch.randn(Cout, device=dev, generator=g)).requires_grad_()
    res = torch.randn(N, Cout, H, H, device=dev, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_() if res_on else None
    dy = torch.randn(N, Cout, H, H, device=dev, generator=g).contiguous(memory_format=torch.channels_last) * 1e-3
    if fused == 2:       # fp64 torch
        xd, gd, bd, wd, b2 = [t.detach().double().requires_grad_() for t in (x, gamma, beta, w, b)]
        ssd = ss.detach().double().requires_grad_() if ss_on else None
        rd = res.detach().double().requires_grad_() if res_on else None
        h = torch.nn.functional.group_norm(xd, 32, gd, bd, 1e-5)
        if ss_on:
            h = h * (1 + ssd[:, :C, None, None]) + ssd[:, C:, None, None]
        h = torch.nn.functional.silu(h)
        out = torch.nn.functional.conv2d(h, wd, b2, padding=1)
        if res_on: out = out + rd
        out.backward(dy.double())
        return [out.detach()] + [t.grad for t in (xd, gd, bd, wd, b2)] + ([ssd.grad] if ss_on else []) + ([rd.grad] if res_on else [])
    if fused:
        out = ops.gn_conv3x3(x, gamma, beta, ss, w, b, res, True, 32, 1e-5)
    else:
        h = ops.group_norm(x, gamma, beta, ss, True, 32, 1e-5)
        out = ops.conv3x3(h, w, b, res)
    out.backward(dy)
    return [out.detach()] + [t.grad for t in (x, gamma, beta, w, b)] + ([ss.grad] if ss_on else []) + ([res.grad] if res_on else [])
names = ["out", "dx", "dgamma", "dbeta", "dw", "db", "dss/dres", "dres"]
for cfg in [(4, 128, 128, 64, True, True), (8, 256, 256, 32, False, False), (8, 384, 384, 16, True, True), (16, 512, 512, 8, True, False), (2, 128, 256, 32, True, True), (32, 128, 128, 64, True, True)]:
    a = run(*cfg, 1); b = run(*cfg, 0); c = run(*cfg, 2)
    print(cfg)
    for n, u, v, r in zip(names, a, b, c):
        sc = r.abs().max().item() + 1e-30
        print(f"   {n:8s} fused-vs-f64 {((u.double() - r).abs().max().item() / sc):.2e}   old-vs-f64 {((v.double() - r).abs().max().item() / sc):.2e}")
