"""Dev tool: a long seeded fuzz of the window weight gradient's geometry (both plane counts, every prefetch distance / LDS layout) against fp64 —
the test suite runs two seeds of this; this runs SEEDS of them (default 40) in one process."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from causaldiffae_amd._lib import check, lib, ptr, stream, splitk_ws, SPLITK_BYTES, precision_scope, tune_scope
from test_gpu_kernels import _wgrad_ref
dev = "cuda:0"
worst = 0.0
n = 0
for seed in range(int(os.environ.get("SEEDS", "40"))):
    rng = random.Random(9000 + seed)
    g = torch.Generator(device=dev).manual_seed(seed)
    W = rng.choice([8, 16, 32, 64])
    H = rng.choice([h for h in (8, 16, 32, 64) if (h * W) % 64 == 0 and h * W <= 4096])
    N = rng.randint(1, 5 if H * W >= 1024 else 40)
    Cin, Cout, acc = 64 * rng.randint(1, 5), 64 * rng.randint(1, 5), rng.randint(0, 1)
    mode = rng.choice(["f16x3", "mixed16"])
    dist, swz, co2 = rng.choice([1, 2]), rng.choice([0, 1]), rng.choice([0, 1])
    a = torch.randn(N, H, W, Cin, device=dev, generator=g)
    dy = torch.randn(N, H, W, Cout, device=dev, generator=g) * 1e-3
    ap = torch.empty((2, N, H, W, Cin), dtype=torch.bfloat16, device=dev); dp = torch.empty((2, N, H, W, Cout), dtype=torch.bfloat16, device=dev)
    check(lib.cdae_split_bf16(ptr(a), ptr(ap[0]), ptr(ap[1]), a.numel(), stream()))
    check(lib.cdae_split_bf16(ptr(dy), ptr(dp[0]), ptr(dp[1]), dy.numel(), stream()))
    dw0 = torch.randn(Cout, 3, 3, Cin, device=dev, generator=g) * 1e-2; db0 = torch.randn(Cout, device=dev, generator=g) * 1e-2
    dw, db = dw0.clone(), db0.clone()
    with precision_scope(mode), tune_scope(wgwin_dist=dist, wgwin_swz=swz, wgwin_co2=2 * co2):
        check(lib.cdae_conv3x3_wgrad_win(ptr(ap[0]), ptr(ap[1]), ptr(dp[0]), ptr(dp[1]), ptr(dw), ptr(db), N, H, W, Cin, Cout, acc, ptr(splitk_ws(torch.device(dev))), SPLITK_BYTES, stream()))
    planes = 1 if mode == "mixed16" else 2
    a_q = sum(ap[i].float() for i in range(planes)).permute(0, 3, 1, 2); dy_q = sum(dp[i].float() for i in range(planes)).permute(0, 3, 1, 2)
    ref_w, ref_b = _wgrad_ref(a_q, dy_q)
    if acc: ref_w, ref_b = ref_w + dw0.double(), ref_b + db0.double()
    e = (dw.double() - ref_w).abs().max().item() / ref_w.abs().max().item()
    eb = (db.double() - ref_b).abs().max().item() / ref_b.abs().max().item()
    worst = max(worst, e, eb); n += 1
    assert e < 3e-5 and eb < 3e-5, (seed, (N, H, W, Cin, Cout, acc, mode, dist, swz, co2), e, eb)
print("wgwin fuzz: %d cases, worst relative error %.2e (bar 3e-5)" % (n, worst))
