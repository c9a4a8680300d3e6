import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unet_ref as U
from oracle.closed_form import fill_state_dict, fill_value, synth
from improved_diffusion.nn import GaussianConvEncoder
from improved_diffusion.unet import encoder_hidden_dims
DEV = "cuda:0"
for tag, C, S, nv in [("enc64", 4, 64, 4), ("enc96", 4, 96, 4)]:
    dims = encoder_hidden_dims(S, nv)
    enc = GaussianConvEncoder(C, 512, hidden_dims=dims, num_vars=nv)
    sd0 = enc.state_dict()
    enc.load_state_dict({k: fill_value("rep_emb." + k, v.shape) for k, v in sd0.items()})
    enc.to(DEV).train()
    x = synth(tag + ".x", (4, C, S, S), 0.0, 1.0)
    xg = x.to(DEV).requires_grad_(True)
    mu, var = enc.encode(xg)
    gmu, gvar = synth(tag + ".gmu", (4, 512)), synth(tag + ".gvar", (4, 512))
    ((mu * gmu.to(DEV)).sum() + (var * gvar.to(DEV)).sum()).backward()
    cfg = U.default_cfg(image_size=64, in_channels=C, n_vars=nv, rep_cond=True, encoder_dims=dims)
    spec = [(k, s) for k, s in U.param_spec(cfg) if k.startswith("rep_emb.")]
    sd = fill_state_dict(spec)
    for v in sd.values():
        if v.dtype == torch.float32: v.requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    m2, v2 = U.encode(sd, xc, len(dims), training=True)
    ((m2 * gmu).sum() + (v2 * gvar).sum()).backward()
    print(tag, "mu err", (mu.cpu() - m2).abs().max().item(), "gx err", (xg.grad.cpu() - xc.grad).abs().max().item(), "gx max", xc.grad.abs().max().item())
    d = (xg.grad.cpu() - xc.grad).abs()
    bad = (d > 1e-3).nonzero()
    print("  n bad", bad.shape[0], "of", d.numel(), "first", bad[:8].tolist())
    for k, p in enc.named_parameters():
        ref = sd["rep_emb." + k].grad
        e = (p.grad.cpu() - ref).abs().max().item()
        print(f"  {k:28s} err {e:.3e}  max {ref.abs().max().item():.3e}")
