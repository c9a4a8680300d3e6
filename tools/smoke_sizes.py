"""Dev tool: forward / DDIM step / training step at the real dataset resolutions (Pendulum 96 px C=4, CausalCircuit 128 px C=3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import causaldiffae_amd  # noqa
from improved_diffusion import script_util as su
import bench

for (S, C, nv, N) in [(96, 4, 4, 4), (128, 3, 4, 2), (32, 1, 2, 8)]:
    cfg = {**su.model_and_diffusion_defaults(), "image_size": S, "in_channels": C, "n_vars": nv, "rep_cond": True,
           "causal_modeling": True, "timestep_respacing": "ddim10", "class_cond": C == 1}
    model, diff = su.create_model_and_diffusion(**cfg)
    bench.randomize(model, 7)
    model.to("cuda:0")
    x0 = torch.rand(N, C, S, S, device="cuda:0")
    c = torch.rand(N, nv, device="cuda:0")
    kw = {"c": c}
    if C == 1:
        kw["y"] = torch.randint(0, 10, (N,), device="cuda:0")
    t = torch.randint(0, diff.num_timesteps, (N,), device="cuda:0")
    model.train()
    terms = diff.training_losses(model, x0, t, model_kwargs=dict(kw), rep_cond=True, causal_modeling=True)
    terms["loss"].mean().backward()
    model.eval()
    with torch.no_grad():
        mu, var = model.rep_emb.encode(x0)
        skw = {k: v for k, v in kw.items() if k != "c"}
        skw["z"] = mu
        out = diff.ddim_sample_loop(model, (N, C, S, S), model_kwargs=skw, use_graph=True)
        out2 = diff.ddim_sample_loop(model, (N, C, S, S), noise=torch.zeros(N, C, S, S, device="cuda:0"), model_kwargs=skw)
        out3 = diff.ddim_sample_loop(model, (N, C, S, S), noise=torch.zeros(N, C, S, S, device="cuda:0"), model_kwargs=skw, use_graph=True)
    torch.cuda.synchronize()
    print(S, C, "loss", float(terms["loss"].mean()), "sample finite", bool(torch.isfinite(out).all()), "graph==eager", bool(torch.equal(out2, out3)))
