"""Host-side logic of the product package (no GPU): API surface, state-dict layout, integer-exact respacing,
f64 schedule tables, and the C-ABI library exporting every symbol include/cdae.h declares."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from causaldiffae_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "cdae.h")).read()
    declared = set(re.findall(r"\b(cdae_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libcdae.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.lib.cdae_version() == 1


def test_ops_refuse_cpu_tensors():
    from causaldiffae_amd import ops
    from causaldiffae_amd._lib import CdaeError
    with pytest.raises(CdaeError):
        ops.silu(torch.zeros(4))


def test_script_util_api_surface():
    from improved_diffusion import script_util as su
    d = su.model_and_diffusion_defaults()
    assert list(d) == ["image_size", "num_channels", "num_res_blocks", "num_heads", "num_heads_upsample", "attention_resolutions",
                       "dropout", "learn_sigma", "sigma_small", "class_cond", "diffusion_steps", "noise_schedule",
                       "timestep_respacing", "use_kl", "predict_xstart", "rescale_timesteps", "rescale_learned_sigmas",
                       "use_checkpoint", "use_scale_shift_norm", "context_cond", "rep_cond", "n_vars", "causal_modeling",
                       "flow_based", "in_channels", "masking"]
    assert d["image_size"] == 64 and d["num_channels"] == 128 and d["attention_resolutions"] == "16,8" and d["rescale_timesteps"]
    assert list(inspect.signature(su.create_model_and_diffusion).parameters) == [
        "image_size", "class_cond", "learn_sigma", "sigma_small", "num_channels", "num_res_blocks", "num_heads",
        "num_heads_upsample", "attention_resolutions", "dropout", "diffusion_steps", "noise_schedule", "timestep_respacing",
        "use_kl", "predict_xstart", "rescale_timesteps", "rescale_learned_sigmas", "use_checkpoint", "use_scale_shift_norm",
        "context_cond", "rep_cond", "n_vars", "causal_modeling", "flow_based", "in_channels", "masking"]
    assert su.NUM_CLASSES == 10 and su.str2bool("yes") and not su.str2bool("0")
    import argparse
    p = argparse.ArgumentParser()
    su.add_dict_to_argparser(p, d)
    a = p.parse_args(["--image_size", "32", "--rep_cond", "True"])
    assert su.args_to_dict(a, ["image_size", "rep_cond"]) == {"image_size": 32, "rep_cond": True}


MODEL_CFG = {
    "M32": dict(image_size=32, in_channels=1, n_vars=2, class_cond=True),
    "P64": dict(image_size=64, in_channels=4, n_vars=4),
    "C64": dict(image_size=64, in_channels=3, n_vars=4),
}


@pytest.mark.parametrize("tag", ["M32", "P64", "C64"])
def test_state_dict_layout_matches_reference(golden, tag):
    from improved_diffusion import script_util as su
    g = golden("g6_unet.npz")
    cfg = {**su.model_and_diffusion_defaults(), "rep_cond": True, "causal_modeling": True, **MODEL_CFG[tag]}
    model, diff = su.create_model_and_diffusion(**cfg)
    sd = model.state_dict()
    assert list(sd.keys()) == list(g[f"{tag}/keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g[f"{tag}/shapes"])
    assert sum(p.numel() for p in model.parameters()) == int(g[f"{tag}/n_params"])
    # 3x3 conv weights are stored OHWI (channels_last) under the reference's logical shape; a reference-layout
    # state dict loads into them
    w = model.input_blocks[1][0].in_layers[2].weight
    assert w.permute(0, 2, 3, 1).is_contiguous()
    model.load_state_dict({k: torch.zeros(v.shape, dtype=v.dtype) for k, v in sd.items()})
    assert model.input_blocks[1][0].in_layers[2].weight.permute(0, 2, 3, 1).is_contiguous()


@pytest.mark.parametrize("rs", ["", "ddim100", "ddim250", "250", "100", "ddim50", "10,10,10"])
def test_schedule_tables_bit_exact(golden, rs):
    from improved_diffusion import script_util as su
    g = golden("g1_schedules.npz")
    d = su.create_gaussian_diffusion(steps=1000, timestep_respacing=rs, rescale_timesteps=True)
    tag = rs or "full"
    np.testing.assert_array_equal(np.array(d.timestep_map, dtype=np.int64), g[f"{tag}/timestep_map"])
    for n in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
              "posterior_mean_coef2"]:
        np.testing.assert_array_equal(getattr(d, n), g[f"{tag}/{n}"], err_msg=n)
    np.testing.assert_array_equal(d._model_variance, g[f"{tag}/fixed_large_variance"])
    assert d.num_timesteps == len(g[f"{tag}/betas"]) and d.kl_weight == 0.0


def test_space_timesteps_exact(golden):
    from improved_diffusion.respace import space_timesteps
    g = golden("g1_schedules.npz")
    for key in [k for k in g.files if k.startswith("space/") and k != "space/errors"]:
        _, T, spec = key.split("/")
        arg = spec if (spec.startswith("ddim") or "," not in spec) else [int(s) for s in spec.split(",")]
        np.testing.assert_array_equal(np.array(sorted(space_timesteps(int(T), arg)), dtype=np.int64), g[key], err_msg=key)
    for (T, spec), bad in zip([(1000, "ddim7"), (10, "20"), (1000, "ddim999")], g["space/errors"]):
        if bad:
            with pytest.raises(ValueError):
                space_timesteps(T, spec)
        else:
            space_timesteps(T, spec)


def test_uniform_sampler_matches_numpy_stream():
    from improved_diffusion import script_util as su
    from improved_diffusion.resample import UniformSampler, create_named_schedule_sampler
    d = su.create_gaussian_diffusion(steps=1000)
    s = create_named_schedule_sampler("uniform", d)
    assert isinstance(s, UniformSampler)
    np.random.seed(3)
    t, w = s.sample(16, torch.device("cpu"))
    np.random.seed(3)
    ref = np.random.choice(1000, size=(16,), p=np.ones(1000) / 1000)
    np.testing.assert_array_equal(t.numpy(), ref)
    assert t.dtype == torch.int64 and w.dtype == torch.float32 and torch.all(w == 1)


def test_loss_second_moment_resampler():
    """resample.py:127-154 by definition: uniform until warmed up, then sqrt(second moment) mixed with a 0.1 % uniform floor."""
    from causaldiffae_amd.resample import LossAwareSampler, LossSecondMomentResampler, create_named_schedule_sampler

    class Diff:
        num_timesteps = 4

    s = create_named_schedule_sampler("loss-second-moment", Diff)
    assert isinstance(s, LossSecondMomentResampler) and isinstance(s, LossAwareSampler)
    assert np.array_equal(s.weights(), np.ones(4))
    hist = {t: [] for t in range(4)}
    rng = np.random.RandomState(0)
    for k in range(13):                       # 13 > history: the oldest three values per step must drop out
        ts = torch.arange(4)
        ls = torch.from_numpy(rng.rand(4).astype(np.float32) * (1 + ts.numpy()))
        if k == 5:
            assert np.array_equal(s.weights(), np.ones(4))          # not warmed up yet
        s.update_with_local_losses(ts, ls)
        for t, l in zip(ts.tolist(), ls.tolist()):
            hist[t] = (hist[t] + [l])[-10:]
    want = np.sqrt(np.array([np.mean(np.square(hist[t])) for t in range(4)]))
    want = want / want.sum() * (1 - 0.001) + 0.001 / 4
    assert np.allclose(s.weights(), want, rtol=1e-10, atol=0)      # ring vs shifted history: summation order only
    np.random.seed(3)
    idx, w = s.sample(64, "cpu")
    p = want / want.sum()
    assert idx.dtype == torch.int64 and np.allclose(w.numpy(), 1 / (4 * p[idx.numpy()]), rtol=1e-6)


def test_metrics_dci_and_mcc_on_known_representations():
    """causaldiffae_amd.metrics (the `mt._compute_dci` / `mt.MCC` surface of the reference's evaluation script): a representation that
    is a permutation of the factors is perfectly disentangled and complete; a fully mixed one is not."""
    import numpy as np
    from improved_diffusion import metrics as mt
    rng = np.random.RandomState(0)
    ys = rng.randint(0, 5, size=(3, 400)).astype(np.float64)             # [factors, points]
    clean = ys[[2, 0, 1]] + 0.01 * rng.randn(3, 400)                      # each code = one factor
    mixed = np.stack([ys.sum(0), ys.sum(0) + 0.01 * rng.randn(400), ys.sum(0) - 0.01 * rng.randn(400)])
    s, imp, share = mt._compute_dci(clean[:, :300], ys[:, :300], clean[:, 300:], ys[:, 300:])
    assert imp.shape == (3, 3) and abs(share.sum() - 1) < 1e-9
    assert s["disentanglement"] > 0.95 and s["completeness"] > 0.95
    s2, _, _ = mt._compute_dci(mixed[:, :300], ys[:, :300], mixed[:, 300:], ys[:, 300:])
    assert s2["disentanglement"] < 0.5
    assert set(s) == {"informativeness_train", "informativeness_test", "disentanglement", "completeness"}
    z = rng.randn(500, 4)
    assert mt.MCC(z, z[:, [3, 1, 0, 2]] * np.array([1.0, -2.0, 0.5, 3.0])) > 0.999
    assert mt.MCC(z, rng.randn(500, 4)) < 0.3


def test_conv_weight_bank_packs_per_weight():
    """ops.ConvWeightBank over a real UNet's flat parameter buffer: the stem conv [128, 3, 3, 3] and the head conv [3, 128, 3, 3] sit
    in the bank beside the ResBlock convs; K-group-major planes (convwin_kernel's weight layout) are a PER-WEIGHT property — one
    un-packable weight must not switch them off for the whole model (round-2 advisor finding)."""
    from causaldiffae_amd import ops
    from improved_diffusion import script_util as su
    cfg = {**su.model_and_diffusion_defaults(), "image_size": 64, "in_channels": 3, "n_vars": 4, "rep_cond": True, "causal_modeling": True,
           "num_channels": 32, "num_res_blocks": 1}
    model, _ = su.create_model_and_diffusion(**cfg)
    params = [p for p in model.parameters()]
    flat = torch.empty(sum(p.numel() for p in params))
    off = 0
    for p in params:
        view = flat.as_strided(p.shape, p.stride(), off)
        view.copy_(p.data)
        p.data = view
        off += p.numel()
    ws = [p for p in params if p.dim() == 4 and tuple(p.shape[2:]) == (3, 3) and p.permute(0, 2, 3, 1).is_contiguous() and p.numel() % 8 == 0
          and ((p.data_ptr() - flat.data_ptr()) // 4) % 8 == 0]
    bank = ops.ConvWeightBank(flat, ws)
    named = dict(model.named_parameters())
    stem, res = named["input_blocks.0.0.weight"], named["input_blocks.1.0.in_layers.2.weight"]
    assert any(w is stem for w in ws) and any(w is res for w in ws)
    assert bank.kpack and id(res) in bank.packable and id(stem) not in bank.packable
    assert bank.packed(stem, False) == (None, None)
    import numpy as np
    flags = bank.desc.numpy().view(np.dtype([("off", "<i8"), ("cout", "<i4"), ("cin", "<i4"), ("tile0", "<i4"), ("flags", "<i4")]))["flags"]
    assert [int(f) & 1 for f in flags] == [1 if (w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0) else 0 for w in ws]
    # bits 8..: the weight's record in the scale table (cdae_wprep_all_k scales the f16 planes by 2^k of that record)
    assert bank.scales is not None and [int(f) >> 8 for f in flags] == [bank.scales.index[id(w)] for w in ws]
    chunk = ops.lib.cdae_weight_scales_chunk()
    sdesc = bank.scales.desc.numpy().view(np.dtype([("off", "<i8"), ("n", "<i8"), ("chunk0", "<i4"), ("pad", "<i4")]))
    assert [int(o) for o in sdesc["off"]] == [(w.data_ptr() - flat.data_ptr()) // 4 for w in ws] and [int(n) for n in sdesc["n"]] == [w.numel() for w in ws]
    assert [int(c) for c in sdesc["chunk0"]] == list(np.cumsum([0] + [(w.numel() + chunk - 1) // chunk for w in ws])[:-1])
    assert bank.scales.chunks == sum((w.numel() + chunk - 1) // chunk for w in ws)


def test_optimizer_checkpoint_uses_torch_adamw_layout():
    """`opt<step>.pt` is read and written by the reference with torch.optim.AdamW.state_dict() / load_state_dict()
    (train_util.py:159-169, 340-343): what FusedAdamWEMA.torch_state_dict() emits must load into a torch AdamW over the same model."""
    from causaldiffae_amd.train_util import FusedAdamWEMA
    m = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Linear(8, 2))
    opt = FusedAdamWEMA.__new__(FusedAdamWEMA)
    from causaldiffae_amd.train_util import FlatParams
    opt.model, opt.flat = m, FlatParams(m)
    opt.lr, opt.weight_decay, opt.betas, opt.eps, opt.t = 1e-4, 0.0, (0.9, 0.999), 1e-8, 3
    opt.m, opt.v = torch.arange(opt.flat.numel, dtype=torch.float32), torch.arange(opt.flat.numel, dtype=torch.float32) * 2
    sd = opt.torch_state_dict()
    ref = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.0)
    ref.load_state_dict(sd)
    got = ref.state_dict()["state"]
    for i, p in enumerate(m.parameters()):
        o = opt.flat.offsets[[id(q) for q in opt.flat.params].index(id(p))]
        assert torch.equal(got[i]["exp_avg"].flatten(), opt.m[o:o + p.numel()])
        assert torch.equal(got[i]["exp_avg_sq"].flatten(), opt.v[o:o + p.numel()])
        assert float(got[i]["step"]) == 3.0


def test_window_conv_k_loop_has_no_compiler_drain():
    """convwin_kernel (the dominant kernel of both benchmark legs) runs at the 256-VGPR limit of two waves per SIMD.  When hipcc's
    register allocator spills a value that the K loop uses, the reload — a vector-memory load — gets `s_waitcnt vmcnt(0)` in front of
    its first use INSIDE the loop: a full drain of the weight / window LDS-DMAs every K step (round 2's build had one at the loop top).
    The loop must contain no compiler-inserted vmcnt wait and no scratch traffic, in all three instantiations (tools/isa_lint.py)."""
    import importlib.util, shutil
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isa_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kernels = [r for r in mod.lint("convwin") if "convwin_kernel" in r["kernel"]]
    # <f16 | bf16, 9 taps> x <plane pairs | one plane>, <f16 | bf16, 4 taps, pairs>, the 96- and 64-column tiles <f16, 9, pairs, 3 | 2>, and the 16-bit
    # torso's <bf16, 9, one plane, bf16 rows out> and <bf16, 9, channel halves in the two plane slots, bf16 rows out> (PAIR: two MFMAs per product pair)
    assert len(kernels) == 10
    for r in kernels:
        assert r["loops"], r["kernel"]
        args = [a.strip() for a in r["kernel"].split("convwin_kernel<")[1].split(">")[0].split(",")]      # BF, taps, planes, column tiles per wave, IO16, PAIR
        want = 8 * int(args[3]) * ((2 if args[5] == "true" else 3) if args[2] == "2" else 1)
        for lp in r["loops"]:
            assert lp["mfmas"] == want and lp["barriers"] == 1, (r["kernel"], lp)
            assert not lp["vmcnt_waits"] and not lp["scratch"], (r["kernel"], lp)
        # (the bf16-row instantiations' channel-major epilogue was written against the allocator: 2 / 0 spilled values, outside the K loop)
        assert 0 <= r["vgpr_spills"] <= (4 if args[4] == "true" else 16), (r["kernel"], r["vgpr_spills"])



def test_wgrad_window_kernel_step_loop_is_clean():
    """wgwin_kernel (weight gradients: a sixth of the training step) keeps 144 accumulator registers, eight dy fragments and a three-deep
    activation fragment pipeline in 242 VGPRs: no spills at all, and its step loop — one barrier per 64-pixel step — holds the 9 taps x
    4 sub-tiles x (3 | 1) MFMAs (+ the bias-gradient MFMAs) with no compiler-inserted vmcnt wait (the kernel's own counted wait is inline asm)."""
    import importlib.util, shutil
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isa_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kernels = [r for r in mod.lint("wgrad") if "wgwin_kernel" in r["kernel"]]
    assert len(kernels) == 3          # hi / lo plane pairs, the single-plane mixed16 form, and its 128-output-channel block tile
    for r in kernels:
        single = "<1, " in r["kernel"]
        co2 = "<1, true>" in r["kernel"]
        assert r["vgpr_spills"] == 0 and r["scratch_ops"] == 0, (r["kernel"], r["vgpr_spills"])
        steps = [lp for lp in r["loops"] if lp["barriers"] == 1]
        assert len(steps) == 1, (r["kernel"], r["loops"])
        lp = steps[0]
        assert lp["mfmas"] == (72 + 8 if co2 else 36 + 4 if single else 108 + 8), (r["kernel"], lp)
        assert not lp["vmcnt_waits"] and not lp["scratch"], (r["kernel"], lp)


def test_entry_sweep_never_copies_registers_of_loads_in_flight():
    """skipgn_kernel (ResBlock entry sweeps, GroupNorm -> qkv, the streaming 1x1 GEMMs) requests its fp32 rows two steps ahead from
    inline assembly and waits for them with its own counted `s_waitcnt`.  The compiler does not know those registers are in flight: in
    one build variant of round 4 (whose turn a step is, left visible to the optimiser: -DSG_BALANCE=2) the register allocator tied the
    wait's "+v" operands to other registers and COPIED the sixteen x registers in front of the wait — garbage planes and NaN sums on
    the GPU, non-deterministically.  The shipped build must contain no such copy in any instantiation; the known-bad variant must be
    caught by the same check (so the check itself is known to see the pattern)."""
    import importlib.util, shutil
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isa_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    good = mod.inflight_copies("skipgn", "skipgn_kernel")
    assert len(good) == 3, list(good)                  # <plain | normalised A> f16, and the bf16 streaming dgrad form
    for name, r in good.items():
        assert r["asm_load_registers"] == 32, (name, r)      # two register sets of 16: the steps kt and kt + 1
        assert not r["copies"], (name, r["copies"][:4])
    bad = mod.inflight_copies("skipgn", "skipgn_kernel", extra=("-DSG_BALANCE=2",))
    assert any(r["copies"] for r in bad.values()), "the known-bad variant no longer shows the pattern: the check may have gone blind"
