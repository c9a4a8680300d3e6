"""ctypes binding of libcdae.so (include/cdae.h).  The HIP library is mandatory: importing this module
without it raises, and calling any op on a non-GPU tensor raises — there is no CPU fallback."""
import ctypes
import os

# ROCm runtime default, set before the HIP runtime reads its settings (its first API call): ROCclr releases a queue's completed commands in
# batches, each batch end being a completion callback served by rocr's AsyncEventsLoop thread.  At the default batch size that thread is
# busy for 23 of the 27.5 ms of a training step (~1000 launches, the host one step ahead): a second host core per rank that eight ranks
# under a 16-core quota do not have.  With batches of 32768 commands it is 1-3 ms (measured, tools/host_env_sweep2.sh; docs/NOTES.md
# round 6).  A user's own setting wins; a process that initialised HIP before importing this package keeps the runtime's default.
os.environ.setdefault("DEBUG_CLR_MAX_BATCH_SIZE", "32768")

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcdae.so")

P = ctypes.c_void_p
I = ctypes.c_int
L = ctypes.c_long
LL = ctypes.c_longlong
F = ctypes.c_float
D = ctypes.c_double
SZ = ctypes.c_size_t

# name -> argtypes (return type is int unless listed in _RESTYPES); mirrors include/cdae.h one to one
SIGNATURES = {
    "cdae_version": [],
    "cdae_last_error": [],
    "cdae_workspace_bytes": [I, P, I],
    "cdae_range_status": [P],
    "cdae_set_default_precision": [I],
    "cdae_get_default_precision": [],
    "cdae_conv3x3_fwd": [P, L, L, L, L, P, P, P, P, P, L, I, I, I, I, I, I, I, I, P, SZ, P],
    "cdae_conv3x3_dgrad": [P, L, P, P, L, I, I, I, I, I, I, I, I, P, SZ, P],
    "cdae_conv3x3_wgrad": [P, L, L, L, L, P, L, P, P, I, I, I, I, I, I, I, I, P, SZ, P],
    "cdae_linear_fwd": [P, L, P, L, P, P, P, P, L, P, P, I, I, I, F, I, P, SZ, P],
    "cdae_conv3x3_fwd_ps": [P, P, L, L, L, P, P, P, P, P, P, L, I, P, P, P, I, I, I, I, I, I, I, P, SZ, P],
    "cdae_conv3x3_fwd_psk": [P, P, L, L, L, P, P, P, P, P, P, P, P, L, I, P, P, P, I, I, I, I, I, I, I, P, SZ, P],
    "cdae_conv3x3_dgrad_psk": [P, P, P, P, P, P, P, L, I, I, I, I, I, P, SZ, P],
    "cdae_conv_wpack": [P, P, P, P, I, I, I, P],
    "cdae_convwin_lds_probe": [P],
    "cdae_gn_coef": [P, P, P, P, P, I, P, I, I, I, P],
    "cdae_gn_stats_from_parts": [P, I, I, P, I, I, I, I, I, F, P, P, P, P],
    "cdae_gn_stats_from_parts_coef": [P, I, I, P, I, I, I, I, I, F, P, P, P, P, P, I, P, P, P],
    "cdae_gn_stats2_coef": [P, I, P, I, I, I, I, I, I, F, P, P, P, P, P, I, P, P, P],
    "cdae_upconv3x3_fwd_ps": [P, P, L, L, L, P, P, P, P, P, L, P, I, I, I, I, I, P, SZ, P],
    "cdae_linear_fwd_ps": [P, P, L, P, P, L, P, P, P, P, L, I, I, I, F, I, P, SZ, P],
    "cdae_split_f16": [P, P, P, L, P],
    "cdae_split_f16w": [P, P, P, P, L, P],
    "cdae_weight_scales": [P, P, I, I, P, P, P],
    "cdae_weight_scales_chunk": [],
    "cdae_weight_scale1": [P, L, P, P, P],
    "cdae_split_bf16": [P, P, P, L, P],
    "cdae_upsample2_split": [P, P, P, P, P, I, I, I, I, P],
    "cdae_wdgrad_planes": [P, P, P, I, I, P],
    "cdae_s2dgrad_wfold": [P, P, P, I, I, P],
    "cdae_wt_planes_bf16": [P, L, P, P, I, I, P],
    "cdae_linear_dgrad_stream": [P, L, P, P, L, P, L, I, I, I, P],
    "cdae_conv3x3_s2_dgrad_ps": [P, P, P, P, P, L, I, I, I, I, I, P, SZ, P],
    "cdae_wprep_all": [P, P, I, I, L, P, P, P, P, P],
    "cdae_wprep_all_k": [P, P, I, I, L, P, P, P, P, P, P, P, P, P, P],
    "cdae_conv3x3_dgrad_ps": [P, P, P, P, P, L, I, I, I, I, I, P, SZ, P],
    "cdae_conv3x3_stem_supported": [I, I, I],
    "cdae_conv3x3_stem": [P, L, L, L, L, P, P, P, L, I, I, I, I, I, P],
    "cdae_conv3x3_stem_gn": [P, L, L, L, L, P, P, P, L, P, I, I, I, I, I, P],
    "cdae_conv3x3_wgrad_fewout": [P, P, L, P, P, I, I, I, I, I, I, P, SZ, P],
    "cdae_conv3x3_wgrad_win_supported": [I, I, I, I, I],
    "cdae_conv3x3_wgrad_win": [P, P, P, P, P, P, I, I, I, I, I, I, P, SZ, P],
    "cdae_conv3x3_wgrad_win_group": [P, I, P, SZ, P],
    "cdae_gn_apply_split_train": [P, P, P, P, P, I, I, I, I, I, I, P, P, P, P, P, I, I, P],
    "cdae_gn_apply_split": [P, P, P, I, I, I, I, I, I, P, P, P, P, P, I, I, P],
    "cdae_gn_stats2": [P, I, P, I, I, I, I, I, I, F, P, P, P, P],
    "cdae_linear_fwd_cat": [P, L, I, P, L, P, L, P, P, P, L, I, I, I, P, SZ, P],
    "cdae_linear_fwd_cat_gn": [P, L, I, P, L, P, L, P, P, P, L, P, I, P, P, I, I, I, I, P, SZ, P],
    "cdae_head_conv_supported": [I, I, I],
    "cdae_head_conv_fwd": [P, L, P, I, P, P, P, I, I, I, I, I, P],
    "cdae_skip_gn_ok": [I, I, I, I, I],
    "cdae_linear_fwd_stream": [P, L, I, P, L, P, P, L, P, P, P, L, P, L, P, P, I, I, I, P],
    "cdae_linear_fwd_stream_gn_part": [P, L, I, P, L, P, P, L, P, P, P, L, P, L, P, P, P, I, I, I, P],
    "cdae_linear_fwd_stream_gn": [P, L, P, P, L, P, P, P, L, P, I, I, I, I, I, P],
    "cdae_skip_gn_fwd": [P, L, I, P, L, P, P, L, P, P, P, L, P, I, P, P, I, I, I, I, I, P],
    "cdae_gn_apply_split2g": [P, I, P, I, I, P, P, I, I, I, I, P, P, P, P, P, I, I, P],
    "cdae_planes_gm_to_pc": [P, P, P, P, L, I, P],
    "cdae_conv3x3_fwd_psg": [P, P, L, L, L, I, P, P, P, P, P, P, P, P, L, I, P, P, P, I, I, I, I, I, I, I, P, SZ, P],
    "cdae_gn_apply_split2": [P, I, P, I, I, P, P, I, I, I, I, I, P, P, P, P, P, I, I, P],
    "cdae_linear_dgrad": [P, L, P, L, P, L, I, I, I, I, P, SZ, P],
    "cdae_linear_wgrad": [P, L, P, L, P, L, P, I, I, I, I, P, SZ, P],
    "cdae_colsum": [P, L, P, L, I, I, P],
    "cdae_qkv_attention_fwd": [P, P, P, I, I, I, I, P],
    "cdae_qkv_attention_fused_supported": [I, I],
    "cdae_qkv_attention_fwd_fused": [P, P, I, I, I, I, P],
    "cdae_qkv_attention_fwd_fused_p": [P, P, P, I, I, I, I, P],
    "cdae_qkv_attention_bwd_q_fused": [P, P, P, P, P, I, I, I, I, P],
    "cdae_qkv_attention_bwd": [P, P, P, P, P, I, I, I, I, P],
    "cdae_gn_workspace_floats": [I, I],
    "cdae_gn_stats": [P, I, I, I, I, I, F, P, P, P, P],
    "cdae_gn_apply": [P, P, I, I, I, I, I, I, P, P, P, P, P, I, I, P],
    "cdae_gn_bwd": [P, P, P, I, I, I, I, I, I, I, P, P, P, P, P, I, I, P, P, I, P, I, I, P, P],
    "cdae_gn_apply_split_train2": [P, I, P, I, I, P, P, P, P, I, I, I, I, I, P, P, P, P, P, I, I, P],
    "cdae_gn_bwd_cat": [P, I, P, I, I, P, I, P, I, P, I, I, I, I, I, P, P, P, P, P, I, I, P, P, I, P, I, I, P, P],
    "cdae_gn_bwd_ex": [P, P, P, I, I, I, I, I, I, I, P, P, P, P, P, I, I, P, P, I, P, I, I, P, I, P, P, P, P],
    "cdae_bn_workspace_floats": [I],
    "cdae_bn_lrelu_fwd": [P, P, L, I, P, P, P, P, I, F, F, F, P, P, P, P, P, P],
    "cdae_bn_lrelu_bwd": [P, P, P, L, I, P, P, P, P, P, F, P, P, I, P, P],
    "cdae_softmax_rows": [P, L, I, P],
    "cdae_softmax_rows_bwd": [P, P, L, I, P],
    "cdae_act_fwd": [P, P, L, I, P],
    "cdae_act_bwd": [P, P, P, L, I, P],
    "cdae_silu_fwd": [P, P, L, P],
    "cdae_silu_bwd": [P, P, P, L, P],
    "cdae_timestep_embed_fwd": [P, P, P, I, I, P],
    "cdae_model_timesteps": [P, P, F, I, P, P, I, P],
    "cdae_embedding_add": [P, P, P, I, I, P],
    "cdae_embedding_bwd": [P, P, P, I, I, P],
    "cdae_axpby": [F, P, F, P, P, L, P],
    "cdae_mul_scale": [P, P, F, P, L, P],
    "cdae_mul_rows": [P, P, I, I, P],
    "cdae_copy2d": [P, P, L, I, L, L, I, P],
    "cdae_nchw_to_nhwc": [P, P, I, I, I, P],
    "cdae_nhwc_to_nchw": [P, P, I, I, I, P],
    "cdae_sumpool2": [P, P, I, I, I, I, P],
    "cdae_upsample2": [P, P, I, I, I, I, F, P],
    "cdae_pool2": [P, P, I, I, I, I, F, P],
    "cdae_gather_u8": [P, P, P, I, L, F, F, P],
    "cdae_q_sample": [P, P, P, P, I, P, I, L, P],
    "cdae_ddim_update": [P, P, P, P, I, F, P, I, P, P, I, L, P],
    "cdae_ddpm_update": [P, P, P, P, I, P, I, P, P, I, L, P],
    "cdae_softplus_fwd": [P, P, L, F, P],
    "cdae_softplus_bwd": [P, P, P, L, P],
    "cdae_reparam": [P, P, F, P, P, L, P],
    "cdae_causal_mask": [P, P, P, I, I, I, I, P],
    "cdae_adamw_ema": [P, P, P, P, P, L, D, D, D, D, D, I, D, D, P],
    "cdae_adamw_ema_multi": [P, P, P, P, P, P, I, L, D, D, D, D, D, I, D, P],
    "cdae_sqsum": [P, L, P, P],
    "cdae_p_mean_variance": [P, P, P, P, I, I, I, I, P, P, P, P, P, P, I, L, P],
    "cdae_vb_terms": [P, P, P, P, P, I, I, I, I, P, P, I, L, P],
    "cdae_vb_terms_bwd": [P, P, P, P, P, I, I, I, I, I, P, P, I, L, P],
    "cdae_mse_rows": [P, P, P, I, L, P],
    "cdae_mse_rows_bwd": [P, P, P, P, I, L, P],
    "cdae_rep_loss": [P, P, P, P, P, I, I, I, P],
    "cdae_rep_loss_bwd": [P, P, P, P, P, P, P, P, I, I, I, P],
    "cdae_conv3x3_fwd16": [P, L, L, L, P, P, P, P, P, L, P, I, I, I, I, I, P, SZ, P],
    "cdae_conv3x3_dgrad16": [P, P, P, P, L, I, I, I, I, I, P, SZ, P],
    "cdae_gemm16_ps": [P, L, P, L, P, P, P, L, P, I, I, I, I, I, P, SZ, P],
    "cdae_linear_fwd_io": [P, L, P, L, P, P, P, P, L, I, I, I, I, P, SZ, P],
    "cdae_linear_dgrad_io": [P, L, P, L, P, L, I, I, I, I, P, SZ, P],
    "cdae_linear_wgrad_group": [P, I, P, SZ, P],
    "cdae_linear_wgrad_group_io": [P, I, I, P, SZ, P],
    "cdae_linear_wgrad_io": [P, L, P, L, P, L, P, I, I, I, I, I, P, SZ, P],
    "cdae_gn_stats16": [P, I, P, I, I, I, I, I, I, F, P, P, P, P, P, I, P, P, P],
    "cdae_gn_apply16": [P, I, P, I, I, P, I, I, I, I, I, P, P, P, P, P, I, I, P],
    "cdae_gn_bwd16": [P, I, P, I, I, P, I, P, I, P, I, I, I, I, I, P, P, P, P, P, I, I, P, P, I, P, I, I, P, I, P, P],
    "cdae_conv3x3_s2_fwd16": [P, L, L, L, P, P, P, L, I, I, I, I, I, P, SZ, P],
    "cdae_im2col3x3_16": [P, P, I, I, I, I, P],
    "cdae_attn16_supported": [I, I],
    "cdae_rows16_supported": [I, I, I, I, I],
    "cdae_attn16_fwd": [P, P, P, I, I, I, I, P],
    "cdae_attn16_bwd": [P, P, P, P, P, P, I, I, I, I, P],
    "cdae_gn_parts16": [P, L, P, L, I, P],
    "cdae_cast_f32_bf16": [P, P, L, P],
    "cdae_cast_bf16_f32": [P, P, L, P],
    "cdae_upsample2_16": [P, P, I, I, I, I, P],
    "cdae_sumpool2_16": [P, P, I, I, I, I, P],
    "cdae_wprep_all_m16": [P, P, I, I, L, P, P, P, P, P, P, P, P, P, P],
    "cdae_tune_set": [I, I],
    "cdae_tune_get": [I],
    "cdae_prof_enable": [I],
    "cdae_prof_read": [P, P, P, P],
    "cdae_calib_mfma": [P, SZ, I, P, P, P],
    "cdae_calib_copy": [P, P, SZ, I, P, P],
}
_RESTYPES = {"cdae_last_error": ctypes.c_char_p, "cdae_gn_workspace_floats": SZ, "cdae_bn_workspace_floats": SZ, "cdae_workspace_bytes": SZ}

TAB_ROWS = 12
PROF_FAMILIES = ("igemm", "groupnorm", "softmax", "elementwise", "optimizer", "convwin", "convwin_dgrad", "convwin_up")


class CdaeError(RuntimeError):
    pass


class LwItem(ctypes.Structure):
    """cdae_lw_item (include/cdae.h): one linear / 1x1 weight gradient of a group launch"""
    _fields_ = [("x", P), ("dy", P), ("dw", P), ("dbias", P), ("ldx", L), ("lddy", L), ("lddw", L), ("M", I), ("N", I), ("K", I), ("accumulate", I)]


class WgItem(ctypes.Structure):
    """cdae_wg_item (include/cdae.h): one weight gradient of a group launch"""
    _fields_ = [("a_hi", P), ("a_lo", P), ("dy_hi", P), ("dy_lo", P), ("dw", P), ("dbias", P),
                ("N", I), ("H", I), ("W", I), ("Cin", I), ("Cout", I), ("accumulate", I)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or causaldiffae_amd/csrc/build.sh).  causaldiffae_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise CdaeError(lib.cdae_last_error().decode())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_dev = torch.cuda.current_device


def stream():
    """hipStream_t of torch's current stream.  Called once per kernel launch (~300 times per training step): the raw getter
    costs a fraction of a microsecond, `torch.cuda.current_stream().cuda_stream` builds a Python Stream object each time (~9 us)."""
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: the product path is GPU only."""
    if t is None:
        return None
    if not t.is_cuda:
        raise CdaeError("causaldiffae_amd ops need tensors on an MI355X device (got a CPU tensor); "
                        "there is no CPU fallback — the CPU oracle lives in oracle/ for tests only")
    return t.data_ptr()


def ptr2(t):
    """(pointer of t[0], pointer of t[1]) of a [2, ...] plane pair without building the two view tensors (`t[i]` is an aten::select:
    ~1500 of them per training step)"""
    if t is None:
        return None, None
    if isinstance(t, (tuple, list)):          # a pair of separate plane tensors
        return ptr(t[0]), ptr(t[1])
    if not t.is_cuda:
        raise CdaeError("causaldiffae_amd ops need tensors on an MI355X device (got a CPU tensor)")
    p = t.data_ptr()
    return p, p + t.stride(0) * t.element_size()


_ws = {}
_ws_retired = []
_PROBED = set()
_WS_LANE = [0]


class ws_lane:
    """with ws_lane(k): ... — the scratch buffers (`workspace`) of lane k.  Work issued to DIFFERENT streams that may overlap on the device
    must not share split-K slabs / norm partials: each concurrent stream of launches runs under its own lane (lane 0 = the default)."""

    def __init__(self, k):
        self.k, self.prev = int(k), 0

    def __enter__(self):
        self.prev, _WS_LANE[0] = _WS_LANE[0], self.k
        return self

    def __exit__(self, *exc):
        _WS_LANE[0] = self.prev
        return False


def workspace(device, name, nbytes):
    """Persistent per-device (and per-lane, see ws_lane) scratch buffers (split-K slabs, norm partials); grown on demand, never shrunk."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _WS_LANE[0], name)
    if key[0] not in _PROBED and not torch.cuda.is_current_stream_capturing():
        # the window conv kernel's padding taps rely on out-of-range LDS reads returning zeros: the library checks that once per device
        # (cdae_convwin_lds_probe allocates and synchronises) — here, at the first workspace of the device, not inside a launch
        _PROBED.add(key[0])
        with torch.cuda.device(key[0]):
            check(lib.cdae_convwin_lds_probe(stream()))
    buf = _ws.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        if buf is not None:
            _ws_retired.append(buf)       # a captured HIP graph may still hold this pointer: outgrown buffers are kept, never recycled
        buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _ws[key] = buf
    return buf


WS_SPLITK, WS_GROUPNORM, WS_GN_PARTS, WS_BATCHNORM = 0, 1, 2, 3


def workspace_bytes(op, *dims):
    """cdae_workspace_bytes(op, dims): the library states what scratch an op wants (include/cdae.h)"""
    arr = (ctypes.c_long * len(dims))(*dims) if dims else None
    return int(lib.cdae_workspace_bytes(op, arr, len(dims)))


SPLITK_BYTES = workspace_bytes(WS_SPLITK)       # the covering size, asked of the library once at import


def splitk_ws(device):
    """ONE split-K workspace per device, sized by the library for every layer of the reference's UNets (no dims = the covering size)"""
    return workspace(device, "splitk", SPLITK_BYTES)


PRECISIONS = {"fp32": 0, "f16x3": 1, "mixed16": 2}


def set_precision(name):
    """'fp32' (v_mfma_f32_32x32x2_f32, exact fp32 chain), 'f16x3' (split-precision f16 MFMA, ~2^-22, default) or
    'mixed16' (single f16 / bf16 plane: the reduced-precision torso, outside the 1e-4 parity bar)."""
    check(lib.cdae_set_default_precision(PRECISIONS[name]))


def get_precision():
    return {v: k for k, v in PRECISIONS.items()}[lib.cdae_get_default_precision()]


class CdaeRangeError(CdaeError):
    pass


_RANGE_PENDING = [None]        # a flag found raised by range_clear: kept for the next range_check (whoever issued that work has not checked yet)


def range_check(what="a contraction"):
    """Synchronise, read and clear the library's range flag; raise if any contraction produced a non-finite value since the last
    check (an operand beyond the f16 range of the split-precision planes, or a genuinely non-finite input) — including a flag that
    a range_clear() in between found raised and set aside."""
    torch.cuda.synchronize()
    bad = ctypes.c_int(0)
    check(lib.cdae_range_status(ctypes.byref(bad)))
    pending, _RANGE_PENDING[0] = _RANGE_PENDING[0], None
    if bad.value or pending:
        whose = what if bad.value else f"{what} (raised by earlier work: {pending})"
        raise CdaeRangeError(f"{whose}: non-finite result — an operand left the range of the f16 split-precision planes (|x| < 65520) "
                             f"or was not finite; rerun with causaldiffae_amd.set_precision('fp32') (IEEE fp32 products, fp32 range)")


def range_clear(owner="work issued before a sampling loop"):
    """Start a fresh reporting interval: the next range_check then speaks for the work issued from here on.  A flag that is ALREADY
    raised (e.g. by training steps since the trainer's last range_guard, when a sampling callback runs in between) is not dropped:
    it is set aside and re-raised by the next range_check — after this loop's own check has passed — so the trainer's guard in
    save() still sees it."""
    torch.cuda.synchronize()
    bad = ctypes.c_int(0)
    check(lib.cdae_range_status(ctypes.byref(bad)))
    if bad.value and _RANGE_PENDING[0] is None:
        _RANGE_PENDING[0] = owner


def range_take_pending():
    """-> the description of a set-aside flag (or None), clearing it: for a caller that wants to check ITS work only and report
    earlier work separately (the sampling loop)."""
    pending, _RANGE_PENDING[0] = _RANGE_PENDING[0], None
    return pending


class precision_scope:
    """with precision_scope("mixed16"): ... — the library's arithmetic mode for the duration of a block, restored on exit (a model
    converted with convert_to_fp16() applies it around its own forward and TrainLoop around forward + backward, so a second model
    or a sampler in the same process keeps the parity mode).  None = leave the mode alone."""

    def __init__(self, name):
        self.name, self.prev = name, None

    def __enter__(self):
        if self.name is not None:
            self.prev = get_precision()
            if self.prev != self.name:
                set_precision(self.name)
        return self

    def __exit__(self, *exc):
        if self.name is not None and self.prev != self.name:
            set_precision(self.prev)
        return False


TUNE_KEYS = {"convwin_min_tiles": 0, "convwin_splitk": 1, "convwin_nj3": 2, "head_mfma": 3, "rows16_min_m": 4, "rows16_ring": 5, "convwin_pair16": 6, "gn_bwd_fold2": 7, "group_big_tiles": 8, "convwin_nj2": 9, "wgwin_dist": 10, "wgwin_swz": 11, "wg16_slots": 12, "wgwin_fixed": 13, "wgwin_co2": 14}


class tune_scope:
    """with tune_scope(convwin_min_tiles=1): ... — move dispatch thresholds of the library (include/cdae.h, cdae_tune_set) for the
    duration of a block; the parity tests use it to run small golden cases on the kernels the benchmark shapes dispatch."""

    def __init__(self, **kv):
        self.kv, self.prev = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.prev[k] = lib.cdae_tune_get(TUNE_KEYS[k])
            check(lib.cdae_tune_set(TUNE_KEYS[k], int(v)))
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            check(lib.cdae_tune_set(TUNE_KEYS[k], v))
        return False


def calibrate(device, copy_bytes=1 << 30):
    """{mfma_sustained_tflops, sclk_under_load_ghz, hbm_copy_tbps} of THIS box, measured through the library (cdae_calib_*): what bench.py
    normalises its roofline fractions with.  Synchronises; ~0.1 s."""
    scratch = torch.empty((4 << 20) + 8192, dtype=torch.uint8, device=device)
    tf, ghz, tb = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
    check(lib.cdae_calib_mfma(ptr(scratch), scratch.numel(), 20000, ctypes.byref(tf), ctypes.byref(ghz), stream()))
    src = torch.empty(copy_bytes // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    check(lib.cdae_calib_copy(ptr(src), ptr(dst), copy_bytes, 5, ctypes.byref(tb), stream()))
    return {"mfma_sustained_tflops": tf.value, "sclk_under_load_ghz": ghz.value, "hbm_copy_tbps": tb.value}


def prof_enable(on):
    check(lib.cdae_prof_enable(1 if on else 0))


def prof_read():
    n = len(PROF_FAMILIES)
    ms = (ctypes.c_double * n)()
    work = (ctypes.c_double * n)()
    nbytes = (ctypes.c_double * n)()
    cnt = (ctypes.c_longlong * n)()
    check(lib.cdae_prof_read(ms, work, nbytes, cnt))
    return {PROF_FAMILIES[i]: dict(ms=ms[i], work=work[i], bytes=nbytes[i], launches=cnt[i]) for i in range(n)}
