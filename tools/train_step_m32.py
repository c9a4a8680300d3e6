"""Dev tool: a few training steps of BASELINE config [1] (M32, batch 256; argv[2] = 1: use_fp16 / mixed16) for rocprofv3 kernel traces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
if os.environ.get("OFF"):               # OFF=name,name: fused paths of causaldiffae_amd.ops.PATH_TOGGLES switched off (same-box A/B)
    from causaldiffae_amd import ops as _ops
    for _n in os.environ["OFF"].split(","):
        setattr(_ops, _ops.PATH_TOGGLES[_n], False)
if os.environ.get("PAIR16"):            # A/B of the window conv's channel-halves form on bf16 rows (CDAE_TUNE_CONVWIN_PAIR16)
    from causaldiffae_amd._lib import lib as _l2
    _l2.cdae_tune_set(__import__("causaldiffae_amd")._lib.TUNE_KEYS["convwin_pair16"], int(os.environ["PAIR16"]))
if os.environ.get("ROWS16_MIN_M"):          # A/B of the streaming kernels' row threshold (include/cdae.h, CDAE_TUNE_ROWS16_MIN_M)
    from causaldiffae_amd._lib import lib as _l
    _l.cdae_tune_set(__import__("causaldiffae_amd")._lib.TUNE_KEYS["rows16_min_m"], int(os.environ["ROWS16_MIN_M"]))
if os.environ.get("TUNE"):                  # TUNE=key=value,key=value: dispatch thresholds of the library (cdae_tune_set) for a same-box A/B
    from causaldiffae_amd import _lib as _l3
    for kv in os.environ["TUNE"].split(","):
        k, v = kv.split("=")
        assert _l3.lib.cdae_tune_set(_l3.TUNE_KEYS[k], int(v)) == 0
fp16 = len(sys.argv) > 2 and sys.argv[2] == "1"
r = bench.train_bench(dev, 1, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 3, 2, 256, use_fp16=fp16, workload="M32", image_size=32, in_channels=1, n_vars=2, class_cond=True)
print({k: r[k] for k in ("value", "ms_per_step", "precision_mode")})
