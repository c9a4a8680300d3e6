"""Dev tool: the C64 batch-32 training step on the 16-bit torso (use_fp16 / mixed16) — not a bench leg (the C64 leg is quoted in the parity
mode); what the torso does for the reference's largest configuration."""
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
for fp16 in (True, False):
    r = bench.train_bench(dev, 1, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 20, 3, 32, regions=2, use_fp16=fp16)
    print({k: r[k] for k in ("value", "ms_per_step", "precision_mode", "last_loss")})
