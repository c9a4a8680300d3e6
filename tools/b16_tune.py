"""Dev tool: the batch-16 DDIM step of bench.py under dispatch thresholds (cdae_tune_set keys from the command line: KEY=VALUE ...)."""
import os, sys, json, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from causaldiffae_amd._lib import lib, TUNE_KEYS
KEYS = list(sys.argv[1:])
for kv in KEYS:
    k, v = kv.split("=")
    assert lib.cdae_tune_set(TUNE_KEYS[k], int(v)) == 0
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--regions", "3", "--no-train", "--no-fp32", "--no-extra", "--no-cpu-baseline", "--batch", "16"]
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
for l in buf.getvalue().splitlines():
    if l.startswith("{"):
        d = json.loads(l)
        print(" ".join(KEYS) or "(defaults)", "| batch 16: %.3f ms per step" % d["ms_per_step"])
