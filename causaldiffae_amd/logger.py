"""Minimal key-value logger with the call surface the reference's scripts use (improved_diffusion/logger.py is
observability, SURVEY §2: out of scope; this keeps log / logkv / logkv_mean / dumpkvs / configure / get_dir)."""
import json
import os
import sys
import time
from collections import defaultdict

_STATE = dict(dir=None, kvs={}, sums=defaultdict(float), counts=defaultdict(int), t0=time.time())


def configure(dir=None, format_strs=None, comm=None, log_suffix=""):
    if dir is None:
        dir = os.environ.get("OPENAI_LOGDIR") or os.path.join("/tmp", time.strftime("cdae-%Y%m%d-%H%M%S"))
    os.makedirs(os.path.expanduser(dir), exist_ok=True)
    _STATE["dir"] = os.path.expanduser(dir)


def get_dir():
    return _STATE["dir"]


def log(*args):
    print(*args, file=sys.stdout, flush=True)


info = log


def warn(*args):
    print("WARN:", *args, file=sys.stderr, flush=True)


def logkv(key, val):
    _STATE["kvs"][key] = val


def logkv_mean(key, val):
    _STATE["sums"][key] += float(val)
    _STATE["counts"][key] += 1


def getkvs():
    out = dict(_STATE["kvs"])
    for k, s in _STATE["sums"].items():
        out[k] = s / max(1, _STATE["counts"][k])
    return out


def dumpkvs():
    kv = getkvs()
    if kv:
        log(" | ".join(f"{k}={v:.6g}" if isinstance(v, float) else f"{k}={v}" for k, v in sorted(kv.items())))
        if _STATE["dir"]:
            with open(os.path.join(_STATE["dir"], "progress.jsonl"), "a") as f:
                f.write(json.dumps({k: (float(v) if hasattr(v, "__float__") else v) for k, v in kv.items()}) + "\n")
    _STATE["kvs"].clear()
    _STATE["sums"].clear()
    _STATE["counts"].clear()
    return kv
