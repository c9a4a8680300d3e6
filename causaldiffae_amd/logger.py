"""Key-value logger with the reference's call surface and on-disk formats (improved_diffusion/logger.py):
`configure / get_dir / log / info / warn / error / debug / logkv / logkv_mean / logkvs / getkvs / dumpkvs / reset`, and the
three file formats its scripts leave behind in the log dir —

  log<suffix>.txt       dashed two-column table per dump, values `%-8.3g`, keys/values cut to 30 chars (logger.py:50-77)
  progress<suffix>.csv  one row per dump; the header grows when new keys appear (new keys sorted, appended; old rows
                        padded with empty cells) (logger.py:119-143)
  progress<suffix>.json one JSON object per line (logger.py:102-107)

Rank comes from torchrun's RANK (the reference looks at PMI_RANK / OMPI_COMM_WORLD_RANK): rank 0 defaults to
"stdout,log,csv" (OPENAI_LOG_FORMAT), other ranks to "log" with a `-rankNNN` suffix (OPENAI_LOG_FORMAT_MPI).
Cross-rank averaging of the dumped means uses torch.distributed when `comm` is truthy (the reference's mpi_weighted_mean)."""
import datetime
import json
import os
import sys
import tempfile

DEBUG, INFO, WARN, ERROR, DISABLED = 10, 20, 30, 40, 50


def _cut(s, maxlen=30):
    return s[:maxlen - 3] + "..." if len(s) > maxlen else s


class _TableSink:
    """stdout / log.txt: free-text lines and the dashed key-value table."""
    takes_lines = True

    def __init__(self, target):
        self.own = isinstance(target, str)
        self.f = open(target, "wt") if self.own else target

    def write_kvs(self, kvs):
        cells = {}
        for k, v in sorted(kvs.items()):
            cells[_cut(k)] = _cut("%-8.3g" % v if hasattr(v, "__float__") else str(v))
        if not cells:
            print("WARNING: tried to write empty key-value dict")
            return
        kw, vw = max(map(len, cells)), max(map(len, cells.values()))
        bar = "-" * (kw + vw + 7)
        rows = ["| %s | %s |" % (k.ljust(kw), v.ljust(vw)) for k, v in sorted(cells.items(), key=lambda kv: kv[0].lower())]
        self.f.write("\n".join([bar] + rows + [bar]) + "\n")
        self.f.flush()

    def write_line(self, parts):
        self.f.write(" ".join(parts) + "\n")
        self.f.flush()

    def close(self):
        if self.own:
            self.f.close()


class _JsonSink:
    takes_lines = False

    def __init__(self, path):
        self.f = open(path, "wt")

    def write_kvs(self, kvs):
        self.f.write(json.dumps({k: (float(v) if hasattr(v, "dtype") else v) for k, v in sorted(kvs.items())}) + "\n")
        self.f.flush()

    def close(self):
        self.f.close()


class _CsvSink:
    """progress.csv.  Rows are kept in memory (one short string per dump) so a header change is a plain rewrite."""
    takes_lines = False

    def __init__(self, path):
        self.path, self.keys, self.rows = path, [], []
        open(path, "wt").close()

    def write_kvs(self, kvs):
        fresh = sorted(set(kvs) - set(self.keys))
        row = None
        if fresh:
            self.keys.extend(fresh)
            self.rows = [r + "," * len(fresh) for r in self.rows]
        row = ",".join("" if kvs.get(k) is None else str(kvs[k]) for k in self.keys)
        self.rows.append(row)
        if fresh:
            with open(self.path, "wt") as f:
                f.write(",".join(self.keys) + "\n" + "".join(r + "\n" for r in self.rows))
        else:
            with open(self.path, "at") as f:
                f.write(row + "\n")

    def close(self):
        pass


def make_output_format(format, ev_dir, log_suffix=""):
    os.makedirs(ev_dir, exist_ok=True)
    if format == "stdout":
        return _TableSink(sys.stdout)
    if format == "log":
        return _TableSink(os.path.join(ev_dir, "log%s.txt" % log_suffix))
    if format == "json":
        return _JsonSink(os.path.join(ev_dir, "progress%s.json" % log_suffix))
    if format == "csv":
        return _CsvSink(os.path.join(ev_dir, "progress%s.csv" % log_suffix))
    raise ValueError("Unknown format specified: %s (tensorboard is not provided)" % (format,))


class Logger:
    def __init__(self, dir, sinks, comm=None):
        self.dir, self.sinks, self.comm, self.level = dir, sinks, comm, INFO
        self.val, self.cnt = {}, {}

    def logkv(self, key, val):
        self.val[key] = val

    def logkv_mean(self, key, val):
        n = self.cnt.get(key, 0)
        old = self.val.get(key, 0.0) if n else 0.0
        self.val[key] = old * n / (n + 1) + val / (n + 1)          # running mean, same update order as logger.py:350-353
        self.cnt[key] = n + 1

    def dumpkvs(self):
        d = self.val if not self.comm else _weighted_mean_over_ranks(self.val, self.cnt)
        out = dict(d)
        for s in self.sinks:
            s.write_kvs(d)
        self.val, self.cnt = {}, {}
        return out

    def log(self, *args, level=INFO):
        if self.level <= level:
            for s in self.sinks:
                if s.takes_lines:
                    s.write_line([str(a) for a in args])

    def close(self):
        for s in self.sinks:
            s.close()


def _weighted_mean_over_ranks(val, cnt):
    """key -> sum(val*count)/sum(count) over ranks, delivered on rank 0 ({'dummy': 1} elsewhere, like the reference)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(val)
    local = {}
    for k, v in val.items():
        try:
            local[k] = (float(v), cnt.get(k, 1))
        except (TypeError, ValueError):
            pass
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, local)
    if dist.get_rank() != 0:
        return {"dummy": 1}
    num, den = {}, {}
    for part in gathered:
        for k, (v, c) in part.items():
            num[k] = num.get(k, 0.0) + v * c
            den[k] = den.get(k, 0.0) + c
    return {k: num[k] / den[k] for k in num}


_DEFAULT = Logger(None, [_TableSink(sys.stdout)])
_CURRENT = _DEFAULT


def get_current():
    return _CURRENT


def _rank():
    for var in ("RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK"):
        if var in os.environ:
            return int(os.environ[var])
    return 0


def configure(dir=None, format_strs=None, comm=None, log_suffix=""):
    global _CURRENT
    if dir is None:
        dir = os.getenv("OPENAI_LOGDIR")
    if dir is None:
        dir = os.path.join(tempfile.gettempdir(), datetime.datetime.now().strftime("openai-%Y-%m-%d-%H-%M-%S-%f"))
    dir = os.path.expanduser(dir)
    os.makedirs(dir, exist_ok=True)
    rank = _rank()
    if rank > 0:
        log_suffix = log_suffix + "-rank%03i" % rank
    if format_strs is None:
        env, default = ("OPENAI_LOG_FORMAT", "stdout,log,csv") if rank == 0 else ("OPENAI_LOG_FORMAT_MPI", "log")
        format_strs = os.getenv(env, default).split(",")
    sinks = [make_output_format(f, dir, log_suffix) for f in format_strs if f]
    if _CURRENT is not _DEFAULT:
        _CURRENT.close()
    _CURRENT = Logger(dir, sinks, comm)
    if sinks:
        log("Logging to %s" % dir)


def reset():
    global _CURRENT
    if _CURRENT is not _DEFAULT:
        _CURRENT.close()
        _CURRENT = _DEFAULT
        log("Reset logger")


def get_dir():
    return _CURRENT.dir


def log(*args, level=INFO):
    _CURRENT.log(*args, level=level)


def debug(*args):
    log(*args, level=DEBUG)


def info(*args):
    log(*args, level=INFO)


def warn(*args):
    log(*args, level=WARN)


def error(*args):
    log(*args, level=ERROR)


def set_level(level):
    _CURRENT.level = level


def logkv(key, val):
    _CURRENT.logkv(key, val)


def logkv_mean(key, val):
    _CURRENT.logkv_mean(key, val)


def logkvs(d):
    for k, v in d.items():
        logkv(k, v)


def getkvs():
    return _CURRENT.val


def dumpkvs():
    return _CURRENT.dumpkvs()
