"""Fold rocprofv3 passes of tools/prof_shapes.py into the per-shape table (markdown on stdout)."""
import collections, csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_shapes import SHAPES, B
pat = re.compile(r"convwin_kernel|pswin_kernel|ps_kernel")
def per_dispatch(d, want):
    out = collections.OrderedDict()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]) and r["Counter_Name"] in want:
                out.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
                out[int(r["Dispatch_Id"])][r["Counter_Name"]] = out[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]
def durations(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    return [(n, us) for _, n, us in sorted(rows)]
root = sys.argv[1]
dur = durations(root + "/trace")
sq = per_dispatch(root + "/sq", {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"})
fe = per_dispatch(root + "/fetch", {"FETCH_SIZE"})
wr = per_dispatch(root + "/write", {"WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"})
assert len(dur) == 4 * len(SHAPES), (len(dur), len(SHAPES))
print("| conv3x3 shape (batch 128) | per step | kernel | us | TFLOP/s | frac of 833 | MFMA busy | waves parked | HBM read MB | HBM write MB | algorithmic MB | traffic ratio | L2 hit |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
tot_us = tot_fl = 0.0
for i, (ci, co, r, cnt) in enumerate(SHAPES):
    sl = slice(4 * i + 1, 4 * i + 4)                        # skip the warm-up launch of each shape
    us = sum(u for _, u in dur[sl]) / 3
    name = dur[4 * i + 1][0]
    kern = "convwin" if "convwin" in name else ("pswin" if "pswin" in name else "ps")
    fl = 2.0 * B * r * r * co * ci * 9
    s = sq[sl] if len(sq) == len(dur) else []
    busy = sum(x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for x in s) / max(1.0, 4 * sum(x.get("SQ_BUSY_CU_CYCLES", 0) for x in s)) if s else float("nan")
    park = sum(x.get("SQ_WAIT_ANY", 0) for x in s) / max(1.0, sum(x.get("SQ_WAVE_CYCLES", 0) for x in s)) if s else float("nan")
    rd = sum(x.get("FETCH_SIZE", 0) for x in fe[sl]) / 3 * 2048 / 1e6 if len(fe) == len(dur) else float("nan")      # KiB, x2 on gfx950 (MI355X_MICROARCH.md)
    w_ = wr[sl] if len(wr) == len(dur) else []
    wb = sum(x.get("WRITE_SIZE", 0) for x in w_) / 3 * 1024 / 1e6 if w_ else float("nan")
    hit = sum(x.get("TCC_HIT_sum", 0) for x in w_) / max(1.0, sum(x.get("TCC_HIT_sum", 0) + x.get("TCC_MISS_sum", 0) for x in w_)) if w_ else float("nan")
    alg = (4.0 * B * r * r * ci + 4.0 * 9 * ci * co + 4.0 * B * r * r * co) / 1e6
    print(f"| {ci}->{co} @{r}x{r} | {cnt} | {kern} | {us:.1f} | {fl / us / 1e6:.1f} | {fl / us / 1e6 / 833.33:.3f} | {busy:.3f} | {park:.3f} | {rd:.0f} | {wb:.0f} | {alg:.0f} | {(rd + wb) / alg:.2f} | {hit:.3f} |")
    tot_us += us * cnt; tot_fl += fl * cnt
print(f"\nweighted by launches per DDIM step: {tot_us / 1e3:.2f} ms per step in these convs, {tot_fl / tot_us / 1e6:.1f} TFLOP/s = {tot_fl / tot_us / 1e6 / 833.33:.3f} of the f16x3 roof")
