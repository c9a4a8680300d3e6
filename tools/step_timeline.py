"""Dev tool: fold a rocprofv3 --kernel-trace CSV of eager DDIM (or training) steps into (a) the ordered kernel list of the LAST step with each
kernel's duration and the idle gap in front of it, (b) per-kernel-name totals of that step.  A step ends at `--end` (default: ddim_update_kernel).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 2 --warmup 1 --regions 1 --no-graph ...
    python3 tools/step_timeline.py DIR [--end adamw_ema] [--list]
"""
import collections, csv, glob, re, sys

d = sys.argv[1]
end = sys.argv[sys.argv.index("--end") + 1] if "--end" in sys.argv else "ddim_update_kernel"
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [i for i, r in enumerate(rows) if end in r[2]]
assert len(ends) >= 2, f"need two '{end}' launches, found {len(ends)}"
step = rows[ends[-2] + 1:ends[-1] + 1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]


wall = step[-1][1] - rows[ends[-2]][1]
busy = sum(e - s for s, e, _ in step)
print(f"last step: {len(step)} launches, wall {wall / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms")
agg = collections.defaultdict(lambda: [0, 0])
prev_end = rows[ends[-2]][1]
gaps = 0
for s, e, n in step:
    a = agg[short(n)]
    a[0] += 1; a[1] += e - s
    if "--list" in sys.argv:
        print(f"{(s - prev_end) / 1e3:8.1f} gap {(e - s) / 1e3:9.1f} us  {short(n)}")
    prev_end = max(prev_end, e)
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / 1e6:8.3f} ms {c:4d}x {t / c / 1e3:9.1f} us  {n}")
