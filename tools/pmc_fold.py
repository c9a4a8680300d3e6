"""Dev tool: fold rocprofv3 counter_collection CSVs under the given directories into per-kernel averages (one line per kernel)."""
import collections, csv, glob, re, sys
pat = re.compile(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                per[(r["Kernel_Name"][:70], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, c, _), v in per.items():
            acc[k][c].append(v)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                acc[r["Kernel_Name"][:70]]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    print(k)
    m = {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in cs.items()}         # skip the first (cold) dispatch
    for c in sorted(m):
        print(f"    {c:40s} {m[c]:16.1f}")
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        print(f"    -> wait_any {m.get('SQ_WAIT_ANY', 0) / w:.3f}  wait_inst {m.get('SQ_WAIT_INST_ANY', 0) / w:.3f}  valu_active {m.get('SQ_ACTIVE_INST_VALU', 0) / w:.3f}"
              f"  mfma_busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1.0, m.get('SQ_BUSY_CU_CYCLES', 1)) / 4:.3f}")
