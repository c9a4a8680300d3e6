"""Dev tool: fold a rocprofv3 --pmc counter_collection CSV by kernel name: per kernel the launch count and the sum of each counter, sorted by the first
counter given — e.g. which kernels of a step have LDS bank conflicts at all.

    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d DIR -o c -- python3 bench.py ...
    python3 tools/pmc_by_kernel.py DIR
"""
import collections, csv, glob, re, sys

d = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
names = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0][:64]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
        if r["Counter_Name"] not in names:
            names.append(r["Counter_Name"])
print("kernel".ljust(66), "launches", " ".join(n.rjust(22) for n in names))
for k in sorted(tot, key=lambda k: -tot[k].get(names[0], 0.0))[:40]:
    print(k.ljust(66), f"{len(cnt[k]):8d}", " ".join(f"{tot[k].get(n, 0.0):22.4g}" for n in names))
