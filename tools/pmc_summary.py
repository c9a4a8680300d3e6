#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes (separate runs, CSV output) of one command into the per-launch summary that bench.py's
`roofline.traffic` reads.  Kernels are selected by a regex on the kernel name.

    python tools/pmc_summary.py --fetch DIR1 --write DIR2 --sq DIR3 --trace DIR4 --kernels 'ps_kernel|pswin_kernel|igemm_kernel' \
        --label f16x3 --out profiles/r01_igemm_pmc_summary_f16x3.json

FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md §HBM); WRITE_SIZE is exact; both are in KiB.
Effective clock = GRBM_GUI_ACTIVE / 8 XCDs / the SAME pass's dispatch duration (the counter CSV's own Start / End timestamps): a
counter pass serialises and stretches short launches, so dividing its cycles by the kernel-trace pass's duration gave 2.7-3.2 "GHz" on a
2.4 GHz part for kernels under 50 us (round 5).  Without same-pass timestamps the clock is reported only for kernels >= 50 us.  SQ_* busy / wait figures are fractions of SQ_WAVE_CYCLES (quad-cycle
units cancel) except SQ_VALU_MFMA_BUSY_CYCLES, which is in cycles per SIMD-summed... so it is reported against SQ_BUSY_CU_CYCLES."""
import argparse
import collections
import csv
import glob
import json
import re


def load_counters(d, pat):
    """-> {counter: [values per matching dispatch in dispatch order]}"""
    out = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                out[r["Counter_Name"]][int(r["Dispatch_Id"])] = out[r["Counter_Name"]].get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return {k: [v[i] for i in sorted(v)] for k, v in out.items()}


def load_pass_durations(d, pat):
    """-> [ns per matching dispatch] from the counter pass's OWN timestamps (rocprofv3 writes Start_Timestamp / End_Timestamp into
    counter_collection.csv), or [] when the columns are missing"""
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]) and r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dt = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                if dt > 0:
                    out[int(r["Dispatch_Id"])] = dt
    return [out[i] for i in sorted(out)]


def load_durations(d, pat):
    out = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                out.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out


def mean(v):
    return sum(v) / len(v) if v else None


def main():
    ap = argparse.ArgumentParser()
    for k in ("fetch", "write", "sq", "trace", "kernels", "label", "out"):
        ap.add_argument("--" + k, required=True)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    pat = re.compile(a.kernels)
    fetch = load_counters(a.fetch, pat)
    write = load_counters(a.write, pat)
    sq = load_counters(a.sq, pat)
    dur = load_durations(a.trace, pat)
    n = len(dur)
    avg_ns = mean(dur)
    fetch_kb = mean(fetch.get("FETCH_SIZE", []))
    write_kb = mean(write.get("WRITE_SIZE", []))
    hit, miss = sum(write.get("TCC_HIT_sum", [])), sum(write.get("TCC_MISS_sum", []))
    wave = sum(sq.get("SQ_WAVE_CYCLES", [])) or None
    gui = mean(sq.get("GRBM_GUI_ACTIVE", []))
    sq_ns = mean(load_pass_durations(a.sq, pat))
    clock, clock_src = None, None
    if gui and sq_ns:
        clock, clock_src = gui / 8 / sq_ns, "GRBM_GUI_ACTIVE / 8 / dispatch duration of the same counter pass"
    elif gui and avg_ns and avg_ns >= 50e3:
        clock, clock_src = gui / 8 / avg_ns, "GRBM_GUI_ACTIVE / 8 / kernel-trace duration (separate pass; kernel >= 50 us)"
    if clock is not None and clock > 2.45:
        clock, clock_src = None, f"dropped: {clock:.2f} GHz is above the part's 2.4 GHz maximum ({clock_src})"
    busy_cu = sum(sq.get("SQ_BUSY_CU_CYCLES", [])) or None
    res = {
        "kernel": f"{a.kernels} ({a.label})",
        "launches": n,
        "fetch_size_kb_raw_avg": fetch_kb,
        "fetch_bytes_corrected_avg": None if fetch_kb is None else fetch_kb * 1024 * 2,
        "write_bytes_avg": None if write_kb is None else write_kb * 1024,
        "hbm_traffic_bytes_per_launch": None if fetch_kb is None or write_kb is None else fetch_kb * 2048 + write_kb * 1024,
        "tcc_hit_rate": hit / (hit + miss) if hit + miss else None,
        "avg_launch_us": None if avg_ns is None else avg_ns / 1e3,
        "avg_launch_us_in_counter_pass": None if sq_ns is None else sq_ns / 1e3,
        "effective_clock_ghz": clock,
        "effective_clock_source": clock_src,
        "mfma_busy_frac": None if not busy_cu else sum(sq.get("SQ_VALU_MFMA_BUSY_CYCLES", [])) / busy_cu / 4,
        "lds_bank_conflict_cycles": sum(sq.get("SQ_LDS_BANK_CONFLICT", [])),
        "wait_any_frac": None if not wave else sum(sq.get("SQ_WAIT_ANY", [])) / wave,
        "wait_inst_frac": None if not wave else sum(sq.get("SQ_WAIT_INST_ANY", [])) / wave,
        "note": a.note or "separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE+TCC | SQ+GRBM) and one --kernel-trace pass of the same command",
    }
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
