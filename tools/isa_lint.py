"""Dev: compile one translation unit of csrc/ to gfx950 assembly and list, per kernel, what the COMPILER put into the loops that
contain MFMAs: `s_waitcnt` it inserted itself (not from inline asm), scratch (spill) traffic, and the instruction mix.  A compiler
vmcnt wait inside a software-pipelined DMA loop is a full drain of the LDS-DMAs the loop has just issued (a spill reload is a
vector-memory load: its first use inside the loop gets `s_waitcnt vmcnt(0)`): this is the check that it is not there.

    python tools/isa_lint.py convwin [extra hipcc flags]      (runs here: hipcc cross-compiles without a GPU)
"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lint(unit, extra=()):
    """[{kernel, lines, mfmas, scratch_ops, vgpr_spills, loops: [{start, end, instructions, mfmas, lds_ops, compiler_waits, vmcnt_waits, scratch}]}]"""
    src = os.path.join(ROOT, "causaldiffae_amd", "csrc", unit + ".hip")
    out = os.path.join(tempfile.gettempdir(), f"isa_{unit}.s")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]
    if unit == "elementwise":
        flags.append("-ffp-contract=off")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, *extra, "-o", out, src], stderr=subprocess.DEVNULL)
    text = open(out).read()
    lines = text.split("\n")
    spills = dict(re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text))
    res = []
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s", l)]
    for s in starts:
        e = next(k for k in range(s, len(lines)) if lines[k].startswith(".Lfunc_end"))      # (not the first s_endpgm: an early exit may be laid out in front of the loops)
        body = lines[s:e + 1]
        sym = lines[s].split(":")[0]
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        in_asm, tagged = False, []
        for l in body:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            tagged.append((t, in_asm))
        labels = {m.group(1): k for k, (t, _) in enumerate(tagged) for m in [re.match(r"^(\.LBB\d+_\d+):", t)] if m}
        loops = []
        for k, (t, _) in enumerate(tagged):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", t)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                loops.append((labels[m.group(1)], k))
        mfma_loops = [(a, b) for a, b in loops if any("v_mfma" in tagged[k][0] for k in range(a, b))]
        inner = [(a, b) for a, b in mfma_loops if not any((c > a or d < b) and c >= a and d <= b for c, d in mfma_loops if (c, d) != (a, b))]
        rec = dict(kernel=name, lines=len(body), mfmas=sum("v_mfma" in t for t, _ in tagged), scratch_ops=sum(t.startswith("scratch_") for t, _ in tagged),
                   vgpr_spills=int(spills.get(sym, -1)), loops=[])
        for a, b in sorted(set(inner)):
            seg = tagged[a:b + 1]
            mix = collections.Counter(t.split()[0] for t, _ in seg if t and not t.startswith((";", ".")))
            cw = [(a + k, t) for k, (t, ia) in enumerate(seg) if t.startswith("s_waitcnt") and not ia]
            rec["loops"].append(dict(start=a, end=b, instructions=sum(mix.values()), mfmas=sum(v for k_, v in mix.items() if k_.startswith("v_mfma")),
                                     lds_ops=sum(v for k_, v in mix.items() if k_.startswith("ds_")), s_nop=mix["s_nop"], barriers=mix["s_barrier"],
                                     compiler_waits=len(cw), vmcnt_waits=[x for x in cw if "vmcnt" in x[1]],
                                     scratch=[(a + k, t[:60]) for k, (t, _) in enumerate(seg) if t.startswith("scratch_")]))
        res.append(rec)
    return res


def inflight_copies(unit, kernel_substr, extra=()):
    """{kernel: [copy instructions]}: `v_mov` / `v_accvgpr_write` instructions whose SOURCE is a destination register of a
    `global_load_dwordx4` issued from inline assembly.  Such loads are invisible to the compiler's wait-count pass (the kernel waits
    for them with its own counted `s_waitcnt`), so a register copy the allocator places between the load and that wait — which it does
    when an asm "+v" operand ends up tied to a different register than the load's output — copies data that has not landed
    (skipgn_kernel, round 4: wrong planes and NaN sums in one build variant, nothing in the source had changed)."""
    src = os.path.join(ROOT, "causaldiffae_amd", "csrc", unit + ".hip")
    out = os.path.join(tempfile.gettempdir(), f"isa_copies_{unit}.s")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, *extra, "-o", out, src], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    res = {}
    for s in [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s*", l)]:
        sym = lines[s].split(":")[0]
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        if kernel_substr not in name:
            continue
        e = next(k for k in range(s, len(lines)) if "s_endpgm" in lines[k])
        dst, in_asm = set(), False
        for l in lines[s:e]:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off", t)
            if m and in_asm:
                dst.update(range(int(m.group(1)), int(m.group(2)) + 1))
        bad = []
        for l in lines[s:e]:
            t = l.strip()
            m = re.match(r"(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_write_b32) (\S+), (\S+)", t)
            if not m:
                continue
            regs = set()
            mm = re.match(r"v\[(\d+):(\d+)\]", m.group(3))
            if mm:
                regs = set(range(int(mm.group(1)), int(mm.group(2)) + 1))
            mm = re.match(r"v(\d+)$", m.group(3))
            if mm:
                regs = {int(mm.group(1))}
            if regs & dst:
                bad.append(t)
        res[name] = dict(asm_load_registers=len(dst), copies=bad)
    return res


if __name__ == "__main__":
    for r in lint(sys.argv[1], sys.argv[2:]):
        print(f"== {r['kernel'][:100]}: {r['lines']} lines, {r['mfmas']} MFMAs, {r['scratch_ops']} scratch ops, {r['vgpr_spills']} VGPR spills")
        for lp in r["loops"]:
            print(f"   loop @{lp['start']}-{lp['end']}: {lp['instructions']} instructions, {lp['mfmas']} MFMAs, {lp['lds_ops']} LDS ops, {lp['s_nop']} s_nop, {lp['barriers']} barriers")
            print(f"      compiler-inserted waits: {lp['compiler_waits']} in all; on vmcnt (DMA / load drains): {lp['vmcnt_waits'] if lp['vmcnt_waits'] else 'none'}")
            if lp["scratch"]:
                print(f"      SPILL traffic inside the loop: {lp['scratch']}")
