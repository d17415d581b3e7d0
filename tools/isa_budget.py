#!/usr/bin/env python3
"""Where the instructions of a kernel's hot loop come from.

usage: isa_budget.py [--src mh_encode.hip] [--kernel enc_region_kernelILb0] [--symbols-per-trip 32768]
Compiles the file with line tables (-gline-tables-only; code generation is unchanged), takes the LONGEST loop of the kernel
(for enc_region_kernel<false> the steady-state loop of two 16 KiB rounds per trip; the loops inside it — the flush's — run one
trip per round and are counted once), attributes every instruction to the source line of its `.loc` and sums per CATEGORY = the innermost
function or lambda of the inlining chain the line belongs to (the table below names them).  Output: instructions per trip and
per lane, and vector-ALU instructions per symbol (a lane handles symbols-per-trip / 1024 symbols per trip)."""
import argparse, os, re, subprocess, sys, collections

ap = argparse.ArgumentParser()
ap.add_argument("--src", default="mh_encode.hip")
ap.add_argument("--kernel", default="enc_region_kernelILb0")
ap.add_argument("--symbols-per-trip", type=int, default=2 * 16384)
ap.add_argument("--pick", default="largest", choices=["largest", "innermost"],
                help="largest: the longest loop of the kernel (inner loops counted once: the flush's loop runs one trip per round); innermost: the longest loop without a loop inside")
a = ap.parse_args()
here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, "..", "markov-huffman-coding_amd", "csrc")
out = "/tmp/isa_budget.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-gline-tables-only", "-S",
                       os.path.join(csrc, a.src), "-o", out], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and a.kernel in l and l.rstrip().endswith(":") or (l.startswith("_Z") and a.kernel in l and ":" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
lab = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
back = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < i:
        back.append((lab[m.group(1)], i))
inner = [(t, i) for (t, i) in back if not any(t < t2 and i2 < i for (t2, i2) in back)]
if a.pick == "innermost":
    cand = inner
else:   # the longest loop that is not merely a wrapper around another long loop (the kernel's outer control flow)
    cand = [(t, i) for (t, i) in back if not any((t, i) != (t2, i2) and t <= t2 and i2 <= i and i2 - t2 >= 1000 for (t2, i2) in back)]
t, i = max(cand, key=lambda p: p[1] - p[0])

# categories: (file, first line, last line, name), innermost (shortest) range wins
def ranges(fname, marks, after=None):
    src = open(os.path.join(csrc, fname)).read().split("\n")
    first = next((k for k, l in enumerate(src) if after and re.search(after, l)), 0)       # lambdas: the ones of THIS kernel
    out = []
    for name, pat_start, pat_end in marks:
        s = next((k for k, l in enumerate(src) if k >= first and re.search(pat_start, l)), None)
        if s is None:
            continue
        e = next((k for k in range(s + 1, len(src)) if re.search(pat_end, src[k])), len(src) - 1)
        out.append((fname, s + 1, e + 1, name))
    return out
cats = []
cats += ranges("mh_dev.hpp", [("slot arithmetic (slots16)", r"void slots16\(", r"^}"), ("scan of the lane's bits (wave_inclusive_sum)", r"uint32_t wave_inclusive_sum\(", r"^}"),
                              ("input load (load_raw / head_byte)", r"LaneIn load_raw\(", r"^}"), ("input load (load_raw / head_byte)", r"uint32_t head_byte\(", r"^}")])
cats += ranges("mh_encode.hip", [("deposit (shifts + ds_or of a group)", r"void deposit\(uint32_t \*stage", r"^}")])
cats += ranges("mh_encode.hip", [("input load (fetch / fetch_full)", r"auto fetch = \[&\]", r"^    };"), ("input load (fetch / fetch_full)", r"auto fetch_full = \[&\]", r"^    };"),
                                 ("lookup (lookup16: 16 LDS reads)", r"auto lookup16 = \[&\]", r"^    };"),
                                 ("flush (LDS image -> HBM, byte swap)", r"auto flush = \[&\]", r"^    };"),
                                 ("pack (four <= 48-bit groups + lengths)", r"auto pack = \[&\]", r"^    };"),
                                 ("round: exchange, index entries, bookkeeping", r"auto round = \[&\]", r"^    };")], after=r"void enc_region_kernel\(")
def cat_of(f, ln):
    best = None
    for (cf, s, e, name) in cats:
        if cf == f and s <= ln <= e and (best is None or e - s < best[0]):
            best = (e - s, name)
    if best:
        return best[1]
    if f == "amd_hip_atomic.h":
        return "deposit (shifts + ds_or of a group)"
    if ln == 0:
        return "compiler-generated (no source line)"
    return "other (%s)" % f

cur = ("?", 0)
# the .loc in force at the loop's top
for l in body[:t]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
acc = collections.defaultdict(lambda: collections.Counter())
kinds = [("valu", r"v_"), ("salu", r"s_(?!waitcnt|nop|barrier|cbranch|branch)"), ("lds", r"ds_"), ("vmem", r"(global|buffer|flat|scratch)_"), ("wait/branch/barrier", r"s_(waitcnt|nop|barrier|cbranch|branch)")]
for l in body[t:i + 1]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    ins = l.strip()
    if not ins or ins.startswith((".", ";")) or ins.endswith(":"):
        continue
    for k, pat in kinds:
        if re.match(pat, ins):
            acc[cat_of(*cur)][k] += 1
            break
sym_per_lane = a.symbols_per_trip / 1024.0
print("%s: loop at asm lines %d-%d (%d lines); one trip = %d symbols per workgroup = %g per lane" % (a.kernel, t + start, i + start, i - t, a.symbols_per_trip, sym_per_lane))
print("%-52s %6s %6s %5s %5s %6s   %s" % ("category", "valu", "salu", "lds", "vmem", "other", "valu / symbol"))
tot = collections.Counter()
for name, c in sorted(acc.items(), key=lambda kv: -kv[1]["valu"]):
    tot.update(c)
    print("%-52s %6d %6d %5d %5d %6d   %.2f" % (name, c["valu"], c["salu"], c["lds"], c["vmem"], c["wait/branch/barrier"], c["valu"] / sym_per_lane))
print("%-52s %6d %6d %5d %5d %6d   %.2f" % ("TOTAL", tot["valu"], tot["salu"], tot["lds"], tot["vmem"], tot["wait/branch/barrier"], tot["valu"] / sym_per_lane))
