#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel (device assembly from hipcc -S).

usage: isa_loop_count.py <mangled-name-substring> [extra hipcc flags...]
Prints, for every backward branch of the kernel, the loop's span and its VALU / SALU / LDS / VMEM counts,
so a change to a hot loop can be priced before it goes to the GPU box."""
import re, subprocess, sys, os
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "markov-huffman-coding_amd", "csrc", os.environ.get("MH_ISA_SRC", "mh_encode.hip"))
name = sys.argv[1]
out = "/tmp/isa_loop_count.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out] + sys.argv[2:],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l)      # the kernel's label line
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
lab = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: lab[m.group(1)] = i
print(lines[start].split(":")[0], len(body), "lines")
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < i:
        t = lab[m.group(1)]
        b = body[t:i + 1]
        c = lambda pat: sum(1 for x in b if re.match(r"\s+" + pat, x))
        if i - t > 200:
            print(f"loop {t + start}-{i + start}: valu {c('v_')} salu {c('s_')} lds {c('ds_')} vmem {c('(global|buffer|flat)_')} branch {c('s_c?branch')} waitcnt {c('s_waitcnt')}")
