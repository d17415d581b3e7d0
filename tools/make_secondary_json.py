#!/usr/bin/env python3
"""profiles/secondary.json from a tools/pmc_passes.sh run (tools/pmc_summary.py JSON):
   python tools/make_secondary_json.py gpurun_out/<dir>/summary.json <size> <path to cite>
The secondary bounds of every kernel — what the integer kernels of this path are limited by long before HBM:
   valu_frac  share of the SIMDs' time in which a vector-ALU instruction issues:
              SQ_ACTIVE_INST_VALU x 4 / (kernel cycles x CUs x 4 SIMDs)   (the SQ counts quad-cycles; MI355X_MICROARCH.md)
   lds_frac   share of the kernel's cycles in which a CU's LDS array is busy: SQ_LDS_IDX_ACTIVE / (kernel cycles x CUs);
              lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
   ta_frac    TA_BUSY_avr / kernel cycles (texture addresser: every vector-memory instruction goes through it)
   wait_frac  SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked at s_waitcnt or a barrier)
   valu_per_symbol = SQ_INSTS_VALU x 64 / size
kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs).  Peak used for valu_frac: 256 CUs x 4 SIMDs, one
wave-instruction per 4 cycles each = 256 x 4 x 16 lanes x 2.4 GHz = 3.9e13 lane-operations per second."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as entry  # noqa: E402
import importlib  # noqa: E402
entry.load_package()
provenance = importlib.import_module("mhc_amd.provenance")

src, size, cite = sys.argv[1], int(sys.argv[2]), sys.argv[3]
CUS = 256
c = json.load(open(src))
out = {"_counters": cite, "_method": __doc__.split("\n", 3)[3].strip()}
for k, v in sorted(c.items()):
    name = re.sub(r"^mhk::", "", k).split("<")[0]
    cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0:
        continue
    e = {"kernel_cycles": int(cyc), "instantiation": k}
    if "SQ_ACTIVE_INST_VALU" in v:
        e["valu_frac"] = round(v["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * CUS * 4), 3)
    if "SQ_LDS_IDX_ACTIVE" in v:
        e["lds_frac"] = round(v["SQ_LDS_IDX_ACTIVE"] / (cyc * CUS), 3)
        if v["SQ_LDS_IDX_ACTIVE"] > 0:
            e["lds_conflict_frac"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 3)
    if "TA_BUSY_avr" in v:
        e["ta_frac"] = round(v["TA_BUSY_avr"] / cyc, 3)
    if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
        e["wait_frac"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3)
    if "SQ_INSTS_VALU" in v:
        e["valu_per_symbol"] = round(v["SQ_INSTS_VALU"] * 64 / size, 2)
    # several instantiations of one kernel (escape / no-escape launch): keep the longest-running one
    key = "%s:%d" % (name, size)
    if key not in out or out[key]["kernel_cycles"] < e["kernel_cycles"]:
        out[key] = e
json.dump(provenance.stamp(out), sys.stdout, indent=1)    # + the hash of every kernel's sources as they are NOW: run this right after the counter run
print()
