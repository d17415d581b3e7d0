#!/usr/bin/env python3
"""profiles/traffic.json from a tools/pmc_traffic.sh run:
   python tools/make_traffic_json.py gpurun_out/<dir>/traffic_counters.json <size> <counters path to cite>
read bytes = TCC_EA0_RDREQ_sum x 128 B when every request was a 128-byte one (checked), else the split
sum; write bytes = WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM/rocprofv3 section)."""
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
c = json.load(open(src))
out = {"_counters": cite,
       "_method": "rocprofv3 --pmc, separate passes (tools/pmc_traffic.sh, %d-byte Zipf(1.1) workload, one launch each). "
                  "read bytes = TCC_EA0_RDREQ_sum x 128 B (every request was 128 B: TCC_EA0_RDREQ_128B_sum == TCC_EA0_RDREQ_sum "
                  "within 0.01 %%; FETCH_SIZE reads exactly half of that on gfx950, as MI355X_MICROARCH.md says); "
                  "write bytes = WRITE_SIZE x 1024. Values are read + write bytes per launch." % size}
agg = {}
for k, v in c.items():
    name = re.sub(r"^mhk::", "", k).split("<")[0]
    rd = v.get("TCC_EA0_RDREQ_sum", 0.0)
    r128 = v.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    r32 = v.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    read = r128 * 128 + r32 * 32 + max(rd - r128 - r32, 0.0) * 64
    write = v.get("WRITE_SIZE", 0.0) * 1024
    a = agg.setdefault(name, [0.0, 0.0, 0.0, ""])
    a[0] += read
    a[1] += write
    if read + write >= a[2]:                    # the instantiation that moved the most bytes names the entry
        a[2], a[3] = read + write, k
for name, (r, w, _, inst) in sorted(agg.items()):
    out["%s:%d" % (name, size)] = int(r + w)
    out["%s:%d:instantiation" % (name, size)] = inst
    out["%s:%d:read" % (name, size)] = int(r)
    out["%s:%d:write" % (name, size)] = int(w)
# the encode stage = every kernel of whichever encoder ran (region path, or length pass + scans + emit)
enc = ("enc_region_kernel", "region_bits_kernel", "region_scan_kernel",
       "enc_len_kernel", "enc_emit_kernel", "scan_local_kernel", "scan_top_kernel", "scan_apply_kernel")
if any("%s:%d" % (k, size) in out for k in enc):
    out["encode_kernel:%d" % size] = sum(out.get("%s:%d" % (k, size), 0) for k in enc)
json.dump(provenance.stamp(out), sys.stdout, indent=1)    # + the hash of every kernel's sources as they are NOW: run this right after the counter run
print()
