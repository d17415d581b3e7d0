#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 counter passes: one histogram, model build, encode, decode of
`--size` bytes of Zipf(1.1) (default 1 GiB), round trip checked.  Usage on the GPU box:
   rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 tools/prof_run.py"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1 << 30)
ap.add_argument("--kind", default="zipf")
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--order", type=int, default=1, choices=[1, 2])
a = ap.parse_args()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mhc = entry.load_package()
if bench.CHUNK == 0:
    bench.CHUNK = 1024 if a.size >= (2 << 30) else 256
data = bench.generate(a.kind, a.size, {"zipf": 2, "uniform": 3, "text": 1}[a.kind], 0, dev)
codec = bench.Codec(mhc, a.size, dev, order=a.order)
prev0 = 0x20 if a.order == 1 else 0x2020
for _ in range(a.reps):
    codec.histogram(data, prev0)
    model = codec.build_model()
    codec.encode(model, data, prev0)
    codec.decode(model)
torch.cuda.synchronize()
assert torch.equal(codec.decoded, data)
print("ok", int(codec.nbits[0].item()))
