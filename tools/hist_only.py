#!/usr/bin/env python3
"""Only the order-1 histogram (region mode) of `--size` bytes of Zipf(1.1): a workload for counter passes over
hist_o1_kernel alone, also with a diagnostic library whose counts are wrong (MH_LIB=..., nothing is checked)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4 << 30)
a = ap.parse_args()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mhc = entry.load_package()
bench.CHUNK = 1024
data = bench.generate("zipf", a.size, 2, 0, dev)
codec = bench.Codec(mhc, a.size, dev)
for _ in range(2):
    codec.histogram(data, 0x20)
torch.cuda.synchronize()
print("ok", int(codec.counts.sum().item()))
