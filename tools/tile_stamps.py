#!/usr/bin/env python3
"""Diagnostic build of the tile decoder (MH_TILE_STAMP=1): where a wave's cycles go.  Prints the shares of
positions (index loads), staging, the 64-symbol decode loop, and output + checks.  Never quote its run time."""
import os, sys
os.environ["MH_TILE_STAMP"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, __graft_entry__ as entry
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
mhc = entry.load_package()
bench.CHUNK = 1024
data = bench.generate("zipf", size, 2, 0, dev)
codec = bench.Codec(mhc, size, dev)
codec.histogram(data, 0x20); model = codec.build_model(); codec.encode(model, data, 0x20); codec.decode(model)
torch.cuda.synchronize()
ws = codec.dec_ws[:64].cpu().numpy().view(np.uint64)
seg = ws[1:5].astype(float)
print("cycles summed over waves: positions %.3g staging %.3g decode %.3g output %.3g" % tuple(seg))
print("shares: positions %.1f%% staging %.1f%% decode %.1f%% output+checks %.1f%%" % tuple(100 * seg / seg.sum()))
npieces = size // 8192
print("per piece (2 tiles): positions %.0f staging %.0f decode %.0f (%.0f per symbol step) output %.0f cycles" %
      (seg[0] / npieces, seg[1] / npieces, seg[2] / npieces, seg[2] / npieces / 64, seg[3] / npieces))
print("round trip", bool(torch.equal(codec.decoded, data)))
