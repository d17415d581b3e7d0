#!/usr/bin/env python3
"""Where does the decode kernel's time go?  Valid-data probes on the GPU box (16 GiB Zipf by default):
  full      the regular decode
  hot_in    every chunk's index entry points into the first 4096 chunks of the payload, so the input
            comes out of L2 while tables, instruction stream and output stay what they are (the chunks that
            wrap end at the wrong bit: the status word says corrupt, the timing does not care)
Output: one line per probe, ms per call (median of 3 after one warm-up)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=16 << 30)
ap.add_argument("--kind", default="zipf")
ap.add_argument("--window", type=int, default=4096)
a = ap.parse_args()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mhc = entry.load_package()
bench.CHUNK = 1024
data = bench.generate(a.kind, a.size, 2, 0, dev)
codec = bench.Codec(mhc, a.size, dev)
codec.histogram(data, 0x20)
model = codec.build_model()
codec.encode(model, data, 0x20)
torch.cuda.synchronize()


def timed(label):
    ts = []
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); codec.decode(model); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("%-8s %.3f ms" % (label, sorted(ts[1:])[1]), flush=True)


timed("full")
print("round trip", bool(torch.equal(codec.decoded, data)), flush=True)
idx = codec.index
n = idx.numel()
sel = torch.arange(n, device=dev) % a.window
idx.copy_(idx[sel])
del sel
timed("hot_in")
