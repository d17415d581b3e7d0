#!/usr/bin/env python3
"""Time of the index builder (streams that come without a sidecar index, the reference's own format) on the
GPU box:  python3 tools/index_free_rate.py [--size BYTES] [--kind zipf|text|uniform]
Encodes the synthetic stream, throws the index away, rebuilds it with mh_dev_build_index (timed), compares it
with the encoder's, and decodes with it."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4 << 30)
ap.add_argument("--kind", default="zipf")
a = ap.parse_args()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mhc = entry.load_package()
bench.CHUNK = 1024
data = bench.generate(a.kind, a.size, {"zipf": 2, "uniform": 3, "text": 1}[a.kind], 0, dev)
codec = bench.Codec(mhc, a.size, dev)
codec.histogram(data, 0x20)
model = codec.build_model()
codec.encode(model, data, 0x20)
torch.cuda.synchronize()
nbits = int(codec.nbits[0].item())
lib = codec.lib
ws_bytes = int(lib.mh_dev_build_index_workspace(nbits))
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
idx2 = torch.zeros_like(codec.index)
nsym = torch.zeros(1, dtype=torch.int64, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = lib.mh_dev_build_index(model.handle, codec.payload.data_ptr(), nbits, 0x20, idx2.data_ptr(), idx2.numel(), bench.CHUNK,
                                nsym.data_ptr(), ws.data_ptr(), ws_bytes, codec.stream())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    changed = ws[64:64 + 16 * 4].view(torch.int32).tolist()          # the pass counters: segments (re)decoded per pass
    print("build_index rc=%d  %.2f ms  (%.1f GB/s of payload)  symbols %d  index equal %s  path %d  changed per pass %s" %
          (rc, dt * 1e3, nbits / 8 / dt / 1e9, int(nsym.item()), bool(torch.equal(idx2, codec.index)),
           lib.mh_dev_index_path(ws.data_ptr(), codec.stream()), changed), flush=True)
