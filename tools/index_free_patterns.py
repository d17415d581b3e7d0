#!/usr/bin/env python3
"""Index builder on awkward streams (GPU box): python3 tools/index_free_patterns.py [--size BYTES]
For each pattern: encode on the device, rebuild the index without the sidecar (timed), compare with the encoder's."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1 << 30)
ap.add_argument("--only", default="")
ap.add_argument("--no-index-build", action="store_true", help="only the regular round trip (encode, decode with the encoder's index)")
a = ap.parse_args()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mhc = entry.load_package()
bench.CHUNK = 1024
n = a.size
g = torch.Generator(device=dev); g.manual_seed(5)
def rnd(k): return torch.randint(0, k, (n,), dtype=torch.uint8, device=dev, generator=g)
patterns = {
    "zeros": lambda: torch.zeros(n, dtype=torch.uint8, device=dev),
    "ab": lambda: (torch.arange(n, device=dev) % 2 + 97).to(torch.uint8),
    "period3": lambda: (torch.arange(n, device=dev) % 3 + 65).to(torch.uint8),
    "two_symbols": lambda: rnd(2) + 48,
    "alphabet8": lambda: rnd(8) + 48,
    "alphabet64": lambda: rnd(64) + 32,
    "alphabet128": lambda: rnd(128),
    "uniform256": lambda: rnd(256),
    "runs": lambda: (torch.arange(n, device=dev) // 4096 % 256).to(torch.uint8),
    "text": lambda: bench.generate("text", n, 1, 0, dev),
}
codec = bench.Codec(mhc, n, dev)
lib = codec.lib
for name, make in patterns.items():
    if a.only and name not in a.only.split(','):
        continue
    data = make()
    codec.histogram(data, 0x20)
    model = codec.build_model()
    codec.encode(model, data, 0x20)
    torch.cuda.synchronize()
    nbits = int(codec.nbits[0].item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); codec.decode(model); e1.record(); torch.cuda.synchronize()
    rt = bool(torch.equal(codec.decoded, data)) and lib.mh_dev_status(codec.dec_ws.data_ptr(), codec.stream()) == 0
    print("%-12s round trip with the encoder's index: %s, decode %.2f ms" % (name, rt, e0.elapsed_time(e1)), flush=True)
    if a.no_index_build:
        continue
    ws_bytes = int(lib.mh_dev_build_index_workspace(max(nbits, 1)))
    ws = torch.empty(max(ws_bytes, 64), dtype=torch.uint8, device=dev)
    idx2 = torch.zeros_like(codec.index)
    nsym = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = lib.mh_dev_build_index(model.handle, codec.payload.data_ptr(), nbits, 0x20, idx2.data_ptr(), idx2.numel(), bench.CHUNK,
                                nsym.data_ptr(), ws.data_ptr(), ws_bytes, codec.stream())
    st = lib.mh_dev_status(ws.data_ptr(), codec.stream())
    dt = time.perf_counter() - t0
    print("%-12s nbits/8/n %.4f  rc=%d status=%d  %9.2f ms  symbols ok %s  index equal %s" %
          (name, nbits / 8 / n, rc, st, dt * 1e3, int(nsym.item()) == n, bool(torch.equal(idx2, codec.index))), flush=True)
    del data, ws, idx2
