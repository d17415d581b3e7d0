#!/usr/bin/env python3
"""Time of the order-2 histogram on the GPU box, with the workspace (the device chooses tag cache or partition per slab)
and without it (tag cache only):  python3 tools/hist2_rate.py [--size BYTES] [--kinds zipf,uniform,text]
Checks that both give the same counts and that they add up to the input's size.  (Order 2: extension, parity unpinned.)"""
import argparse, ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1 << 30)
ap.add_argument("--kinds", default="uniform,zipf,text")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mhc = entry.load_package()
lib = mhc.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws_bytes = int(lib.mh_dev_histogram_o2_workspace(a.size))
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
counts = torch.zeros(1 << 24, dtype=torch.int64, device=dev)
plain = torch.zeros(1 << 24, dtype=torch.int64, device=dev)
for kind in a.kinds.split(","):
    data = bench.generate(kind, a.size, {"zipf": 2, "uniform": 3, "text": 1}[kind], 0, dev)
    for name, use_ws, out in (("workspace", True, counts), ("tag cache only", False, plain)):
        best = 1e9
        for _ in range(a.reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc = lib.mh_dev_histogram_o2_ws(data.data_ptr(), a.size, 0x2020, out.data_ptr(), ws.data_ptr() if use_ws else None,
                                            ws_bytes if use_ws else 0, st)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        assert rc == 0
        path = lib.mh_dev_index_path(ws.data_ptr(), st) if use_ws else 1
        print("%-8s %-15s %8.2f ms  %6.1f GB/s  choices %d  sum ok %s" % (kind, name, best * 1e3, a.size / best / 1e9, path,
              int(out.sum().item()) == a.size), flush=True)
    print("%-8s counts equal: %s  live keys %d" % (kind, bool(torch.equal(counts, plain)), int((counts != 0).sum().item())), flush=True)
    del data
