#!/usr/bin/env python3
"""Streams without an index in two passes (mh_dev_decode_stream_states + mh_dev_decode_stream_emit) on the GPU box:
   python3 tools/stream_rate.py [--size BYTES] [--kind zipf|text|uniform] [--reps N]
Encodes the synthetic stream, throws both indices away, and times the two calls (HIP events on the launch stream; the states
call waits for the device between its sub-passes).  Also the workload for rocprofv3 runs of the two kernels
(tools/pmc_cmd.sh <out> stream_rate.py ...)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4 << 30)
ap.add_argument("--kind", default="zipf")
ap.add_argument("--reps", type=int, default=3)
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
wsb = int(lib.mh_dev_build_index_workspace(nbits))
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
nsym = torch.zeros(1, dtype=torch.int64, device=dev)
ev = lambda: torch.cuda.Event(enable_timing=True)
for it in range(a.reps):
    codec.decoded.zero_()
    e = [ev() for _ in range(3)]
    e[0].record()
    rc1 = lib.mh_dev_decode_stream_states(model.handle, codec.payload.data_ptr(), nbits, 0x20, nsym.data_ptr(), ws.data_ptr(), wsb, codec.stream())
    e[1].record()
    path = lib.mh_dev_index_path(ws.data_ptr(), codec.stream())
    changed = ws[64:64 + 12 * 4].view(torch.int32).tolist()          # segments listed for repair, per pass (first entry: the sample's)
    rc2 = -1
    if path == 6:
        rc2 = lib.mh_dev_decode_stream_emit(model.handle, codec.payload.data_ptr(), nbits, 0x20, codec.decoded.data_ptr(), a.size, ws.data_ptr(), wsb, codec.stream())
    e[2].record()
    torch.cuda.synchronize()
    ok = path == 6 and int(nsym.item()) == a.size and lib.mh_dev_status(ws.data_ptr(), codec.stream()) == 0 and bool(torch.equal(codec.decoded[:a.size], data))
    print("rc %d %d  path %d  states %.3f ms  emit %.3f ms  total %.3f ms  (%.1f GB/s)  bit_exact %s  listed per pass %s" %
          (rc1, rc2, path, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[0].elapsed_time(e[2]),
           a.size / (e[0].elapsed_time(e[2]) * 1e-3) / 1e9, ok, changed), flush=True)
