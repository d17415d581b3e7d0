#!/usr/bin/env python3
"""Soak of the one-pass order-2 encoder (run on the GPU box): many launches back to back over a spread of sizes, every one
checked for status OK and for the payload length the histogram predicts (histogram . code lengths); a sample of them
decoded.  A launch that gave up waiting (MH_ERR_TIMEOUT) or ended elsewhere would show here.
   python3 tools/soak_chain.py [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 400
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mhc = entry.load_package()
sizes = [(1 << 20) + 5, (16 << 20) + 77, 256 << 20, (1 << 30) + 4096 * 3 + 1, 3 << 30]
bad = 0
done = 0
for si, n in enumerate(sizes):
    bench.CHUNK = 1024 if n >= (2 << 30) else 256
    data = bench.generate("text", n, 1, 0, dev)
    codec = bench.Codec(mhc, n, dev, order=2)
    codec.histogram(data, 0x2020)
    model = codec.build_model()
    want = torch.zeros(1, dtype=torch.int64, device=dev)
    codec.payload_bits(model, codec.counts, want)
    torch.cuda.synchronize()
    want = int(want.item())
    reps = max(launches // len(sizes), 1)
    for r in range(reps):
        codec.encode(model, data, 0x2020)
        rc = codec.lib.mh_dev_status(codec.enc_ws.data_ptr(), codec.stream())
        path = codec.lib.mh_dev_encode_path(codec.enc_ws.data_ptr(), codec.stream())
        nbits = int(codec.nbits[0].item())
        ok = rc == 0 and path == 4 and nbits == want
        if ok and r % 16 == 0:
            codec.decode(model)
            torch.cuda.synchronize()
            ok = codec.lib.mh_dev_status(codec.dec_ws.data_ptr(), codec.stream()) == 0 and torch.equal(codec.decoded, data)
        done += 1
        if not ok:
            bad += 1
            print("size %d launch %d: status %d path %d nbits %d (want %d)" % (n, r, rc, path, nbits, want), flush=True)
    print("size %d: %d launches, %d bad so far" % (n, reps, bad), flush=True)
    del codec, data
    torch.cuda.empty_cache()
print("soak: %d launches, %d bad" % (done, bad))
sys.exit(1 if bad else 0)
