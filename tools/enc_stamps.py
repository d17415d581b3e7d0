#!/usr/bin/env python3
"""Diagnostic build of the region encoder (make -C markov-huffman-coding_amd/csrc exp TAG=encstamp EXPFLAGS=-DMH_ENC_STAMP,
run with MH_LIB=.../libmhc_encstamp.so): shares of a round's phases.  Never quote its run time."""
import os, sys
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
codec.histogram(data, 0x20); model = codec.build_model(); codec.encode(model, data, 0x20)
torch.cuda.synchronize()
seg = codec.enc_ws[:64].cpu().numpy().view(np.uint64)[1:6].astype(float)
names = ["flush of the previous round", "barrier 1", "exchange+deposits issued", "lookups+pack+scan of the next round", "barrier 2"]
rounds = size / 16384
print("shares: " + ", ".join("%s %.1f%%" % (n, 100 * v / seg.sum()) for n, v in zip(names, seg)))
print("cycles per round per wave: " + ", ".join("%s %.0f" % (n, v / rounds / 16) for n, v in zip(names, seg)))
