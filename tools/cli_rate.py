#!/usr/bin/env python3
"""End-to-end rate of the CLI (file in page cache -> mapped -> PCIe -> HBM -> PCIe -> mapped file):
   python tools/cli_rate.py [GiB]     (run on the GPU box, from the repo root)"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(gib * (1 << 30))
d = "/tmp/mh_cli_rate"
os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(5)
w = 1.0 / np.arange(1, 257) ** 1.1
block = rng.choice(256, size=64 << 20, p=w / w.sum()).astype(np.uint8)
with open(d + "/in", "wb") as f:
    left = n
    while left > 0:
        k = min(left, block.size)
        f.write(np.roll(block, left % 977)[:k].tobytes())
        left -= k
BIN = os.path.join(ROOT, "bin", "markovhuffman")


def timed(args):
    t = time.perf_counter()
    r = subprocess.run([BIN] + args)
    assert r.returncode == 0, args
    return time.perf_counter() - t


tc = timed([d + "/in", "-o", d + "/c", "-d", d + "/t", "--index", d + "/c.idx"])
tx = timed([d + "/c", "-o", d + "/d", "-x", "-e", d + "/t"])
txi = timed([d + "/c", "-o", d + "/di", "-x", "-e", d + "/t", "--index", d + "/c.idx"])
same = subprocess.run(["cmp", d + "/in", d + "/d"]).returncode == 0 and subprocess.run(["cmp", d + "/in", d + "/di"]).returncode == 0
print({"GiB": gib, "compress_s": round(tc, 2), "compress_GBps": round(n / tc / 1e9, 2),
       "decompress_s": round(tx, 2), "decompress_GBps": round(n / tx / 1e9, 2),
       "decompress_indexed_s": round(txi, 2), "decompress_indexed_GBps": round(n / txi / 1e9, 2), "round_trip_ok": same})
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
