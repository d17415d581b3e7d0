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
# where the files live: tmpfs when there is one (memory-backed, so the figures are the pipeline's and not the
# box's disk and dirty-page throttling: on /tmp the same binary measured between 2 and 5 GB/s from box to box),
# MH_RATE_DIR to choose
d = os.environ.get("MH_RATE_DIR") or ("/dev/shm/mh_cli_rate" if os.path.isdir("/dev/shm") else "/tmp/mh_cli_rate")
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


stages = []
# MH_RATE_PREFAULT=1: the output mapping's pages are allocated before the library call and timed on their own
# ("prefault"), so encode.* / decode.* show upload + kernels + download and nothing of the file system's page allocation
EXTRA_ENV = {"MH_PREFAULT_WAIT": "1"} if os.environ.get("MH_RATE_PREFAULT") else {}


def timed(args):
    """Wall time of the whole process (start-up and HIP initialisation included); the time spent inside
    the library calls comes back on stderr (MH_TIMING=1) and is collected in `stages`."""
    t = time.perf_counter()
    r = subprocess.run([BIN] + args, stderr=subprocess.PIPE, env=dict(os.environ, MH_TIMING="1", **EXTRA_ENV))
    assert r.returncode == 0, (args, r.stderr[-500:])
    wall = time.perf_counter() - t
    inside = [l.split() for l in r.stderr.decode().splitlines() if l.startswith("[mh-timing]")]
    stages.append({"args": " ".join(a for a in args if a.startswith("-")), "wall_s": round(wall, 3),
                   "inside": {l[1]: float(l[4]) for l in inside}})
    return wall


tc = timed([d + "/in", "-o", d + "/c", "-d", d + "/t", "--index", d + "/c.idx"])
tx = timed([d + "/c", "-o", d + "/d", "-x", "-e", d + "/t"])
txi = timed([d + "/c", "-o", d + "/di", "-x", "-e", d + "/t", "--index", d + "/c.idx"])
same = subprocess.run(["cmp", d + "/in", d + "/d"]).returncode == 0 and subprocess.run(["cmp", d + "/in", d + "/di"]).returncode == 0
def pipeline(stage, call):
    """bytes / (upload + device + download) of one library call: the pipeline without the file system's share"""
    ins = stage["inside"]
    t = sum(ins.get("%s.%s" % (call, k), 0.0) for k in ("upload", "device", "download"))
    return round(n / t / 1e9, 2) if t > 0 else None


print({"GiB": gib, "dir": d, "prefault": bool(EXTRA_ENV),
       "pipeline_GBps": {"compress(encode call)": pipeline(stages[0], "encode"), "histogram call": pipeline(stages[0], "histogram"),
                         "decompress": pipeline(stages[1], "decode"), "decompress_indexed": pipeline(stages[2], "decode")}, "compress_s": round(tc, 2), "compress_GBps": round(n / tc / 1e9, 2),
       "decompress_s": round(tx, 2), "decompress_GBps": round(n / tx / 1e9, 2),
       "decompress_indexed_s": round(txi, 2), "decompress_indexed_GBps": round(n / txi / 1e9, 2), "round_trip_ok": same,
       "stages": stages})
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
