"""Random streams that defeat the segment iteration of the index builder (GPU box):
   python3 tools/fuzz_index_free.py [first_seed [count]]
Each case: a first-order walk over a few symbols in which every context has one or two successors (so decodes
from different contexts rarely or never merge), run lengths and counts random; the oracle writes the stream
(no index), the library decodes it and must return the data."""
import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
from oracle import mh_oracle as oracle
mhc = entry.load_package()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    k = int(rng.choice([2, 3, 4, 7, 16, 40]))
    syms = rng.choice(256, size=k, replace=False).astype(np.uint8)
    n = int(rng.integers(1 << 20, 5 << 20))
    kind = int(rng.integers(0, 3))
    if kind == 0:      # cycle with random dwell: each symbol is followed by itself or the next one
        dwell = int(rng.choice([1, 2, 50, 4096]))
        steps = (rng.random(n) < 1.0 / dwell).astype(np.int64)
        idx = np.cumsum(steps) % k
    elif kind == 1:    # two successors per symbol chosen at random (a random 2-regular walk)
        nxt = np.stack([rng.integers(0, k, k), rng.integers(0, k, k)], axis=1)
        bits = rng.integers(0, 2, n)
        # the walk is sequential: advance it a byte of choices at a time through precomputed 8-step transitions
        T8 = np.zeros((k, 256), dtype=np.int64); O8 = np.zeros((k, 256, 8), dtype=np.int64)
        for s0 in range(k):
            for b in range(256):
                c = s0
                for j in range(8):
                    c = nxt[c, (b >> j) & 1]
                    O8[s0, b, j] = c
                T8[s0, b] = c
        nb = n // 8
        bytes8 = np.packbits(bits[:nb * 8].reshape(-1, 8), axis=1, bitorder="little").reshape(-1)
        out = np.empty((nb, 8), dtype=np.int64)
        c = 0
        for i in range(nb):
            out[i] = O8[c, bytes8[i]]
            c = T8[c, bytes8[i]]
        idx = out.reshape(-1)
        n = idx.size
    else:              # pure cycle
        idx = np.arange(n) % k
    data = syms[idx].tobytes()
    om = oracle.Model.from_data(data, 1)
    blob, nbits = om.compress(data)
    m = mhc.Model.from_table(om.table_bytes())
    t0 = time.perf_counter()
    ok = m.decompress(blob) == data
    dt = time.perf_counter() - t0
    print("seed %d kind %d k %d n %d maxlen %d ratio %.3f  %s  %.2f s" % (seed, kind, k, n, int(np.asarray(om.codes()[0]).max()), nbits / 8 / n, "ok" if ok else "MISMATCH", dt), flush=True)
    bad += 0 if ok else 1
print("done, mismatches:", bad)
