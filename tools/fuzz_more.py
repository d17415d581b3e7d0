import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import __graft_entry__ as entry
from oracle import mh_oracle as oracle
import test_gpu_fuzz as F
mhc = entry.load_package()
bad = 0
for seed in range(300):
    rng = np.random.default_rng(50000 + seed)
    n = int(rng.choice([3, 100, 5000, 65537, 262144 + 5, (1 << 20) + 77, (3 << 20) + 1]))
    kindsafe_n = n
    data = F._source(rng, n if n < 40000 else n)
    # kind 5 is a python loop: keep it short
    data = data.tobytes()
    order = int(rng.integers(0, 2))
    chunk = int(rng.choice([256, 512, 1024, 2048, 8192]))
    os.environ["MH_SEGMENT_BYTES"] = str(int(rng.choice([8192, 65536, 1 << 20, 1 << 28])))
    counts = mhc.histogram_o1(data) if order else mhc.histogram_o0(data)
    m = mhc.Model.from_counts(counts, order)
    o = oracle.Model.from_counts(counts, order)
    blob, nbits, idx = m.compress(data, chunk_symbols=chunk)
    ref, ref_bits = o.compress(data)
    ok = (nbits, blob) == (ref_bits, ref) and m.decompress(blob, index=idx, chunk_symbols=chunk, n_symbols=len(data)) == data and m.decompress(blob) == data
    if len(data) >= 2048 and order == 1:
        ok = ok and mhc.Model.from_data(data, 1).compress(data)[0] == blob
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, n, order, chunk, os.environ["MH_SEGMENT_BYTES"], flush=True)
print("done, mismatches:", bad)
