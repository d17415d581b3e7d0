"""Random sources through the two-pass decode of streams without an index (GPU box):
   python3 tools/fuzz_stream.py [first_seed [count]]
Each case: a seeded source of some shape — alphabet of 2..256 symbols, Zipf exponent 0.2..3, iid or first-order Markov with a
random permutation of the ranks per context, optionally a few dozen planted rare pairs (codes of 16..25 bits), optionally long
runs of one symbol (segments of hundreds of symbols), sizes on both sides of the tile boundaries — the ORACLE writes the stream
(= the reference's file, no index), the library decodes it through the host-buffer call (which takes the two passes where they
apply) and through the two device calls, and both must return the data; where the two passes do not apply the path code says
so and the fallback must return the data.  Prints one line per case and a summary of the path codes."""
import os, sys, time, collections
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
from oracle import mh_oracle as oracle
mhc = entry.load_package()
lib = mhc.lib()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
paths = collections.Counter()
for seed in range(first, first + count):
    rng = np.random.default_rng(77000 + seed)
    k = int(rng.choice([2, 3, 5, 16, 40, 64, 100, 200, 256]))
    s = float(rng.uniform(0.2, 3.0))
    n = int(rng.choice([150_000, 400_000, 1 << 20, (1 << 21) + 13, 45056 * 8 * 3, (3 << 20) + 7777]))
    w = 1.0 / np.arange(1, k + 1) ** s
    w /= w.sum()
    syms = rng.permutation(256)[:k].astype(np.uint8)
    ranks = rng.choice(k, size=n, p=w)
    markov = seed % 3 == 0
    if markov:                                                    # every context ranks the symbols differently
        perms = np.stack([rng.permutation(k) for _ in range(k)])
        idx = np.empty(n, dtype=np.int64)
        prev = 0
        rl = ranks.tolist()
        for i in range(n):
            prev = int(perms[prev, rl[i]])
            idx[i] = prev
    else:
        idx = ranks
    data = syms[idx].copy()
    if seed % 4 == 1:                                             # runs: hundreds of symbols per segment
        for _ in range(int(rng.integers(1, 20))):
            a = int(rng.integers(0, n - 1)); data[a:a + int(rng.integers(100, 60000))] = syms[0]
    if seed % 5 == 2 and k >= 16:                                 # planted rare pairs: long codes
        pos = rng.integers(1, n - 1, int(rng.integers(1, 40)))
        data[pos] = syms[rng.integers(k // 2, k, pos.size)]
        data[pos - 1] = syms[0]
    raw = data.tobytes()
    om = oracle.Model.from_data(raw, 1)
    blob, nbits = om.compress(raw)
    m = mhc.Model.from_table(om.table_bytes())
    t0 = time.perf_counter()
    ok_host = m.decompress(blob) == raw
    path_host = lib.mh_last_index_path()
    # the two device calls by themselves
    pl = np.frombuffer(blob[1:], dtype=np.uint8)
    d_pl = mhc.DeviceBuffer(pl.size + 64, init=np.concatenate([pl, np.zeros(64, dtype=np.uint8)]))
    d_ns = mhc.DeviceBuffer(8)
    iws = int(lib.mh_dev_build_index_workspace(nbits))
    d_iws = mhc.DeviceBuffer(iws)
    rc1 = lib.mh_dev_decode_stream_states(m.handle, d_pl.ptr, nbits, 0x20, d_ns.ptr, d_iws.ptr, iws, None)
    st1, path = lib.mh_dev_status(d_iws.ptr, None), lib.mh_dev_index_path(d_iws.ptr, None)
    ok_dev = True
    if path == 6:
        d_out = mhc.DeviceBuffer(n + 256, init=np.full(n + 256, 0xA5, dtype=np.uint8))
        rc2 = lib.mh_dev_decode_stream_emit(m.handle, d_pl.ptr, nbits, 0x20, d_out.ptr, n, d_iws.ptr, iws, None)
        st2 = lib.mh_dev_status(d_iws.ptr, None)
        out = d_out.download()
        ok_dev = rc1 == 0 and rc2 == 0 and st1 == 0 and st2 == 0 and int(d_ns.download(np.uint64)[0]) == n and out[:n].tobytes() == raw and bool(np.all(out[n:] == 0xA5))
    else:
        ok_dev = rc1 == 0 and st1 == 0
    dt = time.perf_counter() - t0
    paths[(path_host, path)] += 1
    ok = ok_host and ok_dev
    print("seed %d k %d s %.2f n %d %s maxlen %d ratio %.3f  host path %d dev path %d  %s  %.2f s" %
          (seed, k, s, n, "markov" if markov else "iid", int(np.asarray(om.codes()[0]).max()), nbits / 8 / n, path_host, path,
           "ok" if ok else "MISMATCH host=%s dev=%s" % (ok_host, ok_dev), dt), flush=True)
    bad += 0 if ok else 1
print("paths (host, device):", dict(paths))
print("done, mismatches:", bad)
