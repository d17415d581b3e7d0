"""Seeded randomised parity: random source shapes x sizes x chunk sizes x segment sizes, every case
compared byte for byte with the oracle (stream, table, index-free and indexed decode)."""
import numpy as np
import pytest

import __graft_entry__ as entry

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhc():
    m = entry.load_package()
    if m.device_count() < 1:
        pytest.skip("no HIP device")
    return m


def _source(rng, n):
    kind = rng.integers(0, 8)
    if kind == 0:
        k = int(rng.integers(1, 257))
        return rng.integers(0, k, n).astype(np.uint8)
    if kind == 1:
        p = float(rng.uniform(0.05, 0.9))
        return np.minimum(rng.geometric(p, n) - 1, 255).astype(np.uint8)
    if kind == 2:
        s = float(rng.uniform(0.6, 2.5))
        w = 1.0 / np.arange(1, 257) ** s
        perm = rng.permutation(256)
        return perm[rng.choice(256, size=n, p=w / w.sum())].astype(np.uint8)
    if kind == 3:
        return np.full(n, int(rng.integers(0, 256)), dtype=np.uint8)
    if kind == 4:
        a, b = rng.integers(0, 256, 2)
        return np.where(np.arange(n) % 2 == 0, a, b).astype(np.uint8)
    if kind == 5:                                    # order-1 structure: next = f(prev) + small noise
        x = np.zeros(n, dtype=np.uint8)
        noise = rng.integers(0, 4, n)
        for i in range(1, n):
            x[i] = (int(x[i - 1]) * 7 + 3 + int(noise[i])) & 255
        return x
    if kind == 6:                                    # a few very rare symbols among a skewed bulk: long codes
        x = np.minimum(rng.geometric(0.5, n) - 1, 255).astype(np.uint8)
        if n:
            x[rng.integers(0, n, max(n // 5000, 1))] = rng.integers(100, 256, max(n // 5000, 1)).astype(np.uint8)
        return x
    return rng.integers(0, 256, n).astype(np.uint8)


SIZES = [0, 1, 2, 15, 16, 17, 255, 256, 257, 4095, 4096, 4097, 20000, 70001, 300003, 1 << 20]


@pytest.mark.parametrize("seed", range(48))
def test_random_case(mhc, oracle, monkeypatch, seed):
    rng = np.random.default_rng(1000 + seed)
    n = SIZES[seed % len(SIZES)]
    if n > 100000 and seed % 3 == 0:
        n += int(rng.integers(0, 4096))
    data = _source(rng, n if (seed // len(SIZES)) != 5 or n < 30000 else 30000).tobytes()
    n = len(data)
    order = int(rng.integers(0, 2))
    chunk = int(rng.choice([256, 512, 1024, 4096]))
    if rng.integers(0, 2):
        monkeypatch.setenv("MH_SEGMENT_BYTES", str(int(rng.choice([8192, 16384, 65536]))))
    if order == 0 and n == 0:
        pytest.skip("empty -h table cannot be decoded (reference crashes, SURVEY 8c)")
    counts = mhc.histogram_o1(data) if order else mhc.histogram_o0(data)
    assert np.array_equal(counts, oracle.histogram_o1(data, 0x20) if order else oracle.histogram_o0(data))
    m = mhc.Model.from_counts(counts, order)
    o = oracle.Model.from_counts(counts, order)
    assert m.table_bytes() == o.table_bytes()
    blob, nbits, idx = m.compress(data, chunk_symbols=chunk)
    ref, ref_bits = o.compress(data)
    assert (nbits, blob) == (ref_bits, ref)
    assert m.decompress(blob, index=idx, chunk_symbols=chunk, n_symbols=n) == data
    assert m.decompress(blob) == data
    if n >= 2048 and order == 1:                     # device-built model = host-built model
        md = mhc.Model.from_data(data, 1)
        assert md.compress(data)[0] == blob


def l2_layout_case(mhc, oracle, seed):
    """One random case for the decoder's L2 ("direct") layout: 256 contexts whose second-level tables cannot
    fit LDS (Zipf-shaped counts with a random exponent -> longest code 9..16 bits, a random rank permutation
    per context so that the contexts' tables differ), data that mixes the model's own distribution with runs
    of the longest codes, a few MiB so that the lanes run their K full chunks.  Returns a description."""
    rng = np.random.default_rng(70000 + seed)
    s = float(rng.uniform(0.85, 1.9))
    base = np.floor((1 << 22) / np.arange(1, 257) ** s) + 1
    counts = np.empty((256, 256), dtype=np.uint64)
    perms = np.empty((256, 256), dtype=np.int64)
    for c in range(256):
        perms[c] = rng.permutation(256) if rng.random() < 0.5 else np.roll(np.arange(256), int(rng.integers(256)))
        counts[c, perms[c]] = base.astype(np.uint64)
    counts = counts.reshape(-1)
    m = mhc.Model.from_counts(counts, 1)
    om = oracle.Model.from_counts(counts, 1)
    lens = np.asarray(om.codes()[0]).reshape(256, 256)
    n = int(rng.integers(2 << 20, 6 << 20)) + int(rng.integers(0, 2000))
    # a first-order walk: next symbol = the context's r-th most likely symbol, r Zipf-distributed or, in
    # bursts, drawn from the tail (the longest codes)
    w = 1.0 / np.arange(1, 257) ** s
    ranks = rng.choice(256, size=n, p=w / w.sum())
    tail = rng.random(n) < float(rng.choice([0.0, 0.05, 0.5]))
    ranks[tail] = rng.integers(200, 256, size=int(tail.sum()))
    data = np.empty(n, dtype=np.uint8)
    prev = 0x20
    # vectorising a Markov walk needs the previous output: do it in blocks with a python loop over a coarse
    # stride only (the context of the block's first symbol), the rest uses the permutation of that context's
    # successor — cheap and still context-dependent
    blk = 4096
    for off in range(0, n, blk):
        r = ranks[off:off + blk]
        out = perms[prev][r]
        # re-map every symbol through the permutation of its true predecessor for a prefix of the block
        k = min(64, len(r))
        p = prev
        for i in range(k):
            out[i] = perms[p][r[i]]
            p = int(out[i])
        data[off:off + blk] = out
        prev = int(out[-1])
    data = data.tobytes()
    chunk = int(rng.choice([256, 512, 1024, 2048]))
    blob, nbits, idx = m.compress(data, chunk_symbols=chunk)
    ref, ref_bits = om.compress(data)
    assert nbits == ref_bits and blob == ref
    assert m.decompress(blob, index=idx, chunk_symbols=chunk, n_symbols=n) == data
    return "s=%.2f maxlen=%d n=%d chunk=%d ratio=%.3f" % (s, int(lens.max()), n, chunk, nbits / 8 / n)


@pytest.mark.parametrize("seed", range(6))
def test_random_l2_layout_case(mhc, oracle, seed):
    l2_layout_case(mhc, oracle, seed)
