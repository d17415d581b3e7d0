"""The index builder's fast path (mh_tile.hip index_tile_kernel): streams that come WITHOUT an index — what the
reference writes, src/coding.cpp:35-59 — get chunk index and fine index from 128 adjacent 352-bit segments per wave.  Every case
compares with positions computed from the oracle's code lengths on the host; which way the index was built is asserted by
path code (5 = tiles, 1 = the segment iteration it replaces), never by the clock."""
import os

import numpy as np
import pytest

import __graft_entry__ as entry

pytestmark = pytest.mark.gpu

IDX_SEGMENTS, IDX_TILES = 1, 5


@pytest.fixture(scope="module")
def mhc():
    mod = entry.load_package()
    mod.lib()
    assert mod.device_count() >= 1
    return mod


def zipf_bytes(n, seed, s=1.1, k=256):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, k + 1) ** s
    return rng.choice(k, size=n, p=w / w.sum()).astype(np.uint8)


def text_like(n, seed):
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, rng.integers(2, 9)).astype(np.uint8)) for _ in range(300)]
    out = bytearray()
    while len(out) < n:
        out += words[int(rng.integers(0, 300))] + (b". " if rng.random() < 0.1 else b" ")
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()


def expected_entries(lens, data, chunk, prev0=0x20):
    """(chunk index, fine index) of the stream of `data` under code lengths lens[prev * 256 + sym]."""
    d = data.astype(np.int64)
    prev = np.concatenate([[prev0], d[:-1]])
    pos = np.concatenate([[0], np.cumsum(lens[prev * 256 + d])[:-1]])
    j = np.arange(0, d.size, chunk)
    index = ((prev[j].astype(np.uint64) << np.uint64(56)) | pos[j].astype(np.uint64))
    f = np.arange(0, d.size, 64)
    fine = ((prev[f] << 24) | (pos[f] & 0xFFFFFF)).astype(np.uint32)
    return index, fine


def build(mhc, m, payload, nbits, chunk, prev0=0x20, with_fine=True):
    lib = mhc.lib()
    pl = np.frombuffer(payload, dtype=np.uint8)
    d_pl = mhc.DeviceBuffer(pl.size + 64, init=np.concatenate([pl, np.zeros(64, dtype=np.uint8)]))
    icap, fcap = nbits // chunk + 2, nbits // 64 + 2
    d_idx, d_fine, d_ns = mhc.DeviceBuffer(icap * 8), mhc.DeviceBuffer(fcap * 4), mhc.DeviceBuffer(8)
    iws = int(lib.mh_dev_build_index_workspace(nbits))
    d_iws = mhc.DeviceBuffer(iws)
    if with_fine:
        rc = lib.mh_dev_build_index_fine(m.handle, d_pl.ptr, nbits, prev0, d_idx.ptr, icap, chunk, d_fine.ptr, fcap, d_ns.ptr, d_iws.ptr, iws, None)
    else:
        rc = lib.mh_dev_build_index(m.handle, d_pl.ptr, nbits, prev0, d_idx.ptr, icap, chunk, d_ns.ptr, d_iws.ptr, iws, None)
    assert rc == 0
    return (lib.mh_dev_status(d_iws.ptr, None), lib.mh_dev_index_path(d_iws.ptr, None), int(d_ns.download(np.uint64)[0]),
            d_idx.download(np.uint64), d_fine.download(np.uint32))


@pytest.mark.parametrize("kind,n,chunk", [("zipf", (12 << 20) + 1001, 1024), ("zipf", 300_000, 256), ("text", (6 << 20) + 77, 512),
                                          ("zipf", 4096 * 64 * 3, 4096), ("text", 262_144 + 5, 8192)])
def test_tiles_build_the_encoder_s_index_for_a_stream_without_one(mhc, oracle, kind, n, chunk):
    """Stream and table from the oracle (= the reference's files); the builder's chunk index and fine index equal the
    positions that follow from the code lengths, for sizes on every side of the segment / tile boundaries."""
    data = zipf_bytes(n, n & 255) if kind == "zipf" else text_like(n, n & 255)
    om = oracle.Model.from_data(data.tobytes(), 1)
    blob, nbits = om.compress(data.tobytes())
    m = mhc.Model.from_table(om.table_bytes())
    lens = np.asarray(om.codes()[0]).astype(np.int64)
    want_idx, want_fine = expected_entries(lens, data, chunk)
    st, path, ns, idx, fine = build(mhc, m, blob[1:], nbits, chunk)
    assert (st, ns) == (0, n)
    assert path == (IDX_TILES if nbits >= 1 << 20 and int(lens.max()) <= 15 else IDX_SEGMENTS), path
    assert np.array_equal(idx[:want_idx.size], want_idx)
    assert np.array_equal(fine[:want_fine.size], want_fine)
    # the same stream through the path it replaces gives the same entries
    os.environ["MH_INDEX_NO_TILES"] = "1"
    try:
        st2, path2, ns2, idx2, fine2 = build(mhc, m, blob[1:], nbits, chunk)
    finally:
        del os.environ["MH_INDEX_NO_TILES"]
    assert (st2, path2, ns2) == (0, IDX_SEGMENTS, n)
    assert np.array_equal(idx2[:want_idx.size], want_idx) and np.array_equal(fine2[:want_fine.size], want_fine)


def test_tiles_without_a_fine_index_and_with_another_start_context(mhc, oracle):
    """mh_dev_build_index (no fine index asked for) and a stream that starts in another context than ' ' (a shard)."""
    data = zipf_bytes(3 << 20, 5)
    om = oracle.Model.from_data(b"\x07" + data.tobytes(), 1)
    m = mhc.Model.from_table(om.table_bytes())
    lens = np.asarray(om.codes()[0]).astype(np.int64)
    # a stream in start context 7 = the oracle's stream of b"\x07" + data without its first code
    blob, nbits = om.compress(b"\x07" + data.tobytes())
    skip = int(lens[0x20 * 256 + 7])
    bits = np.unpackbits(np.frombuffer(blob[1:], dtype=np.uint8))[skip:nbits]
    payload = np.packbits(bits).tobytes()
    want_idx, _ = expected_entries(lens, data, 1024, prev0=7)
    st, path, ns, idx, _ = build(mhc, m, payload, bits.size, 1024, prev0=7, with_fine=False)
    assert (st, path, ns) == (0, IDX_TILES, data.size)
    assert np.array_equal(idx[:want_idx.size], want_idx)


def test_tiles_give_way_when_the_warm_up_does_not_synchronise(mhc, oracle):
    """Near-uniform bytes whose codes are 7, 8 and 9 bits long and assigned differently in every context: two decodes merge
    about once in 256 symbols, a 256-bit warm-up synchronises one segment in nine — the fast path gives up after its first
    repair pass and the segment iteration (or one of its fallbacks) builds the index; the result is the same."""
    c = np.arange(256, dtype=np.uint64)
    counts = (100000 + ((c[None, :] * 7 + c[:, None] * 13) % 5)).astype(np.uint64)
    counts[:, 0] *= 2                                              # one 7-bit code per context: no code-length lattice
    om = oracle.Model.from_counts(counts.reshape(-1), 1)
    lens = np.asarray(om.codes()[0]).astype(np.int64)
    assert set(np.unique(lens)) >= {7, 8} and int(lens.max()) <= 15
    n = 8 << 20
    data = np.random.default_rng(3).integers(0, 256, n, dtype=np.uint8)
    blob, nbits = om.compress(data.tobytes())
    m = mhc.Model.from_table(om.table_bytes())
    want_idx, want_fine = expected_entries(lens, data, 1024)
    st, path, ns, idx, fine = build(mhc, m, blob[1:], nbits, 1024)
    assert (st, ns) == (0, n) and path != IDX_TILES, path
    assert np.array_equal(idx[:want_idx.size], want_idx) and np.array_equal(fine[:want_fine.size], want_fine)


def test_tiles_report_a_stream_that_does_not_belong_to_the_table(mhc, oracle):
    """A payload with a flipped bit re-synchronises (Huffman streams do), but its symbol count or its last code no longer
    fit; a payload cut short ends inside a code: MH_ERR_CORRUPT either way, or a decode that differs — never a wild access."""
    data = zipf_bytes(2 << 20, 9)
    om = oracle.Model.from_data(data.tobytes(), 1)
    blob, nbits = om.compress(data.tobytes())
    m = mhc.Model.from_table(om.table_bytes())
    st, path, ns, _, _ = build(mhc, m, blob[1:], nbits - 3, 1024)       # ends inside the last code
    assert path == IDX_TILES and (st == mhc.MH_ERR_CORRUPT or ns != data.size)


def _random_source(seed):
    """Seeded sources of many shapes: alphabets of 3..256 symbols, Zipf exponents 0.3..2.5, iid or first-order Markov
    (a random permutation of the ranks per context: the decode then depends on the context at every step)."""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([3, 5, 16, 40, 64, 100, 256]))
    s = float(rng.uniform(0.3, 2.5))
    n = int(rng.integers(260_000, 900_000))
    w = 1.0 / np.arange(1, k + 1) ** s
    w /= w.sum()
    if seed % 3 == 0:                                            # Markov: every context ranks the symbols differently
        perms = [rng.permutation(k).tolist() for _ in range(k)]
        ranks = rng.choice(k, size=n, p=w).tolist()
        out = bytearray(n)
        prev = 0
        for i, r in enumerate(ranks):
            prev = perms[prev][r]
            out[i] = prev
        return np.frombuffer(bytes(out), dtype=np.uint8).copy()
    return rng.choice(k, size=n, p=w).astype(np.uint8)


@pytest.mark.parametrize("seed", range(24))
def test_index_builder_on_random_sources_matches_the_oracle_s_positions(mhc, oracle, seed):
    """Differential run over seeded random sources (tools/fuzz_index_free.py's idea, as a test): whichever way the index is
    built — the fast path over tiles where the model allows it, the segment iteration or one of its fallbacks elsewhere — chunk
    index, fine index and symbol count equal what the oracle's code lengths say, and the host-buffer decode of the
    oracle's stream (no index: the reference's file format) gives the input back."""
    data = _random_source(seed)
    om = oracle.Model.from_data(data.tobytes(), 1)
    blob, nbits = om.compress(data.tobytes())
    m = mhc.Model.from_table(om.table_bytes())
    lens = np.asarray(om.codes()[0]).astype(np.int64)
    chunk = [256, 1024, 4096][seed % 3]
    want_idx, want_fine = expected_entries(lens, data, chunk)
    st, path, ns, idx, fine = build(mhc, m, blob[1:], nbits, chunk)
    assert (st, ns) == (0, data.size), (st, ns, path)
    assert np.array_equal(idx[:want_idx.size], want_idx) and np.array_equal(fine[:want_fine.size], want_fine), path
    assert m.decompress(blob) == data.tobytes()
