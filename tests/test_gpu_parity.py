"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the
CPU oracle on the same inputs and against the golden vectors made with the genuine reference.
Integer/byte work: the bar is bit-exact everywhere."""
import numpy as np
import pytest

import __graft_entry__ as entry
from conftest import check_against_golden, golden, golden_names

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhc():
    mod = entry.load_package()
    import os
    if not os.path.exists(mod.LIB_PATH):
        entry.build()          # hipcc is on the GPU box too; normally the built library travels with the repo
    mod.lib()
    assert mod.device_count() >= 1, "GPU tests need a device; the codec has no CPU fallback"
    return mod


def zipf_bytes(n, seed, s=1.1):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, 257) ** s
    return rng.choice(256, size=n, p=w / w.sum()).astype(np.uint8).tobytes()


def text_like(n, seed):
    rng = np.random.default_rng(seed)
    words = [b"lorem", b"ipsum", b"dolor", b"sit", b"amet", b"consectetur", b"adipiscing", b"elit", b"sed", b"do",
             b"eiusmod", b"tempor", b"incididunt", b"ut", b"labore", b"et", b"dolore", b"magna", b"aliqua"]
    out = bytearray()
    while len(out) < n:
        out += words[int(rng.integers(len(words)))] + (b". " if rng.random() < 0.1 else b" ")
    return bytes(out[:n])


# ------------------------------------------------------------------ histogram

@pytest.mark.parametrize("name", golden_names())
def test_histogram_matches_oracle_on_golden_inputs(mhc, oracle, name):
    data = golden()[name]["data"]
    assert np.array_equal(mhc.histogram_o1(data), oracle.histogram_o1(data))
    assert np.array_equal(mhc.histogram_o0(data), oracle.histogram_o0(data))


@pytest.mark.parametrize("n", [1, 15, 16, 17, 31, 4095, 4096, 65537, 1 << 20, (1 << 22) + 5])
def test_histogram_ragged_sizes(mhc, oracle, n):
    data = zipf_bytes(n, n)
    assert np.array_equal(mhc.histogram_o1(data), oracle.histogram_o1(data))
    assert np.array_equal(mhc.histogram_o0(data), oracle.histogram_o0(data))


def test_histogram_counter_overflow_path(mhc, oracle):
    """One pair count far above 32768 exercises the packed 15-bit LDS counters' guard-bit fix-up."""
    data = (b"\x00" * (3 << 20)) + zipf_bytes(1 << 20, 5) + (b"ab" * (1 << 20))
    h = mhc.histogram_o1(data)
    assert np.array_equal(h, oracle.histogram_o1(data))
    assert h[0] > (3 << 20) - 2
    assert np.array_equal(mhc.histogram_o0(data), oracle.histogram_o0(data))


@pytest.mark.parametrize("n", [0, 5, 4096, (3 << 20) + 11])
def test_histogram_with_workspace_equals_atomic_flush(mhc, oracle, n):
    """mh_dev_histogram_o1 with its optional workspace (per-workgroup slabs + reduce kernel) and without
    it (64-bit atomics) give the same counters, both equal to the oracle's; includes a 16-bit counter
    overflow (a run of one pair longer than 32767)."""
    import ctypes as C
    lib = mhc.lib()
    data = bytearray(zipf_bytes(n, 91))
    if n > (1 << 20):
        data[1000:1000 + 70000] = b"\x07" * 70000
    data = bytes(data)
    d_data = mhc.DeviceBuffer(n + 16, init=np.frombuffer(data + bytes(16), dtype=np.uint8))
    wsb = lib.mh_dev_histogram_workspace(n)
    d_ws = mhc.DeviceBuffer(max(wsb, 16))
    got = []
    for ws, nb in ((d_ws.ptr, wsb), (None, 0)):
        d_counts = mhc.DeviceBuffer(65536 * 8, init=np.full(65536, 7, dtype=np.uint64))
        mhc._check(lib.mh_dev_histogram_o1(d_data.ptr, n, 0x41, d_counts.ptr, ws, nb, None), "hist")
        got.append(d_counts.download(np.uint64))
    ref = oracle.histogram_o1(data, 0x41)
    assert np.array_equal(got[0], ref) and np.array_equal(got[1], ref)


def test_histogram_prev0(mhc, oracle):
    data = zipf_bytes(100000, 9)
    for prev0 in (0, 0x20, 0xFF):
        assert np.array_equal(mhc.histogram_o1(data, prev0), oracle.histogram_o1(data, prev0))


# ------------------------------------------------------------------ device tree build

@pytest.mark.parametrize("kind", ["zipf", "text", "uniform", "skewed", "ipsum", "wiki_html", "kat3", "one_Z", "empty"])
def test_device_tree_build_equals_host_build(mhc, oracle, kind):
    """mh_dev_model_from_counts (tree_build_kernel + tree_pack_kernel) must produce, bit for bit, the
    images the host build produces from the same counts — and through its lazy host mirror the same
    table file and codes as the reference."""
    if kind == "zipf":
        data = zipf_bytes(1 << 22, 31)
    elif kind == "text":
        data = text_like(1 << 20, 32)
    elif kind == "uniform":
        data = np.random.default_rng(33).integers(0, 256, 1 << 22, dtype=np.uint8).tobytes()
    elif kind == "skewed":
        data = _skewed(1 << 21, 34)
    elif kind == "ipsum":
        data = golden()["input_ipsum.txt"]["data"]
    elif kind == "wiki_html":
        data = golden()["input_wiki_cpp.html"]["data"]
    else:
        data = golden()[kind]["data"]
    counts = oracle.histogram_o1(data)
    host = mhc.Model.from_counts(counts, 1)
    d_counts = mhc.DeviceBuffer(65536 * 8, counts)
    dev = mhc.Model.from_device_counts(d_counts.ptr, 1)
    assert dev.decode_layout() == host.decode_layout()
    assert dev.max_code_len == host.max_code_len
    for which in range(8):
        assert dev.image(which) == host.image(which), "image %d differs" % which
    o = oracle.Model.from_counts(counts, 1)
    assert dev.table_bytes() == o.table_bytes()          # through the lazily built host mirror
    ld, cd = dev.codes()
    lo, co = o.codes()
    assert np.array_equal(ld, lo) and np.array_equal(cd, co)
    if data:
        blob, nbits, idx = dev.compress(data, chunk_symbols=256)
        assert blob == o.compress(data)[0]
        assert dev.decompress(blob, index=idx, chunk_symbols=256, n_symbols=len(data)) == data


@pytest.mark.parametrize("shape", ["distinct", "one_tie_late", "workspace", "wide_keys"])
def test_device_tree_build_register_heap(mhc, oracle, shape):
    """tree_build_kernel keeps the reference's heap in the wave's registers (RegHeap, mh_tree.hip) with 32-bit
    keys when a context's total count fits and 64-bit keys otherwise.  Large pairwise distinct counts (a
    16 GiB histogram's shape), a tie between a leaf and a merged node in every third context, counts
    above 2^32 (`wide_keys`), and the caller-workspace entry point (`workspace`)."""
    rng = np.random.default_rng(7)
    w = 1.0 / np.arange(1, 257) ** 1.1
    p = np.outer(w, w).ravel()
    counts = rng.permutation(65536).astype(np.uint64) + np.floor(p / p.sum() * 2 ** 34).astype(np.uint64) * np.uint64(65536)
    if shape == "wide_keys":
        counts = counts * np.uint64(1 << 12) + np.uint64(5)
    if shape == "one_tie_late":
        for c in range(0, 256, 3):       # make one symbol weigh exactly what the two lightest weigh together
            row = counts[c * 256:(c + 1) * 256]
            order = np.argsort(row)
            row[order[2]] = row[order[0]] + row[order[1]]
    host = mhc.Model.from_counts(counts, 1)
    d_counts = mhc.DeviceBuffer(65536 * 8, counts)
    if shape == "workspace":
        wsb = mhc.lib().mh_dev_model_workspace(1)
        d_ws = mhc.DeviceBuffer(wsb)
        dev = mhc.Model.from_device_counts_ws(d_counts.ptr, 1, d_ws.ptr, wsb)
    else:
        dev = mhc.Model.from_device_counts(d_counts.ptr, 1)
    assert dev.decode_layout() == host.decode_layout()
    for which in range(8):
        assert dev.image(which) == host.image(which), "image %d differs" % which
    assert dev.table_bytes() == oracle.Model.from_counts(counts, 1).table_bytes()


# ------------------------------------------------------------------ encode

@pytest.mark.parametrize("name", golden_names())
def test_markov_stream_matches_reference_golden(mhc, name):
    data = golden()[name]["data"]
    m = mhc.Model.from_data(data, order=1)
    blob, nbits, _ = m.compress(data)
    check_against_golden(name, "cm", blob)
    check_against_golden(name, "e", m.table_bytes())


@pytest.mark.parametrize("name", golden_names())
def test_huffman_stream_matches_reference_golden(mhc, name):
    data = golden()[name]["data"]
    m = mhc.Model.from_data(data, order=0)
    blob, nbits, _ = m.compress(data)
    check_against_golden(name, "ch", blob)
    check_against_golden(name, "eh", m.table_bytes())


@pytest.mark.parametrize("n", [1, 7, 8, 9, 63, 64, 511, 512, 8191, 8192, 8193, 16384, 100003, 8192 * 300 + 17])
@pytest.mark.parametrize("kind", ["zipf", "text", "uniform"])
def test_encode_matches_oracle_ragged(mhc, oracle, n, kind):
    if kind == "zipf":
        data = zipf_bytes(n, n + 1)
    elif kind == "text":
        data = text_like(n, n + 2)
    else:
        data = np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8).tobytes()
    m = mhc.Model.from_data(data, 1)
    o = oracle.Model.from_data(data, 1)
    assert m.table_bytes() == o.table_bytes()
    blob, nbits, _ = m.compress(data)
    ref, ref_bits = o.compress(data)
    assert nbits == ref_bits
    assert blob == ref


def test_encode_with_loaded_table_and_missing_symbols(mhc, oracle):
    """`-e table` path (src/main.cpp:137-161): symbols absent from the table are silently skipped
    (src/coding.cpp:72 under NDEBUG); contexts absent from it too."""
    train = text_like(50000, 1)
    table = oracle.Model.from_data(train, 1).table_bytes()
    data = text_like(30000, 2) + b"XYZ\x00\x01" * 50 + text_like(1000, 3)
    m = mhc.Model.from_table(table)
    o = oracle.Model.from_table(table)
    blob, nbits, _ = m.compress(data)
    ref, ref_bits = o.compress(data)
    assert (nbits, blob) == (ref_bits, ref)


def _skewed(n, seed, p=0.5):
    """Geometric symbol distribution: Huffman codes grow to ~25 bits, far beyond the 12-bit LDS table."""
    rng = np.random.default_rng(seed)
    return np.minimum(rng.geometric(p, n) - 1, 255).astype(np.uint8).tobytes()


def test_encode_long_codes_escape_path(mhc, oracle):
    data = _skewed(1 << 20, 3)
    m = mhc.Model.from_data(data, 1)
    assert m.max_code_len > 12
    o = oracle.Model.from_data(data, 1)
    assert m.compress(data)[0] == o.compress(data)[0]
    m0 = mhc.Model.from_data(data, 0)
    assert m0.max_code_len > 12
    assert m0.compress(data)[0] == oracle.Model.from_data(data, 0).compress(data)[0]


def test_encode_multi_round_tiles(mhc, oracle):
    """A table trained on skewed data, applied to data made of its RAREST symbols: almost every code is
    an escape and tiles carry more bits than the LDS staging image holds (several rounds per tile)."""
    train = _skewed(1 << 21, 11)
    table = oracle.Model.from_data(train, 0).table_bytes()
    o = oracle.Model.from_table(table)
    lens, _ = o.codes()
    present = [s for s in range(256) if lens[s] >= 14]
    assert len(present) >= 4
    rng = np.random.default_rng(4)
    data = np.array(present, dtype=np.uint8)[rng.integers(0, len(present), 70000)].tobytes()
    data = data + train[:5000] + data[:8192 * 2 + 3]
    m = mhc.Model.from_table(table)
    blob, nbits, idx = m.compress(data, chunk_symbols=256)
    ref, ref_bits = o.compress(data)
    assert nbits == ref_bits and nbits > 12 * len(data)
    assert blob == ref
    assert m.decompress(blob, index=idx, chunk_symbols=256, n_symbols=len(data)) == data


def test_decode_codes_longer_than_both_table_levels(mhc, oracle):
    """Codes of more than P + h = 16 bits are not resolved by the decode tables.  With enough chunks for
    the K-stream hot loop (which carries no tree walk) such chunks go through the redo pass; here a
    third of the chunks contain one."""
    train = _skewed(1 << 21, 11)
    table = oracle.Model.from_data(train, 0).table_bytes()
    o = oracle.Model.from_table(table)
    m = mhc.Model.from_table(table)
    lens, _ = o.codes()
    rare = [s for s in range(256) if lens[s] > 17]
    assert rare
    data = bytearray(_skewed(1 << 20, 5))
    data = bytearray(bytes(b if lens[b] else 0 for b in data))  # only symbols the table knows
    rng = np.random.default_rng(8)
    for pos in rng.integers(0, len(data), 1500):
        data[pos] = rare[pos % len(rare)]
    data = bytes(data)
    blob, nbits, idx = m.compress(data, chunk_symbols=256)
    ref, ref_bits = o.compress(data)
    assert (nbits, blob) == (ref_bits, ref)
    assert m.decompress(blob, index=idx, chunk_symbols=256, n_symbols=len(data)) == data
    assert m.decompress(blob) == data                           # index rebuilt on the device
    # order 1, model of the data itself: the rare pairs carry the long codes
    m1 = mhc.Model.from_data(data, 1)
    blob1, _, idx1 = m1.compress(data, chunk_symbols=256)
    assert blob1 == oracle.Model.from_data(data, 1).compress(data)[0]
    assert m1.decompress(blob1, index=idx1, chunk_symbols=256, n_symbols=len(data)) == data


@pytest.mark.parametrize("cuts", [(0.5,), (0.31, 0.7), (0.0, 0.5, 0.5)])
def test_sharded_encode_preshifted_shards_or_into_the_reference_stream(mhc, oracle, cuts):
    """SURVEY.md 8e on one card: contiguous shards, local histograms summed, each shard's start bit known
    from its local histogram BEFORE it is encoded, payload emitted pre-shifted (mh_dev_encode_at).  The
    shards OR together into exactly the stream the reference writes for the whole input, and every
    shard decodes from its own buffer."""
    import ctypes as C
    lib = mhc.lib()
    data = zipf_bytes(300000 + 7, 21)
    n = len(data)
    bounds = [0] + [min(n, int(n * c) // 16 * 16) for c in cuts] + [n]
    shards = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
    d_data = mhc.DeviceBuffer(n + 16, init=np.frombuffer(data + bytes(16), dtype=np.uint8))
    base = d_data.ptr.value
    local = []
    for lo, hi in shards:
        prev0 = data[lo - 1] if lo else 0x20
        d_counts = mhc.DeviceBuffer(65536 * 8, init=np.zeros(65536, dtype=np.uint64))
        mhc._check(lib.mh_dev_histogram_o1(C.c_void_p(base + lo), hi - lo, prev0, d_counts.ptr, None, 0, None), "hist")
        local.append(d_counts)
    total_counts = sum(c.download(np.uint64) for c in local)
    assert np.array_equal(total_counts, oracle.histogram_o1(data, 0x20))
    m = mhc.Model.from_counts(total_counts, 1)
    whole = oracle.Model.from_data(data, 1)
    ref_blob, ref_bits = whole.compress(data)
    start = 0
    parts = []
    for (lo, hi), d_counts in zip(shards, local):
        d_bits = mhc.DeviceBuffer(8, init=np.zeros(1, dtype=np.uint64))
        mhc._check(lib.mh_dev_payload_bits(m.handle, d_counts.ptr, d_bits.ptr, None), "payload_bits")
        my_bits = int(d_bits.download(np.uint64)[0])
        ns = hi - lo
        cap = lib.mh_encode_bound(m.handle, ns) + 16
        d_payload = mhc.DeviceBuffer(cap, init=np.full(cap, 0xEE, dtype=np.uint8))
        d_start = mhc.DeviceBuffer(8, init=np.array([start], dtype=np.uint64))
        d_nbits = mhc.DeviceBuffer(8, init=np.zeros(1, dtype=np.uint64))
        nidx = max((ns + 255) // 256, 1)
        d_index = mhc.DeviceBuffer(nidx * 8, init=np.zeros(nidx, dtype=np.uint64))
        wsb = lib.mh_dev_encode_workspace(ns)
        d_ws = mhc.DeviceBuffer(wsb + 64)
        prev0 = data[lo - 1] if lo else 0x20
        mhc._check(lib.mh_dev_encode_at(m.handle, C.c_void_p(base + lo), ns, prev0, d_start.ptr, d_payload.ptr, cap,
                                        d_nbits.ptr, d_index.ptr, 256, d_ws.ptr, wsb, None), "encode_at")
        mhc._check(lib.mh_dev_status(d_ws.ptr, None), "status")
        end = int(d_nbits.download(np.uint64)[0])
        assert end == (start & 7) + my_bits                       # the histogram predicted the length
        payload = d_payload.download()[:(end + 7) // 8].tobytes()
        if start & 7 and payload:
            assert payload[0] >> (8 - (start & 7)) == 0           # room for the predecessor's tail bits
        parts.append((start, payload))
        if ns:                                                     # the shard decodes from its own buffer
            idx = d_index.download(np.uint64)[:(ns + 255) // 256]
            assert int(idx[0]) == (prev0 << 56) | (start & 7)
            assert m.decode(payload, end, prev0, index=idx, chunk_symbols=256, n_symbols=ns) == data[lo:hi]
        start += my_bits
    assert start == ref_bits
    out = bytearray((ref_bits + 7) // 8)
    for s0, payload in parts:
        for i, b in enumerate(payload):
            out[s0 // 8 + i] |= b
    assert bytes(out) == ref_blob[1:]


@pytest.mark.parametrize("seg", [8192, 24576])
@pytest.mark.parametrize("kind", ["zipf", "text", "skewed"])
def test_host_buffer_calls_in_segments(mhc, oracle, monkeypatch, seg, kind):
    """The host-pointer entry points stage their data through the card in segments (bounded device
    footprint).  With tiny segments every few KiB is a seam: the context is carried across it, each
    segment is emitted pre-shifted and OR-merged into the output, index entries are rebased."""
    monkeypatch.setenv("MH_SEGMENT_BYTES", str(seg))
    n = 5 * seg + 4097
    data = {"zipf": lambda: zipf_bytes(n, 33), "text": lambda: text_like(n, 7), "skewed": lambda: _skewed(n, 13)}[kind]()
    for order in (1, 0):
        assert np.array_equal(mhc.histogram_o1(data) if order else mhc.histogram_o0(data),
                              oracle.histogram_o1(data, 0x20) if order else oracle.histogram_o0(data))
        m = mhc.Model.from_data(data, order)
        o = oracle.Model.from_data(data, order)
        blob, nbits, idx = m.compress(data, chunk_symbols=256)
        ref, ref_bits = o.compress(data)
        assert (nbits, blob) == (ref_bits, ref)
        monkeypatch.setenv("MH_SEGMENT_BYTES", str(1 << 28))
        blob_whole, _, idx_whole = m.compress(data, chunk_symbols=256)
        monkeypatch.setenv("MH_SEGMENT_BYTES", str(seg))
        assert blob_whole == blob and np.array_equal(idx_whole, idx)      # same stream, same index
        assert m.decompress(blob, index=idx, chunk_symbols=256, n_symbols=len(data)) == data
        assert m.decompress(blob) == data


# ------------------------------------------------------------------ decode

@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("order", [0, 1])
def test_round_trip_golden_inputs_with_index(mhc, name, order):
    data = golden()[name]["data"]
    if order == 0 and not data:
        pytest.skip("empty -h table cannot be decoded (reference crashes, SURVEY §8c)")
    m = mhc.Model.from_data(data, order)
    blob, nbits, idx = m.compress(data, chunk_symbols=256)
    assert m.decompress(blob, index=idx, chunk_symbols=256, n_symbols=len(data)) == data


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("order", [0, 1])
def test_decode_reference_stream_without_index(mhc, oracle, name, order):
    """Drop-in case: the stream was produced by the reference (here: the oracle, pinned to it) and has
    no index.  Table loaded from the reference's table file."""
    data = golden()[name]["data"]
    if order == 0 and not data:
        pytest.skip("empty -h table")
    o = oracle.Model.from_data(data, order)
    blob, _ = o.compress(data)
    m = mhc.Model.from_table(o.table_bytes())
    assert m.decompress(blob) == data


@pytest.mark.parametrize("kind", ["zipf", "text", "uniform", "skewed"])
def test_index_free_decode_large(mhc, oracle, kind):
    """N1: a stream as the reference writes it (no index) is decoded by the parallel segment
    synchronisation pass + the regular chunk decoder; 8 MiB so that thousands of segments take part."""
    n = 8 << 20
    if kind == "zipf":
        data = zipf_bytes(n, 21)
    elif kind == "text":
        data = text_like(n, 22)
    elif kind == "uniform":
        data = np.random.default_rng(23).integers(0, 256, n, dtype=np.uint8).tobytes()
    else:
        data = _skewed(n, 24)
    o = oracle.Model.from_data(data, 1)
    blob, _ = o.compress(data)                      # produced by the (oracle-pinned) reference path
    m = mhc.Model.from_table(o.table_bytes())
    assert m.decompress(blob) == data


@pytest.mark.parametrize("chunk", [256, 1024, 8192])
@pytest.mark.parametrize("n", [1, 255, 256, 257, 8192, 8193, 1 << 20, (1 << 21) + 77])
def test_round_trip_chunk_sizes(mhc, oracle, n, chunk):
    data = zipf_bytes(n, n + chunk)
    m = mhc.Model.from_data(data, 1)
    blob, nbits, idx = m.compress(data, chunk_symbols=chunk)
    assert len(idx) == (n + chunk - 1) // chunk
    assert m.decompress(blob, index=idx, chunk_symbols=chunk, n_symbols=n) == data
    # the GPU stream also decodes on the oracle, and the index points at real code boundaries
    assert oracle.Model.from_data(data, 1).decompress(blob) == data
    assert int(idx[0]) == (0x20 << 56)


def test_index_entries_are_exact(mhc, oracle):
    """Each entry = (previous byte << 56) | bit offset of that chunk's first codeword."""
    data = text_like(40000, 6)
    chunk = 256
    m = mhc.Model.from_data(data, 1)
    _, nbits, idx = m.compress(data, chunk_symbols=chunk)
    lens, _ = oracle.Model.from_data(data, 1).codes()
    arr = np.frombuffer(data, dtype=np.uint8).astype(np.int64)
    prev = np.concatenate([[0x20], arr[:-1]])
    bitpos = np.concatenate([[0], np.cumsum(lens[prev * 256 + arr].astype(np.int64))])
    assert bitpos[-1] == nbits
    for c, e in enumerate(idx):
        assert int(e) & mhc.INDEX_BIT_MASK == bitpos[c * chunk]
        assert int(e) >> 56 == prev[c * chunk]


def test_decode_errors(mhc, oracle):
    data = golden()["input_a.txt"]["data"]
    m = mhc.Model.from_data(data, 1)
    h = mhc.Model.from_data(data, 0)
    blob, _, _ = m.compress(data)
    with pytest.raises(mhc.MhError) as e:
        h.decompress(blob)
    assert e.value.status == mhc.MH_ERR_TYPE
    with pytest.raises(mhc.MhError) as e:
        m.decompress(b"\x10" + blob[1:])
    assert e.value.status == mhc.MH_ERR_CORRUPT
    # a context that does not exist in the table -> null LUT entry -> corrupt
    other = mhc.Model.from_data(b"qqqqqqqq", 1)
    with pytest.raises(mhc.MhError) as e:
        other.decode(b"\xff\xff", 16, prev0=0x41)
    assert e.value.status == mhc.MH_ERR_CORRUPT


def test_large_round_trip_property(mhc, oracle):
    """64 MiB Zipf(1.1): stream identical to the oracle's, round trip exact, and the index lets any
    slice of chunks be decoded on its own."""
    n = 64 << 20
    data = zipf_bytes(n, 2)
    m = mhc.Model.from_data(data, 1)
    blob, nbits, idx = m.compress(data, chunk_symbols=1024)
    ref, ref_bits = oracle.Model.from_data(data, 1).compress(data)
    assert nbits == ref_bits
    assert blob == ref
    assert m.decompress(blob, index=idx, chunk_symbols=1024, n_symbols=n) == data


@pytest.mark.parametrize("s,max_len", [(0.9, 10), (1.1, 11), (1.2, 12)])
def test_decode_l2_layout_runs_of_longest_codes(mhc, oracle, s, max_len):
    """Models whose second-level tables live in L2 (256 Zipf-shaped contexts) and data made almost only of
    the symbols with the LONGEST codes: the hot loop refills its bit window once per two or three symbols,
    counting on code-length bounds, and three 11-bit codes in a row need one bit more than a refill of an
    empty window provides (the round-2 kernel first got that wrong, and only a 16 GiB stream showed it).
    max_len 10 / 11 / 12 select the three-symbol pattern with a single refill, with a double refill, and
    the two-symbol pattern."""
    w = (np.floor((1 << 20) / np.arange(1, 257) ** s) + 1).astype(np.uint64)
    counts = np.tile(w, 256)
    m = mhc.Model.from_counts(counts, 1)
    om = oracle.Model.from_counts(counts, 1)
    lens = np.asarray(om.codes()[0]).reshape(256, 256)
    assert lens.max() == max_len and (lens == lens[0]).all()
    longest = np.flatnonzero(lens[0] == max_len).astype(np.uint8)
    rng = np.random.default_rng(int(s * 100))
    n = (8 << 20) + 123
    data = longest[rng.integers(len(longest), size=n)]
    # other symbols sprinkled in move the code boundaries through every phase of the 32-bit refill
    other = rng.random(n) < 0.07
    data[other] = rng.integers(256, size=int(other.sum()), dtype=np.uint8)
    data = data.tobytes()
    blob, nbits, idx = m.compress(data, chunk_symbols=1024)
    ref, ref_bits = om.compress(data)
    assert nbits == ref_bits and blob == ref
    assert m.decompress(blob, index=idx, chunk_symbols=1024, n_symbols=n) == data
    blob2, nbits2, idx2 = m.compress(data, chunk_symbols=256)
    assert m.decompress(blob2, index=idx2, chunk_symbols=256, n_symbols=n) == data
