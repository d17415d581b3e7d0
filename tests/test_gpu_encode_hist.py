"""mh_dev_encode_hist: the compress path's encoder that prices its regions from the histogram workspace
(no length pass, input read once).  It must write exactly what mh_dev_encode_at writes — payload bits, payload
length, chunk index — and therefore exactly the reference's stream (oracle)."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as entry

pytestmark = pytest.mark.gpu


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
    words = [b"lorem", b"ipsum", b"dolor", b"sit", b"amet", b"consectetur", b"adipiscing", b"elit", b"sed", b"do"]
    out = bytearray()
    while len(out) < n:
        out += words[int(rng.integers(len(words)))] + (b". " if rng.random() < 0.1 else b" ")
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8)


def run_both(mhc, data, chunk=256, start_bit=0, prev0=0x20, model=None):
    """Encodes `data` (uint8 array) with mh_dev_encode_hist and mh_dev_encode_at; returns both results."""
    lib = mhc.lib()
    n = data.size
    d_data = mhc.DeviceBuffer(n + 32, init=np.concatenate([data, np.zeros(32, dtype=np.uint8)]))
    d_counts = mhc.DeviceBuffer(65536 * 8)
    hws = int(lib.mh_dev_histogram_workspace(n))
    d_hws = mhc.DeviceBuffer(hws)
    mhc._check(lib.mh_dev_histogram_o1(d_data.ptr, n, prev0, d_counts.ptr, d_hws.ptr, hws, None), "hist")
    counts = d_counts.download(np.uint64)
    m = model or mhc.Model.from_counts(counts, 1)
    cap = lib.mh_encode_bound(m.handle, n) + 64
    nidx = max((n + chunk - 1) // chunk, 1)
    wsb = lib.mh_dev_encode_workspace(n)
    d_start = mhc.DeviceBuffer(8, init=np.array([start_bit], dtype=np.uint64))
    out = []
    for which in ("hist", "at"):
        d_payload = mhc.DeviceBuffer(cap, init=np.full(cap, 0xEE, dtype=np.uint8))
        d_nbits = mhc.DeviceBuffer(8, init=np.zeros(1, dtype=np.uint64))
        d_index = mhc.DeviceBuffer(nidx * 8, init=np.zeros(nidx, dtype=np.uint64))
        d_ws = mhc.DeviceBuffer(wsb + 64)
        if which == "hist":
            rc = lib.mh_dev_encode_hist(m.handle, d_data.ptr, n, prev0, d_start.ptr, d_payload.ptr, cap, d_nbits.ptr, d_index.ptr, chunk,
                                        d_hws.ptr, hws, d_ws.ptr, wsb, None)
        else:
            rc = lib.mh_dev_encode_at(m.handle, d_data.ptr, n, prev0, d_start.ptr, d_payload.ptr, cap, d_nbits.ptr, d_index.ptr, chunk,
                                      d_ws.ptr, wsb, None)
        mhc._check(rc, which)
        mhc._check(lib.mh_dev_status(d_ws.ptr, None), which + " status")
        nbits = int(d_nbits.download(np.uint64)[0])
        out.append((nbits, d_payload.download()[:(nbits + 7) // 8].tobytes(), d_index.download(np.uint64)[:(n + chunk - 1) // chunk]))
    return m, counts, out[0], out[1]


@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 16383, 16384, 16385, 100000, (1 << 20) + 3, (4 << 20) + 17, (64 << 20) + 5])
def test_encode_hist_equals_two_pass_and_oracle(mhc, oracle, n):
    data = zipf_bytes(n, n)
    m, counts, a, b = run_both(mhc, data)
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2], b[2])
    ref, ref_bits = oracle.Model.from_counts(counts, 1).compress(data.tobytes())
    assert a[0] == ref_bits and a[1] == ref[1:]


@pytest.mark.parametrize("kind", ["text", "uniform", "zeros", "ab", "zipf16"])
def test_encode_hist_on_other_sources(mhc, oracle, kind):
    n = (24 << 20) + 11
    if kind == "text":
        data = text_like(n, 3)
    elif kind == "uniform":
        data = np.random.default_rng(4).integers(0, 256, n, dtype=np.uint8)       # 8-bit codes: the round image is full
    elif kind == "zeros":
        data = np.zeros(n, dtype=np.uint8)                                        # every region's counts live in its crossing list
    elif kind == "ab":
        data = np.tile(np.frombuffer(b"ab", dtype=np.uint8), n // 2 + 1)[:n].copy()
    else:
        data = zipf_bytes(n, 5, k=16)
    m, counts, a, b = run_both(mhc, data, chunk=1024)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
    ref, ref_bits = oracle.Model.from_counts(counts, 1).compress(data.tobytes())
    assert a[0] == ref_bits and a[1] == ref[1:]


@pytest.mark.parametrize("start_bit", [3, 37, (1 << 40) + 5])
def test_encode_hist_pre_shifted_shard(mhc, start_bit):
    """A shard or segment emitted at a bit offset (only start_bit % 8 matters): same bytes as mh_dev_encode_at."""
    data = zipf_bytes((2 << 20) + 9, 9)
    m, counts, a, b = run_both(mhc, data, start_bit=start_bit, prev0=0x41)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
    assert a[0] % 8 == (start_bit % 8 + (b[0] - start_bit % 8)) % 8


def test_encode_hist_escape_model_falls_back(mhc, oracle):
    """A model with codes over 12 bits takes the three-kernel path by itself."""
    rng = np.random.default_rng(12)
    x = rng.integers(0, 256, 1 << 20, dtype=np.uint8) & rng.integers(0, 256, 1 << 20, dtype=np.uint8)
    m, counts, a, b = run_both(mhc, x)
    assert m.max_code_len > 12
    assert a[0] == b[0] and a[1] == b[1]
    assert a[1] == oracle.Model.from_counts(counts, 1).compress(x.tobytes())[0][1:]


def test_encode_hist_rejects_a_foreign_workspace(mhc):
    """The workspace must hold the histogram of the buffer that is being encoded."""
    lib = mhc.lib()
    n = 1 << 20
    a = zipf_bytes(n, 1)
    b = zipf_bytes(n, 2)
    d_a = mhc.DeviceBuffer(n + 32, init=np.concatenate([a, np.zeros(32, dtype=np.uint8)]))
    d_b = mhc.DeviceBuffer(n + 32, init=np.concatenate([b, np.zeros(32, dtype=np.uint8)]))
    d_counts = mhc.DeviceBuffer(65536 * 8)
    hws = int(lib.mh_dev_histogram_workspace(n))
    d_hws = mhc.DeviceBuffer(hws)
    mhc._check(lib.mh_dev_histogram_o1(d_a.ptr, n, 0x20, d_counts.ptr, d_hws.ptr, hws, None), "hist")
    m = mhc.Model.from_counts(np.full(65536, 7, dtype=np.uint64), 1)      # every pair has an 8-bit code: no escape, no fallback
    assert m.max_code_len == 8
    cap = lib.mh_encode_bound(m.handle, n) + 64
    d_payload = mhc.DeviceBuffer(cap)
    d_nbits = mhc.DeviceBuffer(8)
    wsb = lib.mh_dev_encode_workspace(n)
    d_ws = mhc.DeviceBuffer(wsb + 64)
    mhc._check(lib.mh_dev_encode_hist(m.handle, d_b.ptr, n, 0x20, None, d_payload.ptr, cap, d_nbits.ptr, None, 0,
                                      d_hws.ptr, hws, d_ws.ptr, wsb, None), "encode_hist")
    assert lib.mh_dev_status(d_ws.ptr, None) == mhc.MH_ERR_CORRUPT


@pytest.mark.parametrize("block", [8 << 10, 16 << 10, 48 << 10, 1 << 20])
def test_region_encoder_rounds_that_fit_half_the_image_next_to_rounds_that_do_not(mhc, oracle, block):
    """[r4] The region encoder alternates between the two halves of its LDS image (one barrier per 16 KiB round) for rounds
    whose bits fit a half, and takes the whole image (two barriers) for a round that does not — and for the round BEHIND such
    a round, whose predecessor's flush of the whole image may still be running.  Blocks of random bytes (8+ bits per symbol
    under a model that also holds a cheap symbol: a round of them does not fit) alternate with blocks of that cheap symbol
    (1-2 bits: fits), with block lengths under, at and over the round length and far over it: every transition
    half -> whole -> half, runs of whole rounds and runs of half rounds, all in one stream; the bytes are the oracle's."""
    n = (12 << 20) + 345
    rng = np.random.default_rng(block)
    data = rng.integers(0, 256, n, dtype=np.uint8)
    for lo in range(block, n, 2 * block):                        # every second block: one cheap symbol
        data[lo:lo + block] = 0x41
    m, counts, a, b = run_both(mhc, data, chunk=1024)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
    ref, ref_bits = oracle.Model.from_counts(counts, 1).compress(data.tobytes())
    assert a[0] == ref_bits and a[1] == ref[1:]
    lens = np.asarray(oracle.Model.from_counts(counts, 1).codes()[0]).reshape(256, 256)
    assert lens[0x41, 0x41] <= 2 and np.median(lens[lens > 0]) >= 8    # the two kinds of round really differ


@pytest.mark.parametrize("kind,chunk", [("uniform", 256), ("uniform", 8192), ("near8", 1024), ("mixed_regions", 1024), ("ragged_uniform", 512)])
def test_region_encoder_rounds_of_fewer_than_1024_lanes(mhc, oracle, kind, chunk):
    """[r5] A region whose mean 16 KiB round does not fit half the LDS image (8-bit codes: 131 072 bits against 130 304) runs its
    rounds with the first 1008, 992 or 960 lanes, so that they alternate between the halves with one barrier each (BASELINE
    config 4: uniform bytes).  Chunk index entries then fall on other lanes in every round and idle lanes must write none; regions
    of the two kinds sit side by side in one stream (every workgroup chooses for its own region); the stream's ragged end goes
    through the bounds-checked rounds.  The bytes are the oracle's and the length-pass encoder's; the fine index (every 64
    symbols) is checked through the decoder in tests/test_gpu_tile.py's round trips and bench.py's."""
    rng = np.random.default_rng(len(kind) + chunk)
    n = (20 << 20) + 333
    if kind == "uniform":
        data = rng.integers(0, 256, n, dtype=np.uint8)                            # every code 8 bits: 1008 lanes per round
    elif kind == "near8":                                                         # 8- and 9-bit codes, a few 7s: about 8.2 bits per symbol -> 992 / 960 lanes
        w = np.concatenate([np.full(192, 1.0), np.full(64, 0.55)])
        data = rng.choice(256, size=n, p=w / w.sum()).astype(np.uint8)
    elif kind == "mixed_regions":                                                 # uniform bytes and text by turns, 3 MiB each: both geometries in one launch
        data = rng.integers(0, 256, n, dtype=np.uint8)
        t = text_like(n, 9)
        for lo in range(0, n, 6 << 20):
            data[lo:lo + (3 << 20)] = t[lo:lo + (3 << 20)]
    else:
        n = (5 << 20) + 16 * 1008 * 3 + 7                                          # ends three short rounds and seven bytes into a region
        data = rng.integers(0, 256, n, dtype=np.uint8)
    m, counts, a, b = run_both(mhc, data, chunk=chunk)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
    ref, ref_bits = oracle.Model.from_counts(counts, 1).compress(data.tobytes())
    assert a[0] == ref_bits and a[1] == ref[1:]
