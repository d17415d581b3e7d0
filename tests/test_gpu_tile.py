"""The tile decoder (mh_tile.hip, mh_dev_decode_fine) and the device-only fine index that feeds it.

A wave decodes 64 adjacent 64-symbol pieces from one contiguous piece of the payload staged in LDS; what must
hold: the decoded bytes equal the input (and therefore what the reference's decompress, src/coding.cpp:118-157,
produces for the same stream, which test_gpu_parity pins through the oracle), for every first-level width, for
both encoders that write the fine index, for ragged ends, pre-shifted shards, long codes (redo pass) and damaged
indexes (reported, never a wild access)."""
import os

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


@pytest.fixture(autouse=True)
def force_tile_path():
    old = {k: os.environ.get(k) for k in ("MH_DECODE_PATH", "MH_TILE_P")}
    os.environ["MH_DECODE_PATH"] = "tile"
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


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


class Stream:
    """Device-resident encode of `data` with a fine index; decode() runs mh_dev_decode_fine."""

    def __init__(self, mhc, data, chunk=1024, start_bit=0, prev0=0x20, use_hist=True, device_model=True, model=None):
        self.mhc, self.lib, self.data, self.chunk = mhc, mhc.lib(), data, chunk
        lib, n = self.lib, data.size
        self.n = n
        self.d_data = mhc.DeviceBuffer(n + 32, init=np.concatenate([data, np.zeros(32, dtype=np.uint8)]))
        self.d_counts = mhc.DeviceBuffer(65536 * 8)
        hws = int(lib.mh_dev_histogram_workspace(n))
        self.d_hws = mhc.DeviceBuffer(hws)
        mhc._check(lib.mh_dev_histogram_o1(self.d_data.ptr, n, prev0, self.d_counts.ptr, self.d_hws.ptr, hws, None), "hist")
        if model is not None:
            self.model = model
        elif device_model:
            self.model = mhc.Model.from_device_counts(self.d_counts.ptr, 1)
        else:
            self.model = mhc.Model.from_counts(self.d_counts.download(np.uint64), 1)
        m = self.model
        self.cap = lib.mh_encode_bound(m.handle, n) + 64
        self.nidx = max((n + chunk - 1) // chunk, 1)
        self.nfine = max((n + 63) // 64, 1)
        wsb = lib.mh_dev_encode_workspace(n)
        self.d_payload = mhc.DeviceBuffer(self.cap, init=np.full(self.cap, 0xEE, dtype=np.uint8))
        self.d_nbits = mhc.DeviceBuffer(8, init=np.zeros(1, dtype=np.uint64))
        self.d_index = mhc.DeviceBuffer(self.nidx * 8, init=np.zeros(self.nidx, dtype=np.uint64))
        self.d_fine = mhc.DeviceBuffer(self.nfine * 4, init=np.full(self.nfine, 0xDDDDDDDD, dtype=np.uint32))
        d_ws = mhc.DeviceBuffer(wsb + 64)
        d_start = mhc.DeviceBuffer(8, init=np.array([start_bit], dtype=np.uint64))
        mhc._check(lib.mh_dev_encode_fine(m.handle, self.d_data.ptr, n, prev0, d_start.ptr, self.d_payload.ptr, self.cap, self.d_nbits.ptr,
                                          self.d_index.ptr, chunk, self.d_fine.ptr,
                                          self.d_hws.ptr if use_hist else None, hws if use_hist else 0, d_ws.ptr, wsb, None), "encode_fine")
        mhc._check(lib.mh_dev_status(d_ws.ptr, None), "encode status")
        self.nbits = int(self.d_nbits.download(np.uint64)[0])

    def decode(self, fine=True, dn=False):
        lib, mhc, n = self.lib, self.mhc, self.n
        d_out = mhc.DeviceBuffer(n + 64, init=np.full(n + 64, 0xAB, dtype=np.uint8))
        dws = int(lib.mh_dev_decode_workspace(self.nbits, n, self.chunk))
        d_ws = mhc.DeviceBuffer(dws)
        mhc._check(lib.mh_dev_decode_fine(self.model.handle, self.d_payload.ptr, 0 if dn else self.nbits, self.d_nbits.ptr if dn else None,
                                          d_out.ptr, n, self.d_index.ptr, self.chunk, self.d_fine.ptr if fine else None,
                                          d_ws.ptr, dws, None), "decode_fine")
        rc = lib.mh_dev_status(d_ws.ptr, None)
        out = d_out.download()
        assert np.all(out[n:] == 0xAB), "wrote past the end of the output"
        return rc, out[:n]


def fine_reference(data, index, nbits_unused, model, mhc, prev0=0x20):
    """Fine index recomputed on the host from the code lengths: entry j = ctx << 24 | (bit offset & 0xFFFFFF)."""
    lens = np.frombuffer(model.image(1), dtype=np.uint8).astype(np.int64)
    prev = np.concatenate([[prev0], data[:-1]]).astype(np.int64)
    l = lens[prev * 256 + data.astype(np.int64)]
    pos = np.concatenate([[0], np.cumsum(l)[:-1]])
    j = np.arange(0, data.size, 64)
    return ((prev[j] << 24) | (pos[j] & 0xFFFFFF)).astype(np.uint32)


@pytest.mark.parametrize("tile_p", [5, 6, 7, 8])
@pytest.mark.parametrize("n", [8192, 8192 * 3 + 1, (1 << 20) + 77, (5 << 20) + 4099])
def test_tile_decode_round_trip(mhc, tile_p, n):
    os.environ["MH_TILE_P"] = str(tile_p)
    data = zipf_bytes(n, n + tile_p)
    s = Stream(mhc, data)
    assert len(s.model.image(8)) == (256 << tile_p) * 2
    fine = s.d_fine.download(np.uint32)[:(n + 63) // 64]
    assert np.array_equal(fine, fine_reference(data, None, None, s.model, mhc))
    rc, out = s.decode()
    assert rc == 0
    assert np.array_equal(out, data)
    rc, out2 = s.decode(fine=False)                 # the chunk decoder on the same stream
    assert rc == 0 and np.array_equal(out2, data)


@pytest.mark.parametrize("kind", ["text", "uniform", "zeros", "ab", "zipf16"])
def test_tile_decode_other_sources(mhc, kind):
    n = (6 << 20) + 11
    if kind == "text":
        data = text_like(n, 3)
    elif kind == "uniform":
        data = np.random.default_rng(4).integers(0, 256, n, dtype=np.uint8)
    elif kind == "zeros":
        data = np.zeros(n, dtype=np.uint8)
    elif kind == "ab":
        data = np.tile(np.frombuffer(b"ab", dtype=np.uint8), n // 2 + 1)[:n].copy()
    else:
        data = zipf_bytes(n, 5, k=16)
    for use_hist in (True, False):
        s = Stream(mhc, data, use_hist=use_hist)
        rc, out = s.decode(dn=True)
        assert rc == 0 and np.array_equal(out, data), (kind, use_hist)


@pytest.mark.parametrize("chunk", [256, 512, 2048, 4096])
def test_tile_decode_chunk_sizes(mhc, chunk):
    data = zipf_bytes((3 << 20) + 333, chunk)
    s = Stream(mhc, data, chunk=chunk)
    rc, out = s.decode()
    assert rc == 0 and np.array_equal(out, data)


def test_tile_tables_host_build_equals_device_build(mhc):
    data = zipf_bytes(1 << 20, 21)
    for tile_p in (5, 6, 7, 8):
        os.environ["MH_TILE_P"] = str(tile_p)
        a = Stream(mhc, data, device_model=True)
        b = Stream(mhc, data, device_model=False)
        assert a.model.image(8) == b.model.image(8), tile_p
        assert a.model.image(9) == b.model.image(9), tile_p
        rc, out = b.decode()
        assert rc == 0 and np.array_equal(out, data)


def test_tile_decode_pre_shifted_shard_and_other_context(mhc):
    data = zipf_bytes((2 << 20) + 9, 9)
    s = Stream(mhc, data, start_bit=(1 << 40) + 5, prev0=0x41)
    rc, out = s.decode()
    assert rc == 0 and np.array_equal(out, data)


def test_tile_decode_long_codes_take_the_redo_pass(mhc):
    """Codes longer than P + H bits (here up to 15+ with P = 5: H = 8 covers 13) go to the chunk decoder's walk."""
    os.environ["MH_TILE_P"] = "5"
    rng = np.random.default_rng(12)
    x = rng.integers(0, 256, 3 << 20, dtype=np.uint8) & rng.integers(0, 256, 3 << 20, dtype=np.uint8)
    s = Stream(mhc, x)
    assert s.model.max_code_len > 13
    rc, out = s.decode()
    assert rc == 0 and np.array_equal(out, x)


def test_tile_decode_non_stationary_stream(mhc):
    """Half incompressible, half constant: the largest staged piece is far above the average one."""
    n = 4 << 20
    data = np.concatenate([np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8), np.full(n, 7, dtype=np.uint8),
                           zipf_bytes(n + 5, 3)])
    s = Stream(mhc, data)
    rc, out = s.decode()
    assert rc == 0 and np.array_equal(out, data)


def test_tile_decode_reports_a_damaged_fine_index(mhc):
    data = zipf_bytes(2 << 20, 33)
    s = Stream(mhc, data)
    fine = s.d_fine.download(np.uint32)
    for where, value in ((1000, fine[1000] ^ 0x155), (5, 0xFFFFFFFF), (20000, fine[20000] ^ 0x01000000)):
        bad = fine.copy()
        bad[where] = value
        mhc._check(s.lib.mh_dev_upload(s.d_fine.ptr, bad.ctypes.data, bad.nbytes), "upload")
        rc, out = s.decode()
        assert rc == mhc.MH_ERR_CORRUPT, (where, rc)
    mhc._check(s.lib.mh_dev_upload(s.d_fine.ptr, fine.ctypes.data, fine.nbytes), "upload")
    rc, out = s.decode()
    assert rc == 0 and np.array_equal(out, data)


# ------------------------------------------------------------------ order 2 (extension: parity unpinned)

def o2_round_trip(mhc, oracle, data, chunk=1024, expect_tiles=None):
    """Device path of an order-2 model: histogram, device tree build (the live contexts get LDS tables), encode with
    fine index, decode with it (tile decoder when every live context has a slot, the chunk decoder otherwise)."""
    lib = mhc.lib()
    n = data.size
    d_data = mhc.DeviceBuffer(n + 32, init=np.concatenate([data, np.zeros(32, dtype=np.uint8)]))
    d_counts = mhc.DeviceBuffer((1 << 24) * 8)
    mhc._check(lib.mh_dev_histogram_o2(d_data.ptr, n, 0x2020, d_counts.ptr, None), "hist2")
    m = mhc.Model.from_device_counts(d_counts.ptr, 2)
    cap = lib.mh_encode_bound(m.handle, n) + 64
    nidx, nfine = max((n + chunk - 1) // chunk, 1), max((n + 63) // 64, 1)
    wsb = lib.mh_dev_encode_workspace(n)
    d_payload = mhc.DeviceBuffer(cap, init=np.full(cap, 0xEE, dtype=np.uint8))
    d_nbits, d_index, d_fine = mhc.DeviceBuffer(8), mhc.DeviceBuffer(nidx * 8), mhc.DeviceBuffer(nfine * 4)
    d_ws = mhc.DeviceBuffer(wsb + 64)
    mhc._check(lib.mh_dev_encode_ctx_fine(m.handle, d_data.ptr, n, 0x2020, None, d_payload.ptr, cap, d_nbits.ptr, d_index.ptr, chunk,
                                          d_fine.ptr, d_ws.ptr, wsb, None), "encode")
    mhc._check(lib.mh_dev_status(d_ws.ptr, None), "encode status")
    nbits = int(d_nbits.download(np.uint64)[0])
    ref, ref_bits = oracle.Model.from_data(data.tobytes(), 2).compress(data.tobytes())
    assert nbits == ref_bits and d_payload.download()[:(nbits + 7) // 8].tobytes() == ref[1:]
    d_out = mhc.DeviceBuffer(n + 64, init=np.full(n + 64, 0xAB, dtype=np.uint8))
    dws = int(lib.mh_dev_decode_workspace(nbits, n, chunk))
    d_dws = mhc.DeviceBuffer(dws)
    mhc._check(lib.mh_dev_decode_fine(m.handle, d_payload.ptr, nbits, None, d_out.ptr, n, d_index.ptr, chunk, d_fine.ptr,
                                      d_dws.ptr, dws, None), "decode")
    assert lib.mh_dev_status(d_dws.ptr, None) == 0
    out = d_out.download()
    assert np.all(out[n:] == 0xAB) and np.array_equal(out[:n], data)
    if expect_tiles is not None:                 # which decoder ran is on record in the workspace (1 = the tile decoder)
        path = lib.mh_dev_decode_path(d_dws.ptr, None)
        assert (path == 1) == expect_tiles, path


@pytest.mark.parametrize("chunk", [256, 1024])
def test_order2_tile_decode_text_parity_unpinned(mhc, oracle, chunk):
    data = text_like((3 << 20) + 1234, 11)
    o2_round_trip(mhc, oracle, data, chunk=chunk, expect_tiles=True)


def test_order2_many_contexts_takes_the_chunk_decoder_parity_unpinned(mhc, oracle):
    data = zipf_bytes((1 << 20) + 99, 12)        # tens of thousands of live contexts: no slots for all of them
    o2_round_trip(mhc, oracle, data, expect_tiles=False)


def test_order2_tile_decode_long_codes_parity_unpinned(mhc, oracle):
    """Rare successors get codes beyond the 6 + H bits the tile tables resolve: those pieces go to the chunk decoder."""
    rng = np.random.default_rng(4)
    base = text_like(4 << 20, 5).copy()
    for k in range(26):                          # after "t " (sit, amet, elit), very rarely, one of 26 capital letters
        pos = np.flatnonzero((base[:-3] == ord("t")) & (base[1:-2] == ord(" ")))
        pick = pos[rng.integers(0, pos.size, 1 + k)]
        base[pick + 2] = ord("A") + k
    o2_round_trip(mhc, oracle, base)
