"""Round-3 additions to the device path: the histogram's conservation check, encodes that take their own
region-mode histogram, escapes inside the region encoder, the emitted-vs-priced check, and the fine index that the
index builder writes for streams that come without any index (the reference's own files)."""
import os

import numpy as np
import pytest

import __graft_entry__ as entry
from conftest import golden, check_against_golden

pytestmark = pytest.mark.gpu

ENC_REGIONS, ENC_LENGTH_PASS, ENC_REGIONS_ESC = 1, 2, 3


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


def dev_hist(mhc, data, prev0=0x20):
    lib = mhc.lib()
    n = data.size
    d_data = mhc.DeviceBuffer(n + 32, init=np.concatenate([data, np.zeros(32, dtype=np.uint8)]))
    d_counts = mhc.DeviceBuffer(65536 * 8)
    hws = int(lib.mh_dev_histogram_workspace(n))
    d_hws = mhc.DeviceBuffer(hws)
    mhc._check(lib.mh_dev_histogram_o1(d_data.ptr, n, prev0, d_counts.ptr, d_hws.ptr, hws, None), "hist")
    return d_data, d_counts, d_hws, hws


def test_histogram_conservation_check_catches_a_spilled_counter(mhc):
    """64 MiB of zeros: every add of a workgroup goes to ONE 16-bit LDS field.  The product (two guard bits) counts
    them all and the device-side check (sum of counts == n, src/main.cpp:176-178) stays silent; the debug variants
    with one guard bit (round 1's kernel) or none let the field spill into its neighbour, and the check reports it.
    Those variants exist only in the diagnostic library (libmhc_diag.so, -DMH_EXP_PROBES): the shipped one has no switch
    that makes a result wrong."""
    import ctypes as C
    data = np.zeros(64 << 20, dtype=np.uint8)
    lib = mhc.lib()
    d_data, d_counts, d_hws, hws = dev_hist(mhc, data)
    assert lib.mh_dev_status(d_hws.ptr, None) == 0
    counts = d_counts.download(np.uint64)
    assert int(counts.sum()) == data.size and int(counts[0]) == data.size - 1
    os.environ["MH_DEBUG_HIST_GUARD_BITS"] = "0"                 # the shipped library does not look at it
    try:
        d_data, d_counts, d_hws, hws = dev_hist(mhc, data)
        assert lib.mh_dev_status(d_hws.ptr, None) == 0 and int(d_counts.download(np.uint64).sum()) == data.size
    finally:
        del os.environ["MH_DEBUG_HIST_GUARD_BITS"]
    diag = C.CDLL(os.path.join(os.path.dirname(mhc.LIB_PATH), "libmhc_diag.so"))
    diag.mh_dev_histogram_o1.argtypes = [C.c_void_p, C.c_size_t, C.c_uint8, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    diag.mh_dev_status.argtypes = [C.c_void_p, C.c_void_p]
    for bits in ("0", "1"):          # no guard bit: a wrapping field always carries into its neighbour; one: timing decides
        os.environ["MH_DEBUG_HIST_GUARD_BITS"] = bits
        try:
            assert diag.mh_dev_histogram_o1(d_data.ptr, data.size, 0x20, d_counts.ptr, d_hws.ptr, hws, None) == 0
            rc = diag.mh_dev_status(d_hws.ptr, None)
            lost = data.size - int(d_counts.download(np.uint64).sum())
        finally:
            del os.environ["MH_DEBUG_HIST_GUARD_BITS"]
        if bits == "0":
            assert lost != 0, "the debug variant without guard bits did not lose counts: the test no longer exercises the check"
        assert (rc == mhc.MH_ERR_CORRUPT) == (lost != 0), (bits, rc, lost)


def encode_at(mhc, m, data, d_data, chunk=1024, fine=False):
    lib = mhc.lib()
    n = data.size
    cap = lib.mh_encode_bound(m.handle, n) + 64
    wsb = lib.mh_dev_encode_workspace(n)
    d_payload = mhc.DeviceBuffer(cap, init=np.full(cap, 0xEE, dtype=np.uint8))
    d_nbits = mhc.DeviceBuffer(8)
    nidx = max((n + chunk - 1) // chunk, 1)
    d_index = mhc.DeviceBuffer(nidx * 8)
    d_ws = mhc.DeviceBuffer(wsb + 64)
    mhc._check(lib.mh_dev_encode_at(m.handle, d_data.ptr, n, 0x20, None, d_payload.ptr, cap, d_nbits.ptr, d_index.ptr, chunk,
                                    d_ws.ptr, wsb, None), "encode_at")
    mhc._check(lib.mh_dev_status(d_ws.ptr, None), "status")
    nbits = int(d_nbits.download(np.uint64)[0])
    return nbits, d_payload.download()[:(nbits + 7) // 8].tobytes(), lib.mh_dev_encode_path(d_ws.ptr, None)


def test_encode_without_a_histogram_takes_its_own_and_runs_the_region_encoder(mhc, oracle):
    """The reference's `-e table` flow (src/main.cpp:137-161, 208-212): a table from elsewhere, no histogram of this input."""
    data = zipf_bytes((9 << 20) + 13, 5)
    table = oracle.Model.from_data(zipf_bytes(1 << 20, 6).tobytes(), 1)     # every pair of the alphabet occurs in 1 MiB? not all:
    counts = np.maximum(oracle.histogram_o1(zipf_bytes(4 << 20, 6).tobytes()), 1)    # so: a table that has a code for every pair
    om = oracle.Model.from_counts(counts, 1)
    m = mhc.Model.from_table(om.table_bytes())
    d_data = mhc.DeviceBuffer(data.size + 32, init=np.concatenate([data, np.zeros(32, dtype=np.uint8)]))
    nbits, payload, path = encode_at(mhc, m, data, d_data)
    ref, ref_bits = om.compress(data.tobytes())
    assert (nbits, payload) == (ref_bits, ref[1:])
    assert path in (ENC_REGIONS, ENC_REGIONS_ESC), path
    os.environ["MH_ENCODE_LENGTH_PASS"] = "1"                    # the pair it replaces, for comparison
    try:
        nbits2, payload2, path2 = encode_at(mhc, m, data, d_data)
    finally:
        del os.environ["MH_ENCODE_LENGTH_PASS"]
    assert path2 == ENC_LENGTH_PASS and (nbits2, payload2) == (nbits, payload)
    small = data[:100000].copy()
    d_small = mhc.DeviceBuffer(small.size + 32, init=np.concatenate([small, np.zeros(32, dtype=np.uint8)]))
    nb, pl, pth = encode_at(mhc, m, small, d_small)
    assert pth == ENC_LENGTH_PASS and pl == om.compress(small.tobytes())[0][1:]


@pytest.mark.parametrize("name", ["input_wiki_cpp.txt", "input_wiki_cpp.html", "kat4"])
def test_golden_inputs_with_codes_over_12_bits_run_the_region_encoder(mhc, name):
    """The reference's own wiki inputs have 13- and 15-bit Markov codes, kat4 15-bit ones: the region
    encoder takes them with its escape variant, and the stream is the reference's (.cm golden, header byte included)."""
    lib = mhc.lib()
    data = np.frombuffer(golden()[name]["data"], dtype=np.uint8)
    n = data.size
    d_data, d_counts, d_hws, hws = dev_hist(mhc, data)
    m = mhc.Model.from_device_counts(d_counts.ptr, 1)
    assert m.max_code_len > 12
    cap = lib.mh_encode_bound(m.handle, n) + 64
    wsb = lib.mh_dev_encode_workspace(n)
    d_payload = mhc.DeviceBuffer(cap)
    d_nbits = mhc.DeviceBuffer(8)
    d_ws = mhc.DeviceBuffer(wsb + 64)
    mhc._check(lib.mh_dev_encode_hist(m.handle, d_data.ptr, n, 0x20, None, d_payload.ptr, cap, d_nbits.ptr, None, 0,
                                      d_hws.ptr, hws, d_ws.ptr, wsb, None), "encode_hist")
    mhc._check(lib.mh_dev_status(d_ws.ptr, None), "status")
    assert lib.mh_dev_encode_path(d_ws.ptr, None) == ENC_REGIONS_ESC
    nbits = int(d_nbits.download(np.uint64)[0])
    blob = bytes([lib.mh_stream_header(m.handle, nbits)]) + d_payload.download()[:(nbits + 7) // 8].tobytes()
    check_against_golden(name, "cm", blob)


def test_many_long_codes_round_by_round(mhc, oracle):
    """A model whose codes are mostly far over 12 bits (Fibonacci weights): rounds exceed the 12-bits-per-symbol
    image and go through piece by piece; same stream as the oracle's."""
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    row = np.zeros(256, dtype=np.uint64)
    row[:40] = np.array(fib, dtype=np.uint64)
    counts = np.tile(row, 256)
    om = oracle.Model.from_counts(counts, 1)
    rng = np.random.default_rng(8)
    data = rng.integers(0, 12, (1 << 20) + 77, dtype=np.uint8)       # the rare symbols: codes of 28..39 bits, 16 per lane
    lib = mhc.lib()
    d_data, d_counts, d_hws, hws = dev_hist(mhc, data)
    m = mhc.Model.from_table(om.table_bytes())
    cap = lib.mh_encode_bound(m.handle, data.size) + 64
    wsb = lib.mh_dev_encode_workspace(data.size)
    d_payload = mhc.DeviceBuffer(cap)
    d_nbits = mhc.DeviceBuffer(8)
    d_ws = mhc.DeviceBuffer(wsb + 64)
    mhc._check(lib.mh_dev_encode_hist(m.handle, d_data.ptr, data.size, 0x20, None, d_payload.ptr, cap, d_nbits.ptr, None, 0,
                                      d_hws.ptr, hws, d_ws.ptr, wsb, None), "encode_hist")
    mhc._check(lib.mh_dev_status(d_ws.ptr, None), "status")
    nbits = int(d_nbits.download(np.uint64)[0])
    ref, ref_bits = om.compress(data.tobytes())
    assert nbits == ref_bits and nbits > 12 * 16 * data.size // 16
    assert d_payload.download()[:(nbits + 7) // 8].tobytes() == ref[1:]


def test_buffer_refilled_between_histogram_and_encode_is_reported(mhc):
    """Same pointer, same length, other contents: the header still matches; every region compares what it emitted with
    what it was priced at and the status says MH_ERR_CORRUPT; nothing is stored beyond the capacity."""
    lib = mhc.lib()
    n = 8 << 20
    a = zipf_bytes(n, 1)
    b = np.random.default_rng(2).integers(0, 256, n, dtype=np.uint8)           # incompressible: far more bits than priced
    d_data, d_counts, d_hws, hws = dev_hist(mhc, a)
    m = mhc.Model.from_device_counts(d_counts.ptr, 1)
    mhc._check(lib.mh_dev_upload(d_data.ptr, b.ctypes.data, n), "refill")
    cap = ((int(np.frombuffer(m.image(1), dtype=np.uint8).astype(np.int64)[0]) * 0 + n) // 4) * 4      # tight: the priced size is ~0.73 n
    guard = 4096
    d_payload = mhc.DeviceBuffer(cap + guard, init=np.full(cap + guard, 0x5A, dtype=np.uint8))
    d_nbits = mhc.DeviceBuffer(8)
    wsb = lib.mh_dev_encode_workspace(n)
    d_ws = mhc.DeviceBuffer(wsb + 64)
    mhc._check(lib.mh_dev_encode_hist(m.handle, d_data.ptr, n, 0x20, None, d_payload.ptr, cap, d_nbits.ptr, None, 0,
                                      d_hws.ptr, hws, d_ws.ptr, wsb, None), "encode_hist")
    assert lib.mh_dev_status(d_ws.ptr, None) == mhc.MH_ERR_CORRUPT
    assert np.all(d_payload.download()[cap:] == 0x5A), "stored beyond the capacity"


def test_stream_without_an_index_decodes_through_the_tile_decoder(mhc, oracle):
    """What the reference writes has no index: the index builder's fill pass writes chunk index AND fine index, and
    the host-buffer decode then runs the tile decoder (forced here; automatic from 8 MiB per segment on)."""
    data = zipf_bytes((12 << 20) + 1001, 17)
    om = oracle.Model.from_data(data.tobytes(), 1)
    blob, nbits = om.compress(data.tobytes())
    m = mhc.Model.from_table(om.table_bytes())
    os.environ["MH_DECODE_PATH"] = "tile"
    os.environ["MH_DECODE_NO_STREAM"] = "1"                 # (round 5: by default such a stream is decoded without any index,
    try:                                                    #  tests/test_gpu_stream.py; this test is about the index + tile decoder way)
        out = m.decompress(blob)
    finally:
        del os.environ["MH_DECODE_PATH"]
        del os.environ["MH_DECODE_NO_STREAM"]
    assert mhc.lib().mh_last_index_path() == 5              # the index builder's fast path (tiles of 352-bit segments)
    assert out == data.tobytes()
    # device level: the fine index the builder wrote equals the one the encoder writes
    lib = mhc.lib()
    pl = np.frombuffer(blob[1:], dtype=np.uint8)
    d_pl = mhc.DeviceBuffer(pl.size + 64, init=np.concatenate([pl, np.zeros(64, dtype=np.uint8)]))
    icap = nbits // 1024 + 2
    fcap = nbits // 64 + 2
    d_idx, d_fine, d_ns = mhc.DeviceBuffer(icap * 8), mhc.DeviceBuffer(fcap * 4), mhc.DeviceBuffer(8)
    iws = int(lib.mh_dev_build_index_workspace(nbits))
    d_iws = mhc.DeviceBuffer(iws)
    mhc._check(lib.mh_dev_build_index_fine(m.handle, d_pl.ptr, nbits, 0x20, d_idx.ptr, icap, 1024, d_fine.ptr, fcap, d_ns.ptr,
                                           d_iws.ptr, iws, None), "build_index_fine")
    mhc._check(lib.mh_dev_status(d_iws.ptr, None), "status")
    assert int(d_ns.download(np.uint64)[0]) == data.size
    lens = np.frombuffer(m.image(1), dtype=np.uint8).astype(np.int64)
    prev = np.concatenate([[0x20], data[:-1]]).astype(np.int64)
    pos = np.concatenate([[0], np.cumsum(lens[prev * 256 + data.astype(np.int64)])[:-1]])
    j = np.arange(0, data.size, 64)
    want = ((prev[j] << 24) | (pos[j] & 0xFFFFFF)).astype(np.uint32)
    assert np.array_equal(d_fine.download(np.uint32)[:want.size], want)


@pytest.mark.parametrize("kind,forced,want", [("zipf", None, 1), ("uniform", None, 2), ("zipf", "chunk", 2), ("text", "tile", 1)])
def test_decode_path_word_says_which_decoder_ran(mhc, kind, forced, want):
    """mh_dev_decode_path: the library picks the tile decoder from 8 MiB on when the average code is shorter than the tile
    tables' first level (uniform bytes: 8-bit codes, the chunk decoder), MH_DECODE_PATH overrides."""
    import torch
    import bench
    bench.CHUNK = 256
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = 16 << 20
    data = bench.generate(kind, n, {"zipf": 2, "uniform": 3, "text": 1}[kind], 0, dev)
    codec = bench.Codec(mhc, n, dev)
    codec.histogram(data, 0x20)
    model = codec.build_model()
    codec.encode(model, data, 0x20)
    codec.nbits_hint = int(codec.nbits[0].item())            # (the choice by code length needs the payload length on the host)
    if forced:
        os.environ["MH_DECODE_PATH"] = forced
    try:
        codec.decode(model)
        torch.cuda.synchronize()
    finally:
        if forced:
            del os.environ["MH_DECODE_PATH"]
    assert codec.lib.mh_dev_status(codec.dec_ws.data_ptr(), codec.stream()) == 0
    assert codec.lib.mh_dev_decode_path(codec.dec_ws.data_ptr(), codec.stream()) == want
    assert torch.equal(codec.decoded, data)
