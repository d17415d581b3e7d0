"""The one-pass order-2 encoder (enc_chain_kernel: every wave-tile looked up once, start bits by a chained scan over
the tiles, no dword with two writers) against the length pass + emit pair it replaces and against the oracle.

PARITY UNPINNED like everything order 2 (the reference has order 1 only, README.md:158-166): the spec is the
generalised oracle.  What these tests add is that BOTH encoders produce that stream, for sizes on every side of the
4 KiB wave-tile and the 64-tile ticket, with a start offset, with escapes, and that a buffer too small is reported."""
import os

import numpy as np
import pytest

import __graft_entry__ as entry

pytestmark = pytest.mark.gpu

ENC_LENGTH_PASS, ENC_CHAIN = 2, 4


@pytest.fixture(scope="module")
def mhc():
    mod = entry.load_package()
    mod.lib()
    assert mod.device_count() >= 1
    return mod


def text_like(n, seed):
    rng = np.random.default_rng(seed)
    words = [b"lorem", b"ipsum", b"dolor", b"sit", b"amet", b"consectetur", b"adipiscing", b"elit", b"sed", b"do",
             b"eiusmod", b"tempor", b"incididunt", b"ut", b"labore", b"et", b"dolore", b"magna", b"aliqua"]
    out = bytearray()
    while len(out) < n:
        out += words[int(rng.integers(len(words)))] + (b". " if rng.random() < 0.1 else b" ")
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()


def encode(mhc, m, data, start_bit=None, chunk=256, cap=None, fine=True, ctx0=0x2020):
    """-> (rc of the status word, nbits, payload bytes, index, fine, path)"""
    lib = mhc.lib()
    n = data.size
    full = lib.mh_encode_bound(m.handle, n) + 64
    cap = full if cap is None else cap
    wsb = lib.mh_dev_encode_workspace(n)
    d_data = mhc.DeviceBuffer(n + 64, init=np.concatenate([data, np.zeros(64, dtype=np.uint8)]))
    d_payload = mhc.DeviceBuffer(full, init=np.full(full, 0xEE, dtype=np.uint8))
    d_nbits = mhc.DeviceBuffer(8)
    nidx = max((n + chunk - 1) // chunk, 1)
    d_index = mhc.DeviceBuffer(nidx * 8)
    nfine = max((n + 63) // 64, 1)
    d_fine = mhc.DeviceBuffer(nfine * 4)
    d_ws = mhc.DeviceBuffer(wsb + 64)
    d_start = None
    if start_bit is not None:
        d_start = mhc.DeviceBuffer(8, init=np.array([start_bit], dtype=np.uint64).view(np.uint8))
    mhc._check(lib.mh_dev_encode_ctx_fine(m.handle, d_data.ptr, n, ctx0, d_start.ptr if d_start else None, d_payload.ptr, cap,
                                          d_nbits.ptr, d_index.ptr, chunk, d_fine.ptr if fine else None, d_ws.ptr, wsb, None), "encode")
    rc = lib.mh_dev_status(d_ws.ptr, None)
    nbits = int(d_nbits.download(np.uint64)[0])
    raw = d_payload.download()
    return (rc, nbits, raw, d_index.download(np.uint64)[:(n + chunk - 1) // chunk].copy(),
            d_fine.download(np.uint32)[:(n + 63) // 64].copy(), lib.mh_dev_encode_path(d_ws.ptr, None))


def both(mhc, m, data, **kw):
    a = encode(mhc, m, data, **kw)
    os.environ["MH_ENCODE2_PATH"] = "two_pass"
    try:
        b = encode(mhc, m, data, **kw)
    finally:
        del os.environ["MH_ENCODE2_PATH"]
    assert a[5] == ENC_CHAIN and b[5] == ENC_LENGTH_PASS, (a[5], b[5])
    return a, b


@pytest.fixture(scope="module")
def text_model(mhc):
    return mhc.Model.from_data(text_like(4 << 20, 1).tobytes(), 2)


@pytest.mark.parametrize("n", [1, 2, 5, 16, 17, 63, 64, 1023, 1024, 1025, 4095, 4096, 4097, 4096 * 2 + 3, 4096 * 16, 4096 * 16 + 1,
                               4096 * 64 - 1, 4096 * 64, 4096 * 64 + 9, 4096 * 65 + 4000, (3 << 20) + 77])
def test_chain_encoder_equals_the_two_pass_encoder_and_the_oracle_parity_unpinned(mhc, oracle, text_model, n):
    data = text_like(n, n)
    (rc, nbits, raw, idx, fine, _), (rc2, nbits2, raw2, idx2, fine2, _) = both(mhc, text_model, data)
    assert rc == 0 and rc2 == 0
    nb = (nbits + 7) // 8
    assert nbits == nbits2 and raw[:nb].tobytes() == raw2[:nb].tobytes()
    assert np.array_equal(idx, idx2) and np.array_equal(fine, fine2)
    assert np.all(raw[(nb + 3) // 4 * 4:] == 0xEE), "wrote past the payload's last dword"
    o = oracle.Model.from_table(text_model.table_bytes())
    ref, ref_bits = o.compress(data.tobytes())
    assert (nbits, raw[:nb].tobytes()) == (ref_bits, ref[1:])


@pytest.mark.parametrize("start", [1, 5, 7, 8 * 1000 + 3])
def test_chain_encoder_with_a_start_offset_parity_unpinned(mhc, text_model, start):
    """A shard emitted pre-shifted (mh_dev_encode_ctx with a start bit): the first byte keeps its top bits clear for the seam."""
    data = text_like(4096 * 70 + 123, start)
    a, b = both(mhc, text_model, data, start_bit=start, ctx0=0x6520)
    assert a[0] == 0 and b[0] == 0
    nb = (a[1] + 7) // 8
    assert a[1] == b[1] and a[2][:nb].tobytes() == b[2][:nb].tobytes()
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert a[2][0] >> (8 - (start & 7)) == 0


def test_chain_encoder_with_symbols_outside_the_hot_image_parity_unpinned(mhc, oracle):
    """Text with a sprinkle of bytes and contexts the LDS image has no slot for (an escape entry sends the wave's sub-step
    through the symbol-by-symbol path with the full tables), and the same in the symbols a tile encodes for its last dword."""
    rng = np.random.default_rng(11)
    data = text_like((16 << 20) + 5, 4)
    # (the model builder hands the image over only when it covers all but 1e-5 of the input: a few dozen strangers in 16 MiB)
    pos = rng.integers(0, data.size, size=24)
    data[pos] = rng.integers(128, 256, size=pos.size).astype(np.uint8)
    data[4096 * np.arange(1, 9)] = 200                         # the first symbol after a tile: its predecessor encodes it too
    data[4096 * 64 * 3 - 1] = 201                              # and the last symbol of a ticket's last tile
    m = mhc.Model.from_data(data.tobytes(), 2)
    (rc, nbits, raw, idx, fine, _), (rc2, nbits2, raw2, idx2, fine2, _) = both(mhc, m, data)
    assert rc == 0 and rc2 == 0
    nb = (nbits + 7) // 8
    assert nbits == nbits2 and raw[:nb].tobytes() == raw2[:nb].tobytes()
    assert np.array_equal(idx, idx2) and np.array_equal(fine, fine2)
    ref, ref_bits = oracle.Model.from_data(data.tobytes(), 2).compress(data.tobytes())
    assert (nbits, raw[:nb].tobytes()) == (ref_bits, ref[1:])
    blob = bytes([mhc.lib().mh_stream_header(m.handle, nbits)]) + raw[:nb].tobytes()
    assert m.decompress(blob, index=idx, chunk_symbols=256, n_symbols=data.size) == data.tobytes()


def test_chain_encoder_reports_a_buffer_too_small_parity_unpinned(mhc, text_model):
    data = text_like(4096 * 40, 8)
    rc, nbits, raw, _, _, path = encode(mhc, text_model, data)
    assert rc == 0 and path == ENC_CHAIN
    cap = ((nbits + 7) // 8) // 2 & ~3
    rc2, nbits2, raw2, _, _, _ = encode(mhc, text_model, data, cap=cap)
    assert rc2 == mhc.MH_ERR_CAPACITY and nbits2 == nbits          # the length is still reported
    assert np.all(raw2[cap:] == 0xEE), "wrote past the capacity"


def test_chain_encoder_one_bit_codes_parity_unpinned(mhc, oracle):
    """A run of one byte: every context that occurs has one successor, every code is one bit — a tile is 4096 bits, and the
    32 symbols a tile may encode for its last dword are all needed."""
    data = np.full(4096 * 33 + 17, 65, dtype=np.uint8)
    m = mhc.Model.from_data(data.tobytes(), 2)
    for start in (None, 3):
        a, b = both(mhc, m, data, start_bit=start)
        nb = (a[1] + 7) // 8
        assert a[0] == 0 and a[1] == b[1] and a[2][:nb].tobytes() == b[2][:nb].tobytes()
    ref, ref_bits = oracle.Model.from_data(data.tobytes(), 2).compress(data.tobytes())
    a = encode(mhc, m, data)
    assert (a[1], a[2][:(a[1] + 7) // 8].tobytes()) == (ref_bits, ref[1:])


@pytest.mark.parametrize("hook", ["timeout", "timeout_follower"])
def test_chain_encoder_giving_up_is_reported_and_the_host_path_retries_parity_unpinned(mhc, oracle, text_model, hook):
    """The one-pass encoder's waits are bounded; when one runs out (forced here: the leader's look-back, or a follower's wait
    for the leader — ADVICE r03: that wave's tile stays unwritten, so it must report too) the workspace says
    MH_ERR_TIMEOUT, and mh_encode*, which synchronises anyway, runs the two-pass pair instead — and counts it
    (mh_last_encode_retries: 1 under the forced timeout, 0 otherwise)."""
    data = text_like(4096 * 40 + 5, 21)
    lib = mhc.lib()
    ref, ref_bits = oracle.Model.from_table(text_model.table_bytes()).compress(data.tobytes())
    blob, nbits, idx = text_model.compress(data.tobytes(), chunk_symbols=256)
    assert (nbits, blob) == (ref_bits, ref) and lib.mh_last_encode_retries() == 0
    total0 = lib.mh_total_encode_retries()
    os.environ["MH_CHAIN_PROBE"] = hook
    try:
        rc, _, _, _, _, path = encode(mhc, text_model, data)
        assert rc == mhc.MH_ERR_TIMEOUT and path == ENC_CHAIN
        blob, nbits, idx = text_model.compress(data.tobytes(), chunk_symbols=256)
        assert lib.mh_last_encode_retries() == 1 and lib.mh_total_encode_retries() == total0 + 1
    finally:
        del os.environ["MH_CHAIN_PROBE"]
    assert (nbits, blob) == (ref_bits, ref)
    blob, nbits, idx = text_model.compress(data.tobytes(), chunk_symbols=256)
    assert (nbits, blob) == (ref_bits, ref) and lib.mh_last_encode_retries() == 0
