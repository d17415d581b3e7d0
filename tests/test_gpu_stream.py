"""Streams without an index in TWO passes over the payload (round 5; mh_dev_decode_stream_states + mh_dev_decode_stream_emit,
csrc/mh_tile.hip segment_decode_kernel).  What the reference writes carries no index (src/coding.cpp:35-59) and is decoded by
i_coding_provider::decompress bit by bit (src/coding.cpp:96-160); here the first pass finds every 352-bit segment's entry state
and symbol count, the second decodes the segments again and writes the bytes.  Streams and tables come from the oracle (= the
reference's files); which way a stream went is asserted by path code (6 = states + segment decoder, 0 = not taken), never by
the clock."""
import os

import numpy as np
import pytest

import __graft_entry__ as entry
from conftest import golden

pytestmark = pytest.mark.gpu

PATH_STATES = 6


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


def stream_decode(mhc, m, payload, nbits, prev0=0x20, cap=None, pad=0x5A):
    """(status of pass 1, path, symbol count, status of pass 2, output buffer incl. guard bytes) through the two device calls."""
    lib = mhc.lib()
    pl = np.frombuffer(payload, dtype=np.uint8)
    d_pl = mhc.DeviceBuffer(pl.size + 64, init=np.concatenate([pl, np.zeros(64, dtype=np.uint8)]))
    d_ns = mhc.DeviceBuffer(8)
    iws = int(lib.mh_dev_build_index_workspace(nbits))
    d_iws = mhc.DeviceBuffer(iws)
    assert lib.mh_dev_decode_stream_states(m.handle, d_pl.ptr, nbits, prev0, d_ns.ptr, d_iws.ptr, iws, None) == 0
    st1, path = lib.mh_dev_status(d_iws.ptr, None), lib.mh_dev_index_path(d_iws.ptr, None)
    ns = int(d_ns.download(np.uint64)[0])
    if path != PATH_STATES:
        return st1, path, ns, None, None
    cap = ns if cap is None else cap
    guard = 4096
    d_out = mhc.DeviceBuffer(cap + guard, init=np.full(cap + guard, pad, dtype=np.uint8))
    assert lib.mh_dev_decode_stream_emit(m.handle, d_pl.ptr, nbits, prev0, d_out.ptr, cap, d_iws.ptr, iws, None) == 0
    return st1, path, ns, lib.mh_dev_status(d_iws.ptr, None), d_out.download()


@pytest.mark.parametrize("kind,n", [("zipf", (12 << 20) + 1001), ("zipf", 300_000), ("text", (6 << 20) + 77),
                                    ("zipf", 4096 * 64 * 3), ("text", 262_144 + 5), ("zipf", 36864 * 8 * 4)])
def test_two_passes_give_the_input_back(mhc, oracle, kind, n):
    """Sizes on every side of the segment / tile boundaries; the bytes behind the output stay untouched."""
    data = zipf_bytes(n, n & 255) if kind == "zipf" else text_like(n, n & 255)
    om = oracle.Model.from_data(data.tobytes(), 1)
    blob, nbits = om.compress(data.tobytes())
    m = mhc.Model.from_table(om.table_bytes())
    lens = np.asarray(om.codes()[0])
    assert m.min_code_len == int(lens[lens > 0].min())
    assert mhc.Model.from_data(data.tobytes(), order=1).min_code_len == m.min_code_len      # device-built = loaded from the table
    st1, path, ns, st2, out = stream_decode(mhc, m, blob[1:], nbits)
    if nbits < 1 << 20:
        assert (st1, path) == (0, 0)                              # under a megabit: not this path's business
        return
    assert (st1, path, ns, st2) == (0, PATH_STATES, n, 0)
    assert np.array_equal(out[:n], data)
    assert np.all(out[n:] == 0x5A), "wrote beyond the stream's last byte"
    # and the host-buffer call (what host/coding.cpp's decompress makes) takes the same way
    assert m.decompress(blob) == data.tobytes()
    assert mhc.lib().mh_last_index_path() == PATH_STATES


def test_segments_of_many_symbols_take_several_rounds(mhc, oracle):
    """A source of mostly 1- and 2-bit codes: 180 to 352 symbols per 352-bit segment, i.e. up to five rounds of 80 steps per
    lane, the lanes of a wave finishing in different rounds."""
    rng = np.random.default_rng(11)
    n = 9 << 20
    data = rng.choice(6, size=n, p=[0.80, 0.08, 0.05, 0.04, 0.02, 0.01]).astype(np.uint8)
    data[rng.integers(0, n, 4000)] = rng.integers(6, 200, 4000).astype(np.uint8)       # a few long codes in between
    data[1 << 20:(1 << 20) + 70000] = 0                                                # a run: 352 symbols in every segment
    om = oracle.Model.from_data(data.tobytes(), 1)
    blob, nbits = om.compress(data.tobytes())
    assert n / (nbits / 352.0) > 180
    m = mhc.Model.from_table(om.table_bytes())
    st1, path, ns, st2, out = stream_decode(mhc, m, blob[1:], nbits)
    if path != PATH_STATES:                                       # (a model whose codes the tile tables do not resolve)
        pytest.skip("model not eligible: path %d" % path)
    assert (st1, ns, st2) == (0, n, 0)
    assert np.array_equal(out[:n], data) and np.all(out[n:] == 0x5A)


def test_another_start_context(mhc, oracle):
    """A shard: the stream starts in context 7, not ' '."""
    data = zipf_bytes(3 << 20, 5)
    om = oracle.Model.from_data(b"\x07" + data.tobytes(), 1)
    m = mhc.Model.from_table(om.table_bytes())
    lens = np.asarray(om.codes()[0]).astype(np.int64)
    blob, nbits = om.compress(b"\x07" + data.tobytes())
    skip = int(lens[0x20 * 256 + 7])
    bits = np.unpackbits(np.frombuffer(blob[1:], dtype=np.uint8))[skip:nbits]
    st1, path, ns, st2, out = stream_decode(mhc, m, np.packbits(bits).tobytes(), bits.size, prev0=7)
    assert (st1, path, ns, st2) == (0, PATH_STATES, data.size, 0)
    assert np.array_equal(out[:data.size], data)


def test_the_reference_s_wiki_html_model_with_15_bit_codes(mhc, oracle):
    """The reference's own test input with the longest codes (15 bits: first level + the full second level of the tile tables),
    tiled to a few megabits: the stream takes the two-pass path and comes back byte for byte."""
    page = golden()["input_wiki_cpp.html"]["data"]
    data = page * 12
    om = oracle.Model.from_data(data, 1)
    assert int(np.asarray(om.codes()[0]).max()) >= 13
    blob, nbits = om.compress(data)
    m = mhc.Model.from_table(om.table_bytes())
    st1, path, ns, st2, out = stream_decode(mhc, m, blob[1:], nbits)
    assert (st1, path, ns, st2) == (0, PATH_STATES, len(data), 0)
    assert out[:len(data)].tobytes() == data


def test_a_buffer_too_small_is_reported_and_respected(mhc, oracle):
    data = zipf_bytes(2 << 20, 3)
    om = oracle.Model.from_data(data.tobytes(), 1)
    blob, nbits = om.compress(data.tobytes())
    m = mhc.Model.from_table(om.table_bytes())
    cap = data.size - 100_000
    st1, path, ns, st2, out = stream_decode(mhc, m, blob[1:], nbits, cap=cap)
    assert (st1, path, ns) == (0, PATH_STATES, data.size)
    assert st2 == mhc.MH_ERR_CAPACITY
    assert np.all(out[cap:] == 0x5A), "stored beyond the capacity"


def test_a_stream_that_does_not_belong_to_the_table_is_reported(mhc, oracle):
    """Cut short inside its last code, or decoded with the table of another source: MH_ERR_CORRUPT from one of the passes (or
    a symbol count that differs) — never a wild access; the guard bytes stay."""
    data = zipf_bytes(2 << 20, 9)
    om = oracle.Model.from_data(data.tobytes(), 1)
    blob, nbits = om.compress(data.tobytes())
    m = mhc.Model.from_table(om.table_bytes())
    st1, path, ns, st2, out = stream_decode(mhc, m, blob[1:], nbits - 3)
    assert path in (0, PATH_STATES)
    assert st1 == mhc.MH_ERR_CORRUPT or path == 0 or st2 == mhc.MH_ERR_CORRUPT or ns != data.size
    other = mhc.Model.from_table(oracle.Model.from_data(text_like(1 << 20, 4).tobytes() + bytes(range(256)), 1).table_bytes())
    st1, path, ns, st2, out = stream_decode(mhc, other, blob[1:], nbits)
    assert st1 == mhc.MH_ERR_CORRUPT or path == 0 or st2 == mhc.MH_ERR_CORRUPT or ns != data.size
    if out is not None:
        assert np.all(out[max(ns, 0):][-4096:] == 0x5A)


def test_streams_the_path_does_not_take_fall_back_to_the_index(mhc, oracle):
    """Uniform bytes (8-bit codes in every context: a code-length lattice) are not eligible: path 0 from the states call, and
    the host-buffer decode builds an index as before."""
    data = np.random.default_rng(5).integers(0, 256, 3 << 20, dtype=np.uint8)
    om = oracle.Model.from_data(data.tobytes(), 1)
    blob, nbits = om.compress(data.tobytes())
    m = mhc.Model.from_table(om.table_bytes())
    st1, path, ns, _, _ = stream_decode(mhc, m, blob[1:], nbits)
    assert (st1, path) == (0, 0)
    assert m.decompress(blob) == data.tobytes()
    assert mhc.lib().mh_last_index_path() in (1, 2, 3)
    # MH_DECODE_NO_STREAM=1 (A/B switch, same bytes): an eligible stream through index + tile decoder
    z = zipf_bytes(3 << 20, 6)
    oz = oracle.Model.from_data(z.tobytes(), 1)
    bz, _ = oz.compress(z.tobytes())
    mz = mhc.Model.from_table(oz.table_bytes())
    os.environ["MH_DECODE_NO_STREAM"] = "1"
    try:
        assert mz.decompress(bz) == z.tobytes()
        assert mhc.lib().mh_last_index_path() == 5
    finally:
        del os.environ["MH_DECODE_NO_STREAM"]


@pytest.mark.parametrize("seed", range(12))
def test_random_sources_through_both_passes(mhc, oracle, seed):
    """Seeded sources of many shapes (alphabets of 3..256 symbols, Zipf exponents 0.3..2.5, iid or first-order Markov): whatever
    path the stream takes, the host-buffer decode of the oracle's stream gives the input back; where the two-pass path applies
    its output equals the input byte for byte."""
    from test_gpu_index_tiles import _random_source
    data = np.concatenate([_random_source(seed)] * 3)             # (over a megabit for most)
    om = oracle.Model.from_data(data.tobytes(), 1)
    blob, nbits = om.compress(data.tobytes())
    m = mhc.Model.from_table(om.table_bytes())
    st1, path, ns, st2, out = stream_decode(mhc, m, blob[1:], nbits)
    assert st1 == 0
    if path == PATH_STATES:
        assert (ns, st2) == (data.size, 0)
        assert np.array_equal(out[:data.size], data) and np.all(out[data.size:] == 0x5A)
    assert m.decompress(blob) == data.tobytes()


@pytest.mark.parametrize("kind", ["REDO_LDS", "REDO_L2_DIRECT"])
def test_codes_longer_than_the_tile_tables_resolve_go_to_the_walk(mhc, oracle, kind):
    """Real files of some size have contexts with one-in-a-million successors: codes of 20 bits and more, which the tile tables
    (first level 7 bits + second level of at most 8) do not resolve.  Their segments cannot be decoded by the states pass — they
    are always listed, the repair kernel (general tables, tree walk) settles their state and marks them — and the segment
    decoder leaves exactly those to a one-thread-per-segment walk; every other segment takes the fast way.  Models and data:
    the chunk decoder's redo recipes (Fibonacci weights: codes of up to 25 bits, a few hundred of them planted in the data)."""
    from test_gpu_decode_variants import recipe
    counts, data = recipe(kind)
    counts = counts.reshape(-1) + oracle.histogram_o1(data.tobytes()).astype(np.uint64)
    om = oracle.Model.from_counts(counts, 1)
    assert int(np.asarray(om.codes()[0]).max()) > 15
    blob, nbits = om.compress(data.tobytes())
    m = mhc.Model.from_table(om.table_bytes())
    st1, path, ns, st2, out = stream_decode(mhc, m, blob[1:], nbits)
    assert (st1, path, ns, st2) == (0, PATH_STATES, data.size, 0)
    assert np.array_equal(out[:data.size], data) and np.all(out[data.size:] == 0x5A)
    assert m.decompress(blob) == data.tobytes() and mhc.lib().mh_last_index_path() == PATH_STATES
