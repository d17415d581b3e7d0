"""Every instantiation of the chunk decoder the launcher can select (csrc/mh_decode.hip dec_cfg, include/mh.h
mh_dev_decode_variant), each driven by a model and a stream that make the launcher choose it — asserted from the variant code the
launch leaves in the workspace, never forced by a switch — and compared with the input byte for byte; the encoded stream is the
oracle's (= the reference's, src/coding.cpp:61-94) in every case.  The decode semantics are i_coding_provider::decompress,
src/coding.cpp:118-157."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as entry

pytestmark = pytest.mark.gpu

NAMES = {0: "LDS_WIDE", 1: "LDS_SHORT", 2: "LDS_TWO_LEVEL", 3: "LDS_TWO_LEVEL_P8", 4: "L2_DIRECT", 5: "L2_DIRECT_H2", 6: "L2_DIRECT_H3",
         7: "L2_DIRECT_H4", 8: "L2_DIRECT_H8"}


@pytest.fixture(scope="module")
def mhc():
    mod = entry.load_package()
    mod.lib()
    assert mod.device_count() >= 1
    mod.lib().mh_dev_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
    return mod


def zipf_counts(s, k=256, scale=1 << 20):
    return (np.floor(scale / np.arange(1, k + 1) ** s) + 1).astype(np.uint64)


def sample_rows(counts, n, seed):
    """n bytes of a first-order chain whose transition weights are `counts` (256 x 256), starting in context ' '."""
    rng = np.random.default_rng(seed)
    c = counts.reshape(256, 256).astype(np.float64)
    live = c.sum(axis=1) > 0
    cdf = np.cumsum(np.where(live[:, None], c, 1.0), axis=1)
    cdf /= cdf[:, -1:]
    if (c[live] == c[live][0]).all() and live.all():             # every context alike: iid
        return np.searchsorted(cdf[0], rng.random(n), side="right").astype(np.uint8)
    u = rng.random(n)
    out = np.empty(n, dtype=np.uint8)
    prev = 0x20 if live[0x20] else int(np.flatnonzero(live)[0])
    for i in range(n):
        prev = int(np.searchsorted(cdf[prev], u[i], side="right"))
        out[i] = prev
    return out


def device_decode(mhc, m, blob, nbits, idx, n, chunk):
    lib = mhc.lib()
    pl = np.frombuffer(blob[1:], dtype=np.uint8)
    d_pl = mhc.DeviceBuffer(pl.size + 64, init=np.concatenate([pl, np.zeros(64, dtype=np.uint8)]))
    d_idx = mhc.DeviceBuffer(idx.size * 8, init=np.ascontiguousarray(idx, dtype=np.uint64))
    d_out = mhc.DeviceBuffer(n + 64, init=np.full(n + 64, 0x5A, dtype=np.uint8))
    wsb = int(lib.mh_dev_decode_workspace(nbits, n, chunk))
    d_ws = mhc.DeviceBuffer(wsb)
    mhc._check(lib.mh_dev_decode(m.handle, d_pl.ptr, nbits, d_out.ptr, n, d_idx.ptr, chunk, d_ws.ptr, wsb, None), "mh_dev_decode")
    status = lib.mh_dev_status(d_ws.ptr, None)
    redo = int(d_ws.download(np.uint32)[16])                       # chunks the hot loop handed to the redo pass (workspace + 64)
    return status, lib.mh_dev_decode_path(d_ws.ptr, None), lib.mh_dev_decode_variant(d_ws.ptr, None), redo, d_out.download()


def fib_row(k=26):
    f = [1, 1]
    while len(f) < k:
        f.append(f[-1] + f[-2])
    w = np.zeros(256, dtype=np.uint64)
    w[:k] = np.array(f[::-1], dtype=np.uint64)
    return w


def recipe(kind):
    """(counts 65536, bytes) that make the launcher pick the variant named by `kind`."""
    c = np.zeros((256, 256), dtype=np.uint64)
    n = (3 << 20) + 4321
    if kind == "LDS_WIDE":                       # 8-bit codes everywhere, ratio 1.0
        c[:] = 1000
        return c, np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
    if kind == "LDS_SHORT":                      # sixteen symbols, codes of 1..6 bits: no second level, ratio ~0.35
        c[:16, :16] = zipf_counts(1.0, 16)
    elif kind == "LDS_TWO_LEVEL_P8":             # forty symbols, codes up to 9 bits: a small second level beside a full first level
        c[:40, :40] = zipf_counts(1.5, 40)
    elif kind == "LDS_TWO_LEVEL":                # 64 contexts of 256 Zipf symbols, the others with one successor: fits LDS at P = 7 only
        c[:64] = zipf_counts(1.1)
        c[64:, 0] = 1
    elif kind.startswith("L2_DIRECT"):           # 256 busy contexts: second level in L2; the exponent sets the longest code
        s = {"L2_DIRECT": 1.5, "L2_DIRECT_H2": 0.9, "L2_DIRECT_H3": 1.1, "L2_DIRECT_H4": 1.2, "L2_DIRECT_H8": 2.0}[kind]
        c[:] = zipf_counts(s)
    elif kind == "REDO_LDS":                     # Fibonacci weights: codes of up to 25 bits, tables in LDS
        c[:26] = fib_row()
    elif kind == "REDO_L2_DIRECT":               # 255 Zipf contexts + one Fibonacci context: second level in L2, codes of up to 25 bits
        c[:] = zipf_counts(1.1)
        c[0] = fib_row() * np.uint64(1000)        # (heavy enough that the planted pairs below do not reshape its tree)
    else:
        raise ValueError(kind)
    data = sample_rows(c, n if kind.startswith("L2_DIRECT") else (1 << 20) + 77, 7)     # (a chain that is not iid is drawn symbol by symbol)
    if kind.startswith("REDO"):                  # the chain almost never reaches the 20-bit symbols by itself: plant some
        rng = np.random.default_rng(3)
        pos = rng.integers(2, data.size - 2, 700)
        if kind == "REDO_LDS":
            data[pos] = rng.integers(20, 26, 700).astype(np.uint8)
        else:
            data[pos - 1] = 0                     # context 0 is the Fibonacci one
            data[pos] = rng.integers(20, 26, 700).astype(np.uint8)
    return c, data


@pytest.mark.parametrize("kind", ["LDS_WIDE", "LDS_SHORT", "LDS_TWO_LEVEL", "LDS_TWO_LEVEL_P8", "L2_DIRECT", "L2_DIRECT_H2", "L2_DIRECT_H3",
                                  "L2_DIRECT_H4", "L2_DIRECT_H8"])
@pytest.mark.parametrize("chunk", [256, 1024])
def test_every_selectable_variant_decodes_the_oracle_s_stream(mhc, oracle, kind, chunk):
    counts, data = recipe(kind)
    # the model must know every pair of the data: add the data's own histogram to the recipe's weights
    counts = counts.reshape(-1) + oracle.histogram_o1(data.tobytes()).astype(np.uint64)
    m = mhc.Model.from_counts(counts, 1)
    om = oracle.Model.from_counts(counts, 1)
    blob, nbits, idx = m.compress(data.tobytes(), chunk_symbols=chunk)
    ref, ref_bits = om.compress(data.tobytes())
    assert (nbits, blob) == (ref_bits, ref)
    status, path, variant, _, out = device_decode(mhc, m, blob, nbits, np.asarray(idx, dtype=np.uint64), data.size, chunk)
    assert status == 0 and path == 2                              # the chunk decoder
    assert NAMES.get(variant) == kind, (variant, NAMES.get(variant), m.decode_layout(), m.max_code_len)
    assert np.array_equal(out[:data.size], data) and np.all(out[data.size:] == 0x5A)


@pytest.mark.parametrize("kind,main", [("REDO_LDS", "LDS_TWO_LEVEL_P8"), ("REDO_L2_DIRECT", "L2_DIRECT_H8")])
def test_redo_variants_take_the_chunks_with_codes_longer_than_both_levels(mhc, oracle, kind, main):
    """Codes of more than 16 bits are resolved by neither table level: the hot loop lists the chunk, the redo pass (one lane
    per chunk, tree walk: variants 9 / 10, the one of the main variant's table layout) decodes it again."""
    counts, data = recipe(kind)
    counts = counts.reshape(-1) + oracle.histogram_o1(data.tobytes()).astype(np.uint64)
    m = mhc.Model.from_counts(counts, 1)
    om = oracle.Model.from_counts(counts, 1)
    assert m.max_code_len > 16
    blob, nbits, idx = m.compress(data.tobytes(), chunk_symbols=256)
    assert (nbits, blob) == om.compress(data.tobytes())[::-1]
    status, path, variant, redo, out = device_decode(mhc, m, blob, nbits, np.asarray(idx, dtype=np.uint64), data.size, 256)
    assert status == 0 and path == 2 and NAMES.get(variant) == main, (variant, m.decode_layout())
    assert redo > 0, "no chunk went to the redo pass: the recipe has no long code in the data"
    assert np.array_equal(out[:data.size], data)
