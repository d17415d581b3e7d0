"""The order-2 histogram's two paths (csrc/mh_hist2.hip) — PARITY UNPINNED like all of order 2 (the reference has none,
README.md:158-166; the spec is the generalised oracle).  Sources with a few thousand live (context, symbol) keys stay in
the LDS tag cache; sources with millions of them (Zipf or uniform bytes) are partitioned by the context's high byte and
every bucket counted like an order-1 histogram.  The choice is made on the device per slab; here both paths and the
choice itself are held against the oracle's counts, bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

import __graft_entry__ as entry

pytestmark = pytest.mark.gpu

TAG, PARTITION = 1, 2


@pytest.fixture(scope="module")
def mhc():
    mod = entry.load_package()
    mod.lib()
    assert mod.device_count() >= 1
    return mod


@pytest.fixture(scope="module")
def diag(mhc):
    """libmhc_diag.so (-DMH_EXP_PROBES): the only library that reads MH_HIST2_FORCE / MH_HIST2_SLAB."""
    d = C.CDLL(os.path.join(os.path.dirname(mhc.LIB_PATH), "libmhc_diag.so"))
    vp, sz = C.c_void_p, C.c_size_t
    d.mh_dev_histogram_o2_ws.argtypes = [vp, sz, C.c_uint16, vp, vp, sz, vp]
    d.mh_dev_histogram_o2_workspace.argtypes = [sz]
    d.mh_dev_histogram_o2_workspace.restype = sz
    d.mh_dev_index_path.argtypes = [vp, vp]
    d.mh_dev_status.argtypes = [vp, vp]
    return d


def text_like(n, seed):
    rng = np.random.default_rng(seed)
    words = [b"lorem", b"ipsum", b"dolor", b"sit", b"amet", b"consectetur", b"adipiscing", b"elit", b"sed", b"do",
             b"eiusmod", b"tempor", b"incididunt", b"ut", b"labore", b"et", b"dolore", b"magna", b"aliqua"]
    piece = bytearray()
    while len(piece) < (1 << 20):
        piece += words[int(rng.integers(len(words)))] + (b". " if rng.random() < 0.1 else b" ")
    reps = n // len(piece) + 1
    return np.frombuffer(bytes(piece) * reps, dtype=np.uint8)[:n].copy()


def uniform(n, seed):
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8)


def zipf(n, seed, s=1.1):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, 257) ** s
    return rng.choice(256, size=n, p=w / w.sum()).astype(np.uint8)


def run(mhc, lib, data, ctx0=0x2020, with_ws=True):
    n = data.size
    d_data = mhc.DeviceBuffer(n + 32, init=np.concatenate([data, np.zeros(32, dtype=np.uint8)]))
    d_counts = mhc.DeviceBuffer((1 << 24) * 8)
    ws = int(lib.mh_dev_histogram_o2_workspace(n)) if with_ws else 0
    d_ws = mhc.DeviceBuffer(max(ws, 64))
    rc = lib.mh_dev_histogram_o2_ws(d_data.ptr, n, ctx0, d_counts.ptr, d_ws.ptr if with_ws else None, ws, None)
    assert rc == 0
    assert lib.mh_dev_status(d_ws.ptr, None) == 0 if with_ws else True
    paths = lib.mh_dev_index_path(d_ws.ptr, None) if with_ws else 0
    return d_counts.download(np.uint64), paths


def expect(oracle, data, ctx0=0x2020):
    """The oracle starts in the context of two spaces; another start context only moves the first two keys."""
    want = oracle.histogram_o2(data)
    if ctx0 != 0x2020 and data.size >= 2:
        b0, b1 = int(data[0]), int(data[1])
        want[(0x2020 << 8) | b0] -= 1
        want[(((0x20 << 8) | b0) << 8) | b1] -= 1
        want[(ctx0 << 8) | b0] += 1
        want[((((ctx0 & 255) << 8) | b0) << 8) | b1] += 1
    return want


def test_workspace_size_is_bounded(mhc):
    lib = mhc.lib()
    assert lib.mh_dev_histogram_o2_workspace(1 << 20) == 64
    w = lib.mh_dev_histogram_o2_workspace(64 << 20)                  # 2 bytes per byte + the items' images (128 KiB each) + tables
    assert (128 << 20) < w < (128 << 20) + (36 << 20)
    assert lib.mh_dev_histogram_o2_workspace(64 << 30) < (4 << 30) + (98 << 20)     # slabs of 2 GiB: 4 GiB of pairs at most


@pytest.mark.parametrize("extra", [0, 16, 12345, 65536 + 77])
def test_uniform_bytes_take_the_partition_path_parity_unpinned(mhc, oracle, extra):
    data = uniform((48 << 20) + extra, 11 + extra)
    got, paths = run(mhc, mhc.lib(), data)
    assert paths == PARTITION                                    # (the slab's first 4 MiB, the sample, always go through the cache)
    assert np.array_equal(got, expect(oracle, data))


def test_zipf_bytes_take_the_partition_path_parity_unpinned(mhc, oracle):
    data = zipf(40 << 20, 5)
    got, paths = run(mhc, mhc.lib(), data, ctx0=0x4142)
    assert paths == PARTITION
    assert np.array_equal(got, expect(oracle, data, 0x4142))


def test_text_stays_in_the_tag_cache_parity_unpinned(mhc, oracle):
    data = text_like((40 << 20) + 3, 7)
    got, paths = run(mhc, mhc.lib(), data)
    assert paths == TAG
    assert np.array_equal(got, expect(oracle, data))


def test_without_workspace_or_below_the_threshold_everything_goes_through_the_cache(mhc, oracle):
    data = uniform(34 << 20, 3)
    got, _ = run(mhc, mhc.lib(), data, with_ws=False)
    assert np.array_equal(got, expect(oracle, data))
    small = uniform(3 << 20, 4)
    got, paths = run(mhc, mhc.lib(), small)
    assert paths == TAG and np.array_equal(got, expect(oracle, small))


def test_host_call_uses_the_workspace(mhc, oracle):
    data = uniform((36 << 20) + 5, 21)
    assert np.array_equal(mhc.histogram_o2(data.tobytes()), oracle.histogram_o2(data))


@pytest.mark.parametrize("name", ["zeros", "two_values", "text", "ramp"])
def test_partition_path_forced_on_skewed_sources_parity_unpinned(mhc, diag, oracle, name):
    """What the device would never choose for these sources, forced in the diagnostic library: one bucket takes every
    position (zeros: 40 M adds to ONE counter of the bucket kernel — the guard-bit fix-ups — and one run per tile in the
    scatter), two buckets, the few buckets of text, and a ramp that visits every bucket in turn."""
    n = (40 << 20) + 4321
    data = {"zeros": lambda: np.zeros(n, dtype=np.uint8),
            "two_values": lambda: (np.random.default_rng(1).integers(0, 2, size=n, dtype=np.uint8) * 255).astype(np.uint8),
            "text": lambda: text_like(n, 2),
            "ramp": lambda: (np.arange(n, dtype=np.uint32) >> 3).astype(np.uint8)}[name]()
    os.environ["MH_HIST2_FORCE"] = "2"
    try:
        got, paths = run(mhc, diag, data)
    finally:
        del os.environ["MH_HIST2_FORCE"]
    assert paths == PARTITION
    assert np.array_equal(got, expect(oracle, data))


def test_the_choice_is_made_per_slab_parity_unpinned(mhc, diag, oracle):
    """Slabs of 32 MiB (diagnostic library): text, then uniform bytes, then text again and a short last slab — the text slabs
    stay in the cache, the uniform ones are partitioned, the short one follows its predecessor."""
    parts = [text_like(32 << 20, 1), uniform(64 << 20, 2), text_like(32 << 20, 3), uniform((5 << 20) + 99, 4)]
    data = np.concatenate(parts)
    os.environ["MH_HIST2_SLAB"] = str(32 << 20)
    try:
        got, paths = run(mhc, diag, data)
    finally:
        del os.environ["MH_HIST2_SLAB"]
    assert paths == TAG | PARTITION
    assert np.array_equal(got, expect(oracle, data))


def test_forced_cache_equals_forced_partition(mhc, diag):
    data = zipf((33 << 20) + 17, 9, s=0.7)
    got = {}
    for force in ("1", "2"):
        os.environ["MH_HIST2_FORCE"] = force
        try:
            got[force], paths = run(mhc, diag, data, ctx0=0x0102)
        finally:
            del os.environ["MH_HIST2_FORCE"]
        assert paths == int(force)
    assert np.array_equal(got["1"], got["2"]) and int(got["1"].sum()) == data.size


@pytest.mark.parametrize("seed", [101, 202, 303, 404])
def test_partition_path_random_mixtures_parity_unpinned(mhc, diag, oracle, seed):
    """Random sizes and mixtures of sources, the partition path forced (so that skewed pieces go through it too) and slabs of
    32 MiB: every slab boundary, tile boundary and carried remainder lands somewhere else each time."""
    rng = np.random.default_rng(seed)
    parts, left = [], int(rng.integers(33 << 20, 90 << 20)) + int(rng.integers(0, 65536))
    while left > 0:
        n = min(left, int(rng.integers(1, 24 << 20)))
        kind = int(rng.integers(0, 5))
        parts.append(uniform(n, seed + len(parts)) if kind == 0 else zipf(n, seed + len(parts), s=float(rng.uniform(0.5, 2.0))) if kind == 1
                     else text_like(n, seed + len(parts)) if kind == 2 else np.full(n, int(rng.integers(0, 256)), dtype=np.uint8) if kind == 3
                     else (uniform(n, seed + len(parts)) & np.uint8(rng.integers(1, 256))))
        left -= n
    data = np.concatenate(parts)
    ctx0 = int(rng.integers(0, 65536))
    os.environ["MH_HIST2_FORCE"] = "2"
    os.environ["MH_HIST2_SLAB"] = str(32 << 20)
    try:
        got, paths = run(mhc, diag, data, ctx0=ctx0)
    finally:
        del os.environ["MH_HIST2_FORCE"], os.environ["MH_HIST2_SLAB"]
    assert paths == PARTITION
    assert np.array_equal(got, expect(oracle, data, ctx0))


def test_host_call_in_segments_parity_unpinned(mhc, oracle):
    """mh_histogram_o2 stages a large input through HBM in segments (MH_SEGMENT_BYTES): 40 MiB segments over 100 MiB of uniform
    bytes — every segment takes the partition path on its own, the two-byte context is carried across the seams."""
    data = uniform((100 << 20) + 1234, 77)
    os.environ["MH_SEGMENT_BYTES"] = str(40 << 20)
    try:
        got = mhc.histogram_o2(data.tobytes())
    finally:
        del os.environ["MH_SEGMENT_BYTES"]
    assert np.array_equal(got, oracle.histogram_o2(data))
