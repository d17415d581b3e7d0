"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/mh.h
declares, and its host logic (tree build with the reference's tie-breaks, code/LUT derivation, table
files, stream header) agrees with the oracle and the golden vectors.  No compute call is made here
unless it is to check that it refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import __graft_entry__ as entry
from conftest import ROOT, check_against_golden, golden, golden_names


@pytest.fixture(scope="module")
def mhc():
    entry.build()
    return entry.load_package()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "mh.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", text)) - {"mh_index_entries", "mh_fine_entries"})   # static inline helpers


def test_library_exports_every_declared_symbol(mhc):
    lib = ctypes.CDLL(mhc.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libmhc.so does not export %s" % name
    assert sorted(mhc.EXPORTS) == declared


def test_strerror_and_device_count(mhc):
    assert mhc.lib().mh_strerror(0) == b"ok"
    assert b"corrupt" in mhc.lib().mh_strerror(mhc.MH_ERR_CORRUPT)
    assert mhc.device_count() >= 0


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("order", [0, 1])
def test_host_model_matches_reference_tables(mhc, oracle, name, order):
    """Tree build + table file: counts from the oracle's histogram, model from the product's host code."""
    data = golden()[name]["data"]
    counts = oracle.histogram_o1(data) if order else oracle.histogram_o0(data)
    m = mhc.Model.from_counts(counts, order)
    check_against_golden(name, "e" if order else "eh", m.table_bytes())
    assert m.type == order


@pytest.mark.parametrize("name", ["input_ipsum.txt", "input_wiki_cpp.html", "kat3", "kat4", "one_Z", "empty"])
def test_host_codes_and_lut_match_oracle(mhc, oracle, name):
    data = golden()[name]["data"]
    counts = oracle.histogram_o1(data)
    m = mhc.Model.from_counts(counts, 1)
    o = oracle.Model.from_counts(counts, 1)
    lm, cm = m.codes()
    lo, co = o.codes()
    assert np.array_equal(lm, lo)
    assert np.array_equal(cm, co)
    assert m.max_code_len == int(lo.max())
    assert m.min_code_len == (int(lo[lo > 0].min()) if (lo > 0).any() else 0)      # bounds the symbols a payload can hold
    for prev in (0, 0x20, ord("e"), 255):
        for w in range(0, 256, 3):
            assert m.lut(prev, w) == o.lut(prev, w)


@pytest.mark.parametrize("name", ["input_a.txt", "input_wiki_cpp.txt", "kat2", "nine_Z"])
@pytest.mark.parametrize("order", [0, 1])
def test_table_file_round_trip(mhc, oracle, name, order):
    data = golden()[name]["data"]
    counts = oracle.histogram_o1(data) if order else oracle.histogram_o0(data)
    built = mhc.Model.from_counts(counts, order)
    loaded = mhc.Model.from_table(built.table_bytes())
    assert loaded.type == order
    assert loaded.table_bytes() == built.table_bytes()
    lb, cb = built.codes()
    ll, cl = loaded.codes()
    assert np.array_equal(lb, ll) and np.array_equal(cb, cl)


def test_bad_table_is_rejected(mhc):
    with pytest.raises(mhc.MhError) as e:
        mhc.Model.from_table(b"\xff\xff\xff")          # Markov marker, then a truncated tree
    assert e.value.status == mhc.MH_ERR_BADTABLE
    with pytest.raises(mhc.MhError):
        mhc.Model.from_table(b"")                      # no tree at all


def test_stream_header_helpers(mhc, oracle):
    """Header byte of src/coding.cpp:88 and its validation (src/coding.cpp:100-116)."""
    data = golden()["input_b.txt"]["data"]
    m1 = mhc.Model.from_counts(oracle.histogram_o1(data), 1)
    m0 = mhc.Model.from_counts(oracle.histogram_o0(data), 0)
    lib = mhc.lib()
    assert lib.mh_stream_header(m1.handle, 10) == 0x36      # "36 f3 80": 10 payload bits, Markov
    assert lib.mh_stream_header(m0.handle, 19) == 0x3D      # "3d 0a fb c0": 19 bits, Huffman
    assert lib.mh_stream_header(m1.handle, 16) == 0x30
    nb = ctypes.c_uint64()
    assert lib.mh_stream_parse_header(m1.handle, 0x36, 3, ctypes.byref(nb)) == 0 and nb.value == 10
    assert lib.mh_stream_parse_header(m1.handle, 0x3D, 4, ctypes.byref(nb)) == mhc.MH_ERR_TYPE
    assert lib.mh_stream_parse_header(m0.handle, 0x3D, 4, ctypes.byref(nb)) == 0 and nb.value == 19
    assert lib.mh_stream_parse_header(m1.handle, 0x80, 4, ctypes.byref(nb)) == mhc.MH_ERR_CORRUPT


def test_argument_errors(mhc):
    lib = mhc.lib()
    h = ctypes.c_void_p()
    assert lib.mh_model_from_counts(None, 1, ctypes.byref(h)) == mhc.MH_ERR_ARG
    counts = np.zeros(65536, dtype=np.uint64)
    assert lib.mh_model_from_counts(counts.ctypes.data, 3, ctypes.byref(h)) == mhc.MH_ERR_ARG
    assert lib.mh_model_from_counts(counts.ctypes.data, -1, ctypes.byref(h)) == mhc.MH_ERR_ARG
    if mhc.device_count() == 0:      # the order-2 extension builds on the device only: no host-only model
        big = np.zeros(1 << 24, dtype=np.uint64)
        assert lib.mh_model_from_counts(big.ctypes.data, 2, ctypes.byref(h)) == mhc.MH_ERR_NO_DEVICE


def test_order2_histogram_workspace_is_plain_arithmetic(mhc):
    """mh_dev_histogram_o2_workspace needs no device: nothing below 32 MiB (tag cache only), then two bytes per input byte
    of a slab (2 GiB at most) plus the bucket images and tables — bounded whatever the input size."""
    lib = mhc.lib()
    assert lib.mh_dev_histogram_o2_workspace(0) == 64 and lib.mh_dev_histogram_o2_workspace((32 << 20) - 1) == 64
    small, big, huge = (lib.mh_dev_histogram_o2_workspace(n) for n in (32 << 20, 2 << 30, 1 << 40))
    assert (64 << 20) < small < (64 << 20) + (40 << 20)
    assert (4 << 30) < big < (4 << 30) + (100 << 20) and huge == big
    assert small % 256 == 0 and big % 256 == 0


def test_compute_refuses_without_gpu(mhc):
    """No CPU fallback: without a device every compute call reports MH_ERR_NO_DEVICE."""
    if mhc.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(mhc.MhError) as e:
        mhc.histogram_o1(b"hello world")
    assert e.value.status == mhc.MH_ERR_NO_DEVICE
    m = mhc.Model.from_counts(np.ones(65536, dtype=np.uint64), 1)
    with pytest.raises(mhc.MhError) as e:
        m.encode(b"hello")
    assert e.value.status == mhc.MH_ERR_NO_DEVICE
    with pytest.raises(mhc.MhError) as e:
        m.decode(b"\x00", 8)
    assert e.value.status == mhc.MH_ERR_NO_DEVICE
    # the device entry points of the two-pass stream decode (round 5): argument errors first, then no device — never a fallback
    lib = mhc.lib()
    buf = (ctypes.c_uint8 * 256)()
    ns = ctypes.c_uint64()
    p = ctypes.addressof(buf)
    assert lib.mh_dev_decode_stream_states(None, p, 64, 0x20, ctypes.byref(ns), p, 256, None) == mhc.MH_ERR_ARG
    assert lib.mh_dev_decode_stream_states(m.handle, p, 1 << 30, 0x20, ctypes.byref(ns), p, 256, None) == mhc.MH_ERR_CAPACITY   # workspace too small
    ws = int(lib.mh_dev_build_index_workspace(64))
    assert lib.mh_dev_decode_stream_states(m.handle, p, 64, 0x20, ctypes.byref(ns), p, ws, None) == mhc.MH_ERR_NO_DEVICE
    assert lib.mh_dev_decode_stream_emit(m.handle, p, 64, 0x20, None, 16, p, ws, None) == mhc.MH_ERR_ARG
    assert lib.mh_dev_decode_variant(None, None) == mhc.MH_ERR_ARG


def test_cli_rejects_bad_chunk_sizes_before_touching_a_device(tmp_path):
    """--chunk is validated at parse time (a zero or non-power-of-two value used to reach a division)."""
    import subprocess
    binp = os.path.join(ROOT, "bin", "markovhuffman")
    if not os.path.exists(binp):
        entry.build()
    src = tmp_path / "in"
    src.write_bytes(b"hello")
    for bad in ("0", "300", "16384", "abc"):
        r = subprocess.run([binp, str(src), "-o", str(tmp_path / "o"), "--index", str(tmp_path / "i"), "--chunk", bad],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1
        assert b"--chunk must be a power of two" in r.stderr


def test_tie_free_counts_host_model_matches_oracle(mhc, oracle):
    """Large pairwise-distinct counts (the shape of a 16 GiB histogram): the host twin of the device tree
    build against the oracle.  (On the GPU the same counts take the sorted two-queue path of
    tree_build_kernel; tests/test_gpu_parity.py compares its images with this host build.)"""
    rng = np.random.default_rng(99)
    w = 1.0 / np.arange(1, 257) ** 1.1
    p = np.outer(w, w).ravel()
    counts = (rng.permutation(65536).astype(np.uint64) + np.floor(p / p.sum() * 2 ** 34).astype(np.uint64) * np.uint64(65536))
    m = mhc.Model.from_counts(counts, 1)
    assert m.table_bytes() == oracle.Model.from_counts(counts, 1).table_bytes()
