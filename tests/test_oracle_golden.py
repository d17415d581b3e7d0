"""Pins the CPU oracle (oracle/mh_oracle.c) to the genuine reference.

(1) golden vectors made with the compiled reference (tests/golden/make_golden.py): stream and table
    bytes for Markov (.cm/.e) and plain Huffman (.ch/.eh) on the reference's own test/input files and
    on formula-defined known-answer inputs, incl. the edge cases SURVEY.md §8(c) lists;
(2) when oracle/_ref/markovhuffman exists (build container, or shipped to the GPU box), a live
    differential test on seeded random inputs.
These are CPU tests (-m "not gpu").
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import check_against_golden, golden, golden_names


@pytest.mark.parametrize("name", golden_names())
def test_markov_stream_and_table_match_reference(oracle, name):
    data = golden()[name]["data"]
    m = oracle.Model.from_data(data, order=1)
    blob, nbits = m.compress(data)
    check_against_golden(name, "cm", blob)
    check_against_golden(name, "e", m.table_bytes())
    assert len(blob) == 1 + (nbits + 7) // 8
    assert m.decompress(blob) == data


@pytest.mark.parametrize("name", golden_names())
def test_huffman_stream_and_table_match_reference(oracle, name):
    data = golden()[name]["data"]
    m = oracle.Model.from_data(data, order=0)
    blob, _ = m.compress(data)
    check_against_golden(name, "ch", blob)
    check_against_golden(name, "eh", m.table_bytes())
    if len(data):  # the reference itself crashes decoding with an empty -h table
        assert m.decompress(blob) == data


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("order", [0, 1])
def test_loaded_table_equals_built_table(oracle, name, order):
    """src/main.cpp:137-161 path: a table re-read from its file gives the same codes and stream."""
    data = golden()[name]["data"]
    if order == 0 and not data:
        pytest.skip("empty Huffman table file has no tree")
    built = oracle.Model.from_data(data, order=order)
    loaded = oracle.Model.from_table(built.table_bytes())
    assert loaded.type == order
    lb, cb = built.codes()
    ll, cl = loaded.codes()
    assert np.array_equal(lb, ll) and np.array_equal(cb, cl)
    assert loaded.compress(data)[0] == built.compress(data)[0]
    assert loaded.table_bytes() == built.table_bytes()


def test_known_small_vectors(oracle):
    """Bytes quoted in SURVEY.md §8(c)."""
    m = oracle.Model.from_data(b"aaaabbcd", 1)
    assert m.compress(b"aaaabbcd")[0] == bytes([0x30, 0xF3])
    h = oracle.Model.from_data(b"aaaabbcd", 0)
    assert h.compress(b"aaaabbcd")[0] == bytes([0x3A, 0x0A, 0xDC])
    assert h.table_bytes() == bytes([0x58, 0x56, 0x25, 0x8E, 0xC8])
    m = oracle.Model.from_data(b"aaaabbcdcb", 1)
    assert m.compress(b"aaaabbcdcb")[0] == bytes([0x36, 0xF3, 0x80])
    assert oracle.Model.from_data(b"", 1).compress(b"")[0] == b"\x30"
    assert len(oracle.Model.from_data(b"", 1).table_bytes()) == 33
    assert oracle.Model.from_data(b"Z", 1).compress(b"Z")[0] == bytes([0x37, 0x80])
    assert oracle.Model.from_data(b"Z" * 9, 0).compress(b"Z" * 9)[0] == bytes([0x3F, 0xFF, 0x80])


def test_header_and_type_errors(oracle):
    data = golden()["input_a.txt"]["data"]
    m = oracle.Model.from_data(data, 1)
    h = oracle.Model.from_data(data, 0)
    blob, _ = m.compress(data)
    with pytest.raises(ValueError):
        h.decompress(blob)  # type mismatch, src/coding.cpp:107-110
    with pytest.raises(ValueError):
        m.decompress(b"\x10" + blob[1:])  # bad magic, src/coding.cpp:103-106


def test_lut_shape(oracle):
    """Appendix A.2 item 6: non-empty context fills all 256 entries; leaf depth<=8 or internal at 8."""
    data = golden()["kat4"]["data"]
    m = oracle.Model.from_data(data, 1)
    lens, _ = m.codes()
    saw_internal = False
    for prev in (0, 1, 3, 255):
        if not lens[prev * 256:(prev + 1) * 256].any():
            continue
        for w in range(256):
            present, internal, value, depth = m.lut(prev, w)
            assert present
            if internal:
                saw_internal = True
                assert depth == 8
            else:
                assert 1 <= depth <= 8 and lens[prev * 256 + value] == depth
    assert saw_internal  # kat4 has codes up to 15 bits


def _ref_bin(oracle):
    return oracle.REF_BIN if os.path.exists(oracle.REF_BIN) else None


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("order", [0, 1])
def test_live_differential_vs_reference_binary(oracle, tmp_path, seed, order):
    ref = _ref_bin(oracle)
    if ref is None:
        pytest.skip("oracle/_ref/markovhuffman not built (needs /root/reference)")
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1000, 200000))
    kind = seed % 3
    if kind == 0:
        data = rng.integers(0, 256, n, dtype=np.uint8)
    elif kind == 1:
        data = np.minimum(rng.geometric(0.08, n) - 1, 255).astype(np.uint8)
    else:
        w = 1.0 / np.arange(1, 257) ** 1.1
        data = rng.choice(256, size=n, p=w / w.sum()).astype(np.uint8)
    data = data.tobytes()
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    args = [ref, str(src), "-o", str(tmp_path / "out.c"), "-d", str(tmp_path / "out.e")]
    if order == 0:
        args.insert(2, "-h")
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    m = oracle.Model.from_data(data, order)
    assert m.compress(data)[0] == (tmp_path / "out.c").read_bytes()
    assert m.table_bytes() == (tmp_path / "out.e").read_bytes()
    # and the oracle decodes the reference's stream with the reference's table
    loaded = oracle.Model.from_table((tmp_path / "out.e").read_bytes())
    assert loaded.decompress((tmp_path / "out.c").read_bytes()) == data


def test_order2_generalisation_self_consistency_parity_unpinned(oracle):
    """The order-2 section of the oracle is a generalisation the reference does not have (parity
    unpinned).  What CAN be checked on the CPU: it round-trips, its table file reloads to the same
    codes, and restricted to data whose second context byte never matters it reproduces order 1."""
    data = golden()["input_ipsum.txt"]["data"]
    m = oracle.Model.from_data(data, 2)
    blob, nbits = m.compress(data)
    assert blob[0] & 0xF8 == 0x40
    assert m.decompress(blob) == data
    t = m.table_bytes()
    assert t[:33] == b"\x80" + bytes(32) and t[33:37] == b"MH2\x01"
    m2 = oracle.Model.from_table(t)
    assert m2.type == 2 and m2.compress(data)[0] == blob
    assert oracle.Model.from_table(t[:33]).type == 1               # the prefix alone is the reference's empty Markov table
    # kat3 = bytes(range(256)) * 64: every byte determines its successor, so order 1 and order 2 both
    # give 1-bit codes everywhere and the payloads have the same length
    k3 = golden()["kat3"]["data"]
    assert oracle.Model.from_data(k3, 2).compress(k3)[1] == oracle.Model.from_data(k3, 1).compress(k3)[1]


def test_order2_table_file_loads_on_the_host_side_of_the_library(oracle):
    """The library parses an order-2 table file without a device (type, write-back)."""
    import __graft_entry__ as entry
    entry.build()
    mhc = entry.load_package()
    data = golden()["input_wiki_cpp.txt"]["data"]
    t = oracle.Model.from_data(data, 2).table_bytes()
    m = mhc.Model.from_table(t)
    assert m.type == 2
    assert m.table_bytes() == t
