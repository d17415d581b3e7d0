"""Drop-in CLI (bin/markovhuffman) on the GPU: the four invocations of the reference's own test
(test/main.py:17-50,68-77 — Huffman encode, Markov encode, both decodes, filecmp), plus byte equality of
every produced file with the golden outputs of the genuine reference, the -g dump, and the error exits."""
import os
import subprocess

import pytest

import __graft_entry__ as entry
from conftest import ROOT, check_against_golden, expected_file, golden, golden_names

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "bin", "markovhuffman")


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(BIN):
        entry.build()
    assert os.path.exists(BIN)


def run(args, **kw):
    return subprocess.run([BIN] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


@pytest.mark.parametrize("name", golden_names())
def test_reference_test_py_round_trip_and_golden_bytes(tmp_path, name):
    data = golden()[name]["data"]
    src = tmp_path / "in"
    src.write_bytes(data)
    p = lambda ext: tmp_path / ("out." + ext)
    # same command shapes as test/main.py (incl. the bare "-" argument of the Markov encode)
    assert run([src, "-o", p("ch"), "-h", "-d", p("eh")]).returncode == 0
    assert run([src, "-o", p("cm"), "-", "-d", p("e")]).returncode == 0
    for ext in ("ch", "eh", "cm", "e"):
        check_against_golden(name, ext, p(ext).read_bytes())
    assert run([p("cm"), "-o", p("dm"), "-x", "-e", p("e")]).returncode == 0
    assert p("dm").read_bytes() == data
    if data:  # the reference itself cannot decode an empty -h table
        assert run([p("ch"), "-o", p("dh"), "-xh", "-e", p("eh")]).returncode == 0
        assert p("dh").read_bytes() == data


def test_compress_with_existing_table_equals_built(tmp_path):
    """`-e table` without -x compresses with the loaded table (src/main.cpp:137-161,208-212)."""
    data = golden()["input_wiki_cpp.txt"]["data"]
    src = tmp_path / "in"
    src.write_bytes(data)
    assert run([src, "-o", tmp_path / "a.cm", "-d", tmp_path / "a.e"]).returncode == 0
    assert run([src, "-o", tmp_path / "b.cm", "-e", tmp_path / "a.e"]).returncode == 0
    assert (tmp_path / "a.cm").read_bytes() == (tmp_path / "b.cm").read_bytes()


def test_index_sidecar_extension(tmp_path):
    data = golden()["input_wiki_cpp.html"]["data"]
    src = tmp_path / "in"
    src.write_bytes(data)
    assert run([src, "-o", tmp_path / "c", "-d", tmp_path / "t", "--index", tmp_path / "c.idx", "--chunk", "512"]).returncode == 0
    check_against_golden("input_wiki_cpp.html", "cm", (tmp_path / "c").read_bytes())   # payload unchanged by the index
    assert run([tmp_path / "c", "-o", tmp_path / "d", "-x", "-e", tmp_path / "t", "--index", tmp_path / "c.idx"]).returncode == 0
    assert (tmp_path / "d").read_bytes() == data


@pytest.mark.parametrize("name", ["input_a.txt", "input_b.txt", "input_ipsum.txt", "one_Z", "nine_Z"])
@pytest.mark.parametrize("mode", ["g", "gh"])
def test_debug_dump_matches_reference(tmp_path, name, mode):
    """-g: print_table + print_tree on stdout (src/main.cpp:186-190), byte for byte."""
    src = tmp_path / "in"
    src.write_bytes(golden()[name]["data"])
    args = [src, "-o", os.devnull, "-g"] + (["-h"] if mode == "gh" else [])
    r = run(args)
    assert r.returncode == 0
    assert r.stdout == expected_file(name, mode)


def test_error_exits(tmp_path):
    data = golden()["input_a.txt"]["data"]
    src = tmp_path / "in"
    src.write_bytes(data)
    r = run([])
    assert r.returncode == 1 and b"markov-huffman <input>" in r.stderr              # src/main.cpp:42-45
    assert run(["-o", tmp_path / "x"]).returncode == 1                              # no input
    assert run([src, "-x", "-o", tmp_path / "x"]).returncode == 1                   # -x without -e
    assert run([src, "-e", "a", "-d", "b"]).returncode == 1                         # -e with -d
    assert run([src, "-o", tmp_path / "c.cm", "-d", tmp_path / "c.e"]).returncode == 0
    assert run([src, "-o", tmp_path / "c.ch", "-h", "-d", tmp_path / "c.eh"]).returncode == 0
    r = run([tmp_path / "c.cm", "-x", "-h", "-e", tmp_path / "c.e", "-o", tmp_path / "x"])   # wrong table kind
    assert r.returncode == 1 and b"Incorrect encoding table" in r.stderr
    r = run([tmp_path / "c.ch", "-x", "-e", tmp_path / "c.e", "-o", tmp_path / "x"])         # stream/table mismatch
    assert r.returncode == 1 and b"does not match" in r.stderr
    bad = tmp_path / "bad"
    bad.write_bytes(b"\x10" + (tmp_path / "c.cm").read_bytes()[1:])
    r = run([bad, "-x", "-e", tmp_path / "c.e", "-o", tmp_path / "x"])
    assert r.returncode == 1 and b"corrupt" in r.stderr
    assert run([tmp_path / "nope", "-o", tmp_path / "x"]).returncode == 1           # missing input
    r = run([src, "-q", "-o", tmp_path / "y"])                                      # unknown flag only warns
    assert r.returncode == 0 and b"Unknown option q" in r.stderr


def test_large_file_in_segments_and_stdout(tmp_path):
    """N2: a file staged through the card in several segments (mapped input and output files, seams
    OR-merged), the same stream through stdout (no mapping possible), and both decode paths."""
    import numpy as np
    from oracle import mh_oracle
    rng = np.random.default_rng(77)
    w = 1.0 / np.arange(1, 257) ** 1.1
    data = rng.choice(256, size=(24 << 20) + 12345, p=w / w.sum()).astype(np.uint8).tobytes()
    src = tmp_path / "in"
    src.write_bytes(data)
    env = dict(os.environ, MH_SEGMENT_BYTES=str(4 << 20))
    assert run([src, "-o", tmp_path / "c", "-d", tmp_path / "t", "--index", tmp_path / "c.idx"], env=env).returncode == 0
    ref, _ = mh_oracle.Model.from_data(data, 1).compress(data)
    assert (tmp_path / "c").read_bytes() == ref
    piped = run([src, "-d", tmp_path / "t2"], env=env)                       # compressed stream on stdout
    assert piped.returncode == 0 and piped.stdout == ref
    assert run([tmp_path / "c", "-o", tmp_path / "d1", "-x", "-e", tmp_path / "t"], env=env).returncode == 0
    assert (tmp_path / "d1").read_bytes() == data
    assert run([tmp_path / "c", "-o", tmp_path / "d2", "-x", "-e", tmp_path / "t", "--index", tmp_path / "c.idx"], env=env).returncode == 0
    assert (tmp_path / "d2").read_bytes() == data
    out = run([tmp_path / "c", "-x", "-e", tmp_path / "t"], env=env)          # decoded bytes on stdout
    assert out.returncode == 0 and out.stdout == data


def test_order2_extension_round_trip_parity_unpinned(tmp_path, oracle):
    """--order2 (contexts of two previous bytes; the reference has no such mode, parity unpinned): the CLI's
    files equal the generalised oracle's, extraction works with and without the index sidecar, and the
    order-1 extractor refuses the stream."""
    data = golden()["input_wiki_cpp.txt"]["data"]
    src = tmp_path / "in"
    src.write_bytes(data)
    c, t, idx = tmp_path / "c2", tmp_path / "t2", tmp_path / "c2.idx"
    assert run([src, "-o", c, "-d", t, "--order2", "--index", idx]).returncode == 0
    o = oracle.Model.from_data(data, 2)
    assert c.read_bytes() == o.compress(data)[0]
    assert t.read_bytes() == o.table_bytes()
    assert run([c, "-o", tmp_path / "d1", "-x", "-e", t, "--order2", "--index", idx]).returncode == 0
    assert (tmp_path / "d1").read_bytes() == data
    assert run([c, "-o", tmp_path / "d2", "-x", "-e", t, "--order2"]).returncode == 0
    assert (tmp_path / "d2").read_bytes() == data
    r = run([c, "-o", tmp_path / "d3", "-x", "-e", t])              # order-1 mode, order-2 files
    assert r.returncode == 1
