"""The coding.h face as a drop-in for the reference's own caller.

oracle/Makefile (`make dropin`, part of __graft_entry__.build() when /root/reference is present) compiles
the reference's src/main.cpp WHERE IT LIES against markov-huffman-coding_amd/host/ (its quoted includes
"bitbuffer.h", "coding.h", "huffman.h", "markov_huffman.h", "utils.h" resolve to the product's headers)
and links it with host/coding.cpp + libmhc.so into oracle/_ref/markovhuffman_refmain.  Nothing of the
reference is copied into the repo; the binary is a built artefact like oracle/_ref/markovhuffman.

CPU: the compile and link succeed and the binary starts (argc < 2 -> help text, exit 1, src/main.cpp:42-45).
GPU: the reference's main() drives the HIP path: its outputs equal the golden files of the genuine reference.
"""
import os
import subprocess

import pytest

from conftest import ROOT, check_against_golden, golden, golden_names

REF_MAIN = "/root/reference/src/main.cpp"
BIN = os.path.join(ROOT, "oracle", "_ref", "markovhuffman_refmain")


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference checkout not present (GPU box)")
def test_reference_main_compiles_and_links_against_the_host_face():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "markov-huffman-coding_amd", "csrc"), "-s"])
    if os.path.exists(BIN):
        os.remove(BIN)                       # force the recipe to run: this IS the test
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "dropin"])
    assert os.path.exists(BIN)
    r = subprocess.run([BIN], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1
    assert b"markov-huffman <input> [-o output] [options]" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names())
def test_reference_main_runs_on_the_hip_path(tmp_path, name):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/markovhuffman_refmain not built (needs /root/reference at build time)")
    data = golden()[name]["data"]
    src = tmp_path / "in"
    src.write_bytes(data)
    run = lambda a: subprocess.run([BIN] + [str(x) for x in a], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    p = lambda ext: tmp_path / ("out." + ext)
    assert run([src, "-o", p("cm"), "-d", p("e")]).returncode == 0
    assert run([src, "-o", p("ch"), "-h", "-d", p("eh")]).returncode == 0
    for ext in ("cm", "e", "ch", "eh"):
        check_against_golden(name, ext, p(ext).read_bytes())
    assert run([p("cm"), "-o", p("dm"), "-x", "-e", p("e")]).returncode == 0
    assert p("dm").read_bytes() == data
