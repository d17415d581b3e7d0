"""CPU sanitizer configuration (SURVEY.md 5 "race detection / sanitizers").  The reference strips its 29 asserts with -DNDEBUG
(Makefile:20-21; e.g. src/coding.cpp:72,127-128) and its table loader recurses over untrusted bits with no bounds checks
(src/huffman.cpp:166-172).  Here the HOST code — mh_model.cpp, mh_api*.cpp, host/coding.cpp, host/main.cpp — is built with
AddressSanitizer + UndefinedBehaviorSanitizer (`make -C markov-huffman-coding_amd/csrc asan`, CPU build only: kernels are
not instrumented) and driven three ways, none of which needs a GPU:
  * the table-file mutation fuzz (csrc/sanitize/fuzz_table.cpp): 10 000 truncations / bit flips / splices of the golden
    `.e` / `.eh` files through mh_model_from_table_bits and every accessor of whatever loads;
  * tests/test_abi.py (host tree build, codes, LUT, table files, headers, no-device behaviour) against libmhc_asan.so;
  * the CLI's argument, table-file and error paths (it stops at "no usable HIP device" without a GPU)."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN_DIR, ROOT

CSRC = os.path.join(ROOT, "markov-huffman-coding_amd", "csrc")
HOST = os.path.join(ROOT, "markov-huffman-coding_amd", "host")
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}


@pytest.fixture(scope="module")
def asan_build():
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j4", "all"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "asan"], stderr=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", HOST, "-s", "asan"], stderr=subprocess.DEVNULL)
    rt = subprocess.check_output(["/opt/rocm/bin/hipcc", "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    assert os.path.exists(rt), rt
    return {"fuzz": os.path.join(CSRC, "build", "fuzz_table"), "lib": os.path.join(ROOT, "markov-huffman-coding_amd", "libmhc_asan.so"),
            "cli": os.path.join(ROOT, "bin", "markovhuffman_asan"), "rt": rt}


def clean(proc):
    text = (proc.stdout or "") + (proc.stderr or "")
    assert "AddressSanitizer" not in text and "runtime error" not in text, text[-4000:]
    return text


def test_table_file_mutation_fuzz_is_clean_under_asan_and_ubsan(asan_build):
    seeds = sorted(glob.glob(os.path.join(GOLDEN_DIR, "expected", "*.e")) + glob.glob(os.path.join(GOLDEN_DIR, "expected", "*.eh")))
    assert len(seeds) >= 10
    p = subprocess.run([asan_build["fuzz"], "10000", "4"] + seeds, capture_output=True, text=True, env=dict(os.environ, **SAN_ENV), timeout=900)
    text = clean(p)
    assert p.returncode == 0, text[-2000:]
    assert "10000 cases" in text and " 0 other status" in text, text       # every case: a model, or MH_ERR_BADTABLE
    loaded = int(text.split("cases,")[1].split("loaded")[0])
    assert 500 < loaded < 9500                                              # both outcomes are exercised


def test_abi_suite_is_clean_against_the_sanitized_library(asan_build):
    env = dict(os.environ, **SAN_ENV)
    env.update({"LD_PRELOAD": asan_build["rt"], "MH_LIB": asan_build["lib"]})
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    text = clean(p)
    assert p.returncode == 0, text[-3000:]
    assert " passed" in text


def test_cli_argument_table_and_error_paths_are_clean(asan_build, tmp_path):
    """src/main.cpp:55-115 (argument rules), 137-161 (table type detection) and the error exits, through the sanitized CLI.
    Without a GPU every run ends in an error message and exit code 1 — what is checked is that nothing trips a sanitizer."""
    cli, exp = asan_build["cli"], os.path.join(GOLDEN_DIR, "expected")
    src = tmp_path / "in.txt"
    src.write_bytes(b"hello sanitizer " * 100)
    bad = tmp_path / "bad.e"
    bad.write_bytes(open(os.path.join(exp, "input_ipsum.txt.e"), "rb").read()[:97])           # a truncated table file
    runs = [
        [],                                                                        # help, exit 1 (src/main.cpp:42-45)
        [str(src)],                                                                # no -o
        [str(src), "-o", str(tmp_path / "o.cm"), "-d", str(tmp_path / "o.e")],
        [str(src), "-o", str(tmp_path / "o.ch"), "-h", "-d", str(tmp_path / "o.eh")],
        [str(src), "-x", "-o", str(tmp_path / "o.txt")],                           # -x without -e
        [str(src), "-o", str(tmp_path / "o.cm"), "-e", os.path.join(exp, "input_ipsum.txt.e"), "-d", str(tmp_path / "o.e")],   # -e with -d
        [os.path.join(exp, "input_ipsum.txt.cm"), "-x", "-e", os.path.join(exp, "input_ipsum.txt.e"), "-o", str(tmp_path / "back.txt")],
        [os.path.join(exp, "input_ipsum.txt.cm"), "-xh", "-e", os.path.join(exp, "input_ipsum.txt.e"), "-o", str(tmp_path / "back.txt")],   # wrong type
        [os.path.join(exp, "input_ipsum.txt.ch"), "-xh", "-e", os.path.join(exp, "input_ipsum.txt.eh"), "-o", str(tmp_path / "back.txt")],
        [os.path.join(exp, "input_ipsum.txt.cm"), "-x", "-e", str(bad), "-o", str(tmp_path / "back.txt")],
        [str(src), "-q", "-g", "-o", str(tmp_path / "o.cm")],                       # unknown flag + -g
        [str(tmp_path / "missing.txt"), "-o", str(tmp_path / "o.cm")],
    ]
    for args in runs:
        p = subprocess.run([cli] + args, capture_output=True, text=True, env=dict(os.environ, **SAN_ENV), timeout=120)
        clean(p)
        assert p.returncode in (0, 1), (args, p.returncode, p.stderr[-500:])
