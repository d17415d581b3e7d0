import hashlib
import json
import os
import sys

import pytest

try:                      # torch (bench.py, the sharded tests) brings its own HIP runtime: when libmhc.so is loaded first the
    import torch  # noqa: F401   # process ends up with two runtimes and torch then sees no GPU, so torch is always imported first
except Exception:         # (the codec itself does not need torch)
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def kat_inputs():
    """Formula-defined known-answer inputs (same formulas as tests/golden/make_golden.py)."""
    out = {}
    out["kat1"] = b"aaaabbcd"
    out["kat2"] = bytes((i * i + 7 * i) % 251 for i in range(100000))
    out["kat3"] = bytes(range(256)) * 64
    x = 12345
    buf = bytearray()
    for _ in range(1 << 20):
        x = (x * 1103515245 + 12345) & 0x7FFFFFFF
        buf.append(((x >> 16) & 0xFF) & ((x >> 8) & 0xFF))
    out["kat4"] = bytes(buf)
    out["empty"] = b""
    out["one_Z"] = b"Z"
    out["nine_Z"] = b"Z" * 9
    return out


_GOLDEN = None


def golden():
    """name -> dict(data=bytes, meta=golden.json entry)."""
    global _GOLDEN
    if _GOLDEN is None:
        with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
            meta = json.load(f)
        kats = kat_inputs()
        out = {}
        for name, m in meta.items():
            if m["formula"]:
                data = kats[name]
            else:
                with open(os.path.join(GOLDEN_DIR, "inputs", name), "rb") as f:
                    data = f.read()
            assert hashlib.sha256(data).hexdigest() == m["sha256"], name
            out[name] = {"data": data, "meta": m}
        _GOLDEN = out
    return _GOLDEN


def golden_names():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return sorted(json.load(f).keys())


def expected_file(name, ext):
    """Bytes of the reference's output if committed in full, else None (hash-only entry)."""
    p = os.path.join(GOLDEN_DIR, "expected", name + "." + ext)
    if os.path.exists(p):
        with open(p, "rb") as f:
            return f.read()
    return None


def check_against_golden(name, ext, blob):
    m = golden()[name]["meta"][ext]
    assert len(blob) == m["size"], "%s.%s size %d != %d" % (name, ext, len(blob), m["size"])
    assert hashlib.sha256(blob).hexdigest() == m["sha256"], "%s.%s sha mismatch" % (name, ext)
    full = expected_file(name, ext)
    if full is not None:
        assert blob == full


@pytest.fixture(scope="session")
def oracle():
    from oracle import mh_oracle
    mh_oracle.build()
    return mh_oracle
