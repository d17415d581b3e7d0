#!/usr/bin/env python3
"""Regenerate tests/golden/ with the GENUINE reference binary (oracle/_ref/markovhuffman).

Runs only in the build container (needs /root/reference to build oracle/_ref; see oracle/Makefile).
What is committed is DATA: the inputs (the reference's own test/input files and formula-defined
known-answer inputs) and the reference's outputs on them:
    X.cm / X.e   Markov stream / table      (markovhuffman X -o X.cm -d X.e)
    X.ch / X.eh  Huffman stream / table     (markovhuffman X -o X.ch -h -d X.eh)
plus golden.json with sizes and sha256 of everything.  Both decodes are run and compared with the
input before anything is written (the round trip of test/main.py:17-50,77).
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "markovhuffman")
REF_INPUTS = "/root/reference/test/input"


def kat_inputs():
    """Formula-defined inputs; regenerable anywhere without the reference."""
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


def sha(b):
    return hashlib.sha256(b).hexdigest()


def run(args):
    subprocess.run([REF_BIN] + args, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main():
    if not os.path.exists(REF_BIN):
        sys.exit("oracle/_ref/markovhuffman missing: run `make -C oracle ref` in the build container")
    inputs = {}
    for name in sorted(os.listdir(REF_INPUTS)):
        with open(os.path.join(REF_INPUTS, name), "rb") as f:
            inputs[name] = f.read()
    inputs.update(kat_inputs())

    in_dir = os.path.join(HERE, "inputs")
    exp_dir = os.path.join(HERE, "expected")
    for d in (in_dir, exp_dir):
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)

    meta = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, data in inputs.items():
            src = os.path.join(tmp, name)
            with open(src, "wb") as f:
                f.write(data)
            p = lambda ext: os.path.join(tmp, name + ext)
            run([src, "-o", p(".cm"), "-d", p(".e")])
            run([src, "-o", p(".ch"), "-h", "-d", p(".eh")])
            entry = {"size": len(data), "sha256": sha(data), "formula": name in kat_inputs()}
            # round trips (Huffman decode of an empty table crashes in the reference: skip it)
            run([p(".cm"), "-o", p(".dm"), "-x", "-e", p(".e")])
            assert open(p(".dm"), "rb").read() == data, name
            if len(data):
                run([p(".ch"), "-o", p(".dh"), "-xh", "-e", p(".eh")])
                assert open(p(".dh"), "rb").read() == data, name
            for ext in (".cm", ".e", ".ch", ".eh"):
                b = open(p(ext), "rb").read()
                entry[ext[1:]] = {"size": len(b), "sha256": sha(b)}
                # large formula-defined cases are pinned by hash only (regenerable); the rest in full
                if not (entry["formula"] and len(data) > 1000):
                    with open(os.path.join(exp_dir, name + ext), "wb") as f:
                        f.write(b)
            # -g debug dumps (stdout of print_table + print_tree, src/main.cpp:186-190) for small cases
            if name in ("input_a.txt", "input_b.txt", "input_ipsum.txt", "one_Z", "nine_Z"):
                for ext, extra in ((".g", []), (".gh", ["-h"])):
                    r = subprocess.run([REF_BIN, src, "-o", os.devnull, "-g"] + extra, check=True,
                                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                    entry[ext[1:]] = {"size": len(r.stdout), "sha256": sha(r.stdout)}
                    with open(os.path.join(exp_dir, name + ext), "wb") as f:
                        f.write(r.stdout)
            # the reference's own inputs are committed as data; formula inputs are regenerated
            if not entry["formula"]:
                with open(os.path.join(in_dir, name), "wb") as f:
                    f.write(data)
            meta[name] = entry
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(meta), "golden entries")


if __name__ == "__main__":
    main()
