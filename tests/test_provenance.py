"""Counter figures on the bench line are tied to the kernel sources they were collected on."""
import importlib
import json
import os
import shutil

import __graft_entry__ as entry


def _prov():
    entry.load_package()
    return importlib.import_module("mhc_amd.provenance")


def test_counters_of_other_sources_are_dropped(tmp_path):
    prov = _prov()
    csrc = tmp_path / "csrc"
    shutil.copytree(prov.CSRC, csrc, ignore=shutil.ignore_patterns("build", "sanitize", "profiles"))
    prof = tmp_path / "profiles"
    prof.mkdir()
    n = 1 << 30
    traffic = prov.stamp({"_counters": "x/traffic.json", "decode_tile_kernel:%d" % n: 123, "hist_o1_kernel:%d" % n: 77}, str(csrc))
    secondary = prov.stamp({"_counters": "x/pmc.json", "decode_tile_kernel:%d" % n: {"valu_frac": 0.4}}, str(csrc))
    json.dump(traffic, open(prof / "traffic.json", "w"))
    json.dump(secondary, open(prof / "secondary.json", "w"))
    t, s, where = prov.counters_for("decode_tile_kernel", n, str(prof), str(csrc))
    assert t == 123 and s == {"valu_frac": 0.4} and "collected on these kernel sources" in where["traffic"]
    # the tile decoder's file changes: its figures go, the histogram's (another file) stay
    with open(csrc / "mh_tile.hip", "a") as f:
        f.write("// changed\n")
    t, s, where = prov.counters_for("decode_tile_kernel", n, str(prof), str(csrc))
    assert t is None and s is None and "dropped" in where["traffic"] and "dropped" in where["secondary"]
    assert prov.counters_for("hist_o1_kernel", n, str(prof), str(csrc))[0] == 77
    # a shared header changes: everything goes
    with open(csrc / "mh_kernels.h", "a") as f:
        f.write("// changed\n")
    assert prov.counters_for("hist_o1_kernel", n, str(prof), str(csrc))[0] is None
    # a file without hashes (round 4's) is never trusted; another size or kernel has no figure at all
    del traffic["_csrc_sha256"]
    json.dump(traffic, open(prof / "traffic.json", "w"))
    shutil.rmtree(csrc)
    shutil.copytree(prov.CSRC, csrc, ignore=shutil.ignore_patterns("build", "sanitize", "profiles"))
    assert prov.counters_for("decode_tile_kernel", n, str(prof), str(csrc))[0] is None
    assert prov.counters_for("decode_tile_kernel", n + 1, str(prof), str(csrc)) == (None, None, {})


def test_every_kernel_file_has_an_owner():
    prov = _prov()
    for k, f in (("mhk::decode_tile_kernel<2, 7, 0, 2, 0, 0, 0>", "mh_tile.hip"), ("enc_region_kernel", "mh_encode.hip"),
                 ("hist_o1_kernel", "mh_hist.hip"), ("hist2_scatter_kernel", "mh_hist2.hip"), ("decode_kernel", "mh_decode.hip"),
                 ("tree_build_kernel", "mh_tree.hip"), ("index_sync_kernel", "mh_index.hip")):
        assert prov.kernel_sources(k)[0] == f
        assert all(os.path.exists(os.path.join(prov.CSRC, x)) for x in prov.kernel_sources(k))


def test_committed_counter_files_carry_hashes():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in ("traffic.json", "secondary.json"):
        blob = json.load(open(os.path.join(root, "profiles", f)))
        assert isinstance(blob.get("_csrc_sha256"), dict) and blob["_csrc_sha256"], f
