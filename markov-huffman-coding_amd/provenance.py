"""Which kernel source a committed counter figure belongs to.

`profiles/traffic.json` / `profiles/secondary.json` hold rocprofv3 --pmc figures collected in their own runs; bench.py only
replays them.  Each file records, per kernel, the sha256 of the source files that kernel is compiled from at the time of the
counter run (`_csrc_sha256`), and bench.py drops a figure whose kernel's sources have changed since: a stale counter can not
reach a bench line.  tools/make_traffic_json.py and tools/make_secondary_json.py write the hashes.
"""
import hashlib
import json
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# headers every kernel file includes
COMMON = ("mh_kernels.h", "mh_dev.hpp", "mh_decode_dev.hpp", "mh_model.hpp")
# kernel name prefix -> its .hip file (first match wins)
KERNEL_FILES = (
    ("decode_tile_kernel", "mh_tile.hip"), ("index_tile", "mh_tile.hip"), ("segment_decode", "mh_tile.hip"),
    ("hist2_", "mh_hist2.hip"), ("hist_o2", "mh_hist2.hip"), ("hist_", "mh_hist.hip"),
    ("enc", "mh_encode.hip"), ("region_", "mh_encode.hip"), ("scan_", "mh_encode.hip"),
    ("decode", "mh_decode.hip"), ("index_", "mh_index.hip"), ("tree_", "mh_tree.hip"), ("o2_hot", "mh_tree.hip"),
)


def kernel_sources(kernel, csrc=CSRC):
    """Source files (relative to csrc/) the named kernel is compiled from."""
    name = kernel.split("::")[-1]
    for prefix, f in KERNEL_FILES:
        if name.startswith(prefix):
            return (f,) + COMMON
    return tuple(sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".h", ".cpp"))))   # unknown: everything


def sources_hash(files, csrc=CSRC):
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(f.encode() + b"\0")
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()


def kernel_hash(kernel, csrc=CSRC):
    return sources_hash(kernel_sources(kernel, csrc), csrc)


def stamp(blob, csrc=CSRC):
    """Adds `_csrc_sha256` = {kernel: hash of its sources now} for every `kernel:size` key of a counters file."""
    kernels = sorted({k.split(":")[0] for k in blob if not k.startswith("_")})
    blob["_csrc_sha256"] = {k: kernel_hash(k, csrc) for k in kernels}
    return blob


def counters_for(kernel, n, profiles_dir, csrc=CSRC):
    """(traffic, secondary, counters_from) of `kernel` at size n from the committed counter files, each None when the file
    has no figure for it OR when the kernel's sources differ from the ones the figure was collected on."""
    traffic, secondary, where = None, None, {}
    for fname, key in (("traffic.json", "traffic"), ("secondary.json", "secondary")):
        path = os.path.join(profiles_dir, fname)
        v = None
        try:
            blob = json.load(open(path))
            v = blob.get("%s:%d" % (kernel, n))
            if v is not None:
                then = (blob.get("_csrc_sha256") or {}).get(kernel)
                if then is None or then != kernel_hash(kernel, csrc):
                    where[key] = ("profiles/%s has a figure for %s from %s, but the kernel's sources have changed since it was "
                                  "collected (or the file carries no source hash): dropped" % (fname, kernel, blob.get("_counters")))
                    v = None
                else:
                    where[key] = ("profiles/%s <- %s (rocprofv3 --pmc, committed; collected on these kernel sources, sha256 %s; "
                                  "not collected in this run)" % (fname, blob.get("_counters"), then[:12]))
        except (OSError, ValueError):
            v = None
        if key == "traffic":
            traffic = v
        else:
            secondary = v
    return traffic, secondary, where
