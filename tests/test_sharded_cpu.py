"""world_size-2 (and 3) gloo tests of the multi-GPU orchestration (markov-huffman-coding_amd/sharded.py)
on CPU.  The per-shard compute is a TEST DOUBLE backed by the oracle (tests may use the oracle; the
product's only backend is HipBackend) — what is under test is the sharding itself: shard bounds, the
context hand-off between shards, the histogram all-reduce, identical models on every rank, the
bit-offset all-gather, and that the shard payloads concatenated bit for bit ARE the reference stream of
the whole input."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """Test double: same duck type as sharded.HipBackend, computed by oracle/mh_oracle.c."""

    def __init__(self, oracle):
        self.o = oracle

    def length(self, shard):
        return len(shard)

    def histogram(self, shard, prev0):
        return torch.from_numpy(self.o.histogram_o1(shard, prev0).astype(np.int64))

    def build_model(self, counts):
        return self.o.Model.from_counts(counts.numpy().astype(np.uint64), 1)

    def merge(self, local, group=None):
        import importlib
        sharded = importlib.import_module("mhc_amd.sharded")
        return sharded.merged_histogram(local.clone(), group)

    def payload_bits(self, model, counts):
        lens, _ = model.codes()
        return torch.tensor([int((counts.numpy().astype(np.int64) * np.asarray(lens, dtype=np.int64)).sum())], dtype=torch.int64)

    def encode(self, model, shard, prev0, start_bit=None):
        # the oracle's compress starts at context ' '; emulate an arbitrary first context by
        # prepending that byte and dropping its code afterwards
        start_bit = 0 if start_bit is None else int(start_bit.item())
        lens, _ = model.codes()
        if prev0 == 0x20:
            blob, nbits = model.compress(shard)
            bits = np.unpackbits(np.frombuffer(blob[1:], dtype=np.uint8))[:nbits]
        else:
            blob, nbits = model.compress(bytes([prev0]) + shard)
            skip = int(lens[0x20 * 256 + prev0])
            bits = np.unpackbits(np.frombuffer(blob[1:], dtype=np.uint8))[skip:nbits]
        lead = np.zeros(start_bit & 7, dtype=np.uint8)            # pre-shifted like mh_dev_encode_at
        shifted = np.concatenate([lead, bits])
        return np.packbits(shifted).tobytes(), len(shifted), None

    def finish(self, encoded):
        return encoded


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, data, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as entry
        from oracle import mh_oracle
        mhc = entry.load_package()
        import importlib
        sharded = importlib.import_module("mhc_amd.sharded")
        lo, hi = sharded.shard_bounds(len(data), world)[rank]
        shard = data[lo:hi]
        res = sharded.compress_shard(OracleBackend(mh_oracle), shard, shard[-1] if shard else 0)
        q.put((rank, lo, hi, res["prev0"], res["start_bit"], res["total_bits"], res["nbits"],
               res["model"].table_bytes(), res["payload"]))
    finally:
        dist.destroy_process_group()


def _run(world, data):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, data, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return sorted(out)


@pytest.mark.parametrize("world,n", [(2, 100000), (2, 33), (3, 50001), (2, 0), (2, 10), (8, 200003)])   # 8: the node the scaling bench runs on
def test_sharded_compress_equals_whole_stream(oracle, world, n):
    rng = np.random.default_rng(n + world)
    w = 1.0 / np.arange(1, 257) ** 1.1
    data = rng.choice(256, size=n, p=w / w.sum()).astype(np.uint8).tobytes()
    res = _run(world, data)
    whole = oracle.Model.from_data(data, 1)
    ref_blob, ref_bits = whole.compress(data)
    ref_bitarr = np.unpackbits(np.frombuffer(ref_blob[1:], dtype=np.uint8))[:ref_bits]
    pos = 0
    parts = []
    for rank, lo, hi, prev0, start, total, nbits, table, packed in res:
        assert table == whole.table_bytes()                      # identical model on every rank
        assert prev0 == (0x20 if lo == 0 or n == 0 else data[lo - 1]) or lo == hi
        assert start == pos and total == ref_bits                # placement known before encoding
        assert len(packed) == ((start & 7) + nbits + 7) // 8
        parts.append((start, packed))
        pos += nbits
    assert pos == ref_bits
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    entry.load_package()
    import importlib
    sharded = importlib.import_module("mhc_amd.sharded")
    assert sharded.stitch(parts, ref_bits) == ref_blob[1:]       # pre-shifted shards OR together into THE stream
    del ref_bitarr
    bounds = [(lo, hi) for _, lo, hi, *_ in res]
    assert bounds[0][0] == 0 and bounds[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    assert all(lo % 16 == 0 for lo, _ in bounds)                 # device alignment requirement


def test_shard_bounds_cover_and_align():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    entry.load_package()
    import importlib
    sharded = importlib.import_module("mhc_amd.sharded")
    for n in (0, 1, 15, 16, 17, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            b = sharded.shard_bounds(n, world)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert all(lo % 16 == 0 for lo, _ in b)


# ------------------------------------------------------------------ order 2: only the live contexts' rows travel

def _compact_worker(rank, world, port, shards, dense_above, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as entry
        entry.load_package()
        import importlib
        sharded = importlib.import_module("mhc_amd.sharded")
        data, ctx0 = shards[rank]
        # the shard's own order-2 counts, its first symbols in the context of the two bytes before the shard
        full = np.frombuffer(bytes([ctx0 >> 8, ctx0 & 255]) + data, dtype=np.uint8).astype(np.int64)
        keys = (full[:-2] << 16) | (full[1:-1] << 8) | full[2:]
        counts = torch.from_numpy(np.bincount(keys, minlength=1 << 24).astype(np.int64))
        nlive = sharded.merged_histogram_o2_compact(counts, dense_above=dense_above)
        q.put((rank, nlive, counts.numpy().tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,dense_above", [(2, "text", 0.25), (3, "text", 0.25), (2, "flat", 0.25), (2, "text", 0.0), (2, "empty", 0.25)])
def test_order2_compact_merge_equals_the_histogram_of_the_whole_input_parity_unpinned(oracle, world, kind, dense_above):
    """merged_histogram_o2_compact over gloo ranks on the CPU: the counts every rank ends with are the order-2 histogram of
    the whole input, whether the compact path ran (text: a few hundred live contexts), the dense fallback (flat bytes, or a
    threshold of zero), or nothing was live at all."""
    rng = np.random.default_rng(world)
    if kind == "text":
        words = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta"]
        data = b" ".join(words[int(i)] for i in rng.integers(0, len(words), size=6000))
    elif kind == "flat":
        data = rng.integers(0, 256, size=40000, dtype=np.uint8).tobytes()
    else:
        data = b""
    cuts = [len(data) * r // world // 16 * 16 for r in range(world)] + [len(data)]
    shards = []
    for r in range(world):
        lo, hi = cuts[r], cuts[r + 1]
        ctx0 = 0x2020 if lo == 0 else (data[lo - 2] << 8 | data[lo - 1])
        shards.append((data[lo:hi], ctx0))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_compact_worker, args=(r, world, port, shards, dense_above, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = oracle.histogram_o2(data).astype(np.int64)
    live = int((want.reshape(65536, 256) != 0).any(axis=1).sum())
    for rank, nlive, raw in out:
        assert np.array_equal(np.frombuffer(raw, dtype=np.int64), want), rank
        assert nlive == live


def test_bench_self_launch_starts_the_ranks_as_a_child_and_relays_them(tmp_path):
    """`python bench.py --gpus 2` without a launcher: bench.self_launch builds a torch.distributed.run command of this very
    script (127.0.0.1, a free port, N ranks) as a CHILD process and relays its output and exit code.  No GPU here: the
    child is replaced by a stand-in that prints what it was started with."""
    import importlib.util
    import json
    import subprocess
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class FakeChild:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(['{"metric": "x", "n_gpus": 2}\n'])

        def wait(self):
            return 7

    real = subprocess.Popen
    subprocess.Popen = FakeChild
    old_argv = sys.argv
    sys.argv = ["bench.py", "--gpus", "2", "--size", "1048576", "--backend", "gloo"]
    try:
        rc = bench.self_launch(2)
    finally:
        subprocess.Popen = real
        sys.argv = old_argv
    assert rc == 7                                             # the child's exit code comes back
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "2", "--size", "1048576", "--backend", "gloo"]
    assert os.path.samefile(cmd[-7], os.path.join(ROOT, "bench.py"))
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
