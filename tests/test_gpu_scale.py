"""GPU tests at the sizes the north star is quoted on, and of the multi-GPU code path as far as one card
allows (run with -m gpu on an MI355X).

  * config 2 at its real size: input_ipsum.txt tiled to 2^28 bytes, known answer from SURVEY.md §8(d)
  * >= 5 GiB Zipf(1.1): payload windows beyond bit offset 2^32 compared with the oracle (not just a round trip)
  * one 8 GiB shard of config 4 (uniform random): properties the domain gives (nbits = 8n, table 81 921 B)
  * sharded.HipBackend driven by two gloo ranks on the one device: stitch == the oracle's stream
  * packed LDS histogram counters overflowing many times per workgroup
  * index-free decode of a fixed-length-code stream (3-bit codes never re-synchronise off their lattice)
"""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

import __graft_entry__ as entry
from conftest import golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mhc():
    mod = entry.load_package()
    if not os.path.exists(mod.LIB_PATH):
        entry.build()
    mod.lib()
    assert mod.device_count() >= 1, "GPU tests need a device; the codec has no CPU fallback"
    return mod


@pytest.fixture(scope="module")
def bench_mod():
    import bench
    bench.CHUNK = 1024
    return bench


# ------------------------------------------------------------------ config 2 at 2^28 bytes

def test_config2_tiled_ipsum_known_answer(mhc, oracle):
    """BASELINE config 2: 256 MiB of Lorem Ipsum.  SURVEY.md §8(d) measured the genuine reference on
    test/input/input_ipsum.txt tiled to 2^28 bytes: payload 112 260 663 B (+ 1 header byte), table 439 B.
    The GPU stream must have those sizes and the oracle port's bytes."""
    base = golden()["input_ipsum.txt"]["data"]
    n = 1 << 28
    data = (base * (n // len(base) + 1))[:n]
    m = mhc.Model.from_data(data, 1)
    table = m.table_bytes()
    blob, nbits, idx = m.compress(data, chunk_symbols=1024)
    assert len(table) == 439
    assert len(blob) == 112260663 + 1
    o = oracle.Model.from_data(data, 1)
    ref, ref_bits = o.compress(data)
    assert table == o.table_bytes()
    assert nbits == ref_bits
    assert hashlib.sha256(blob).digest() == hashlib.sha256(ref).digest()
    assert m.decompress(blob, index=idx, chunk_symbols=1024, n_symbols=n) == data


# ------------------------------------------------------------------ bit offsets beyond 2^32 against the oracle

def _oracle_bits(oracle_model, lens, window, prev0):
    """Payload bits the reference writes for `window` when the byte before it was prev0."""
    if prev0 == 0x20:
        blob, nbits = oracle_model.compress(window)
        skip = 0
    else:   # the reference always starts at context ' ': prepend the context byte and drop its code again
        blob, nbits = oracle_model.compress(bytes([prev0]) + window)
        skip = int(lens[0x20 * 256 + prev0])
        assert skip > 0
    return np.unpackbits(np.frombuffer(blob[1:], dtype=np.uint8))[skip:nbits]


@pytest.mark.parametrize("gib", [5, 16])
def test_payload_windows_beyond_2_pow_32_bits_match_the_oracle(mhc, oracle, bench_mod, gib):
    """5 GiB, and BASELINE config 3's full 16 GiB, + a ragged tail of device-generated Zipf(1.1), encoded in one call
    (the bench's own calls: region histogram, region encoder with fine index, tile decoder).  For three 64 MiB windows
    (the last one included) the window's start state comes from its index entry, the same input slice is
    encoded by the oracle with the GPU-built table, and the payload bits are compared: this covers the
    64-bit offset arithmetic (scan, seams, index) against the reference, where a round trip alone would
    not notice an encoder and a decoder that are wrong in the same way."""
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = (gib << 30) + 12345
    chunk = 1024
    data = bench_mod.generate("zipf", n, 2, 0, dev)
    codec = bench_mod.Codec(mhc, n, dev)
    codec.histogram(data, 0x20)
    model = codec.build_model()
    codec.encode(model, data, 0x20)
    torch.cuda.synchronize()
    assert codec.lib.mh_dev_status(codec.enc_ws.data_ptr(), codec.stream()) == 0
    nbits = int(codec.nbits[0].item())
    assert nbits > (1 << 34)
    assert codec.lib.mh_dev_status(codec.hist_ws.data_ptr(), codec.stream()) == 0      # the counts add up to n
    om = oracle.Model.from_table(model.table_bytes())
    lens, _ = om.codes()
    W = 64 << 20
    starts = [1 << 30, 3 << 30, ((n - W) // chunk) * chunk]
    if gib == 16:                                # one window past payload bit 2^36 (byte 12 GiB sits at bit ~7.5e10), the last one too
        starts = [1 << 30, 12 << 30, ((n - W) // chunk) * chunk]
        assert (int(codec.index[starts[1] // chunk].item()) & mhc.INDEX_BIT_MASK) > (1 << 36)
    for s in starts:
        e = min(s + W, n)
        entry0 = int(codec.index[s // chunk].item())
        pos, ctx = entry0 & mhc.INDEX_BIT_MASK, (entry0 >> 56) & 0xFF
        window = data[s:e].cpu().numpy().tobytes()
        assert ctx == int(data[s - 1].item())
        ref = _oracle_bits(om, lens, window, ctx)
        end = nbits if e == n else int(codec.index[e // chunk].item()) & mhc.INDEX_BIT_MASK
        assert end - pos == len(ref)
        if s != starts[0]:
            assert pos > (1 << 32)
        raw = codec.payload[pos // 8:(end + 7) // 8].cpu().numpy()
        got = np.unpackbits(raw)[pos % 8:pos % 8 + (end - pos)]
        assert np.array_equal(got, ref), "payload bits differ in window at byte %d" % s
    # and the whole stream still decodes
    codec.decode(model)
    torch.cuda.synchronize()
    assert codec.lib.mh_dev_status(codec.dec_ws.data_ptr(), codec.stream()) == 0
    assert torch.equal(codec.decoded, data)


# ------------------------------------------------------------------ config 4: one 8 GiB uniform shard

def test_config4_one_uniform_shard_properties(mhc, bench_mod):
    """BASELINE config 4 holds 8 GiB of uniform random bytes per GPU.  Properties of that shard: every pair
    count is about n / 65536, so all codes are exactly 8 bits (payload bits = 8 n), the table file is 256 full
    trees = 81 921 bytes, and the shard round-trips."""
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = 8 << 30
    data = bench_mod.generate("uniform", n, 3, 0, dev)
    codec = bench_mod.Codec(mhc, n, dev)
    codec.histogram(data, 0x20)
    assert int(codec.counts.sum().item()) == n
    model = codec.build_model()
    codec.encode(model, data, 0x20)
    codec.decode(model)
    torch.cuda.synchronize()
    assert codec.lib.mh_dev_status(codec.enc_ws.data_ptr(), codec.stream()) == 0
    assert codec.lib.mh_dev_status(codec.dec_ws.data_ptr(), codec.stream()) == 0
    assert int(codec.nbits[0].item()) == 8 * n
    assert model.max_code_len == 8
    assert len(model.table_bytes()) == 81921
    assert torch.equal(codec.decoded, data)


# ------------------------------------------------------------------ sharded.HipBackend, two ranks on one card

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _hip_rank(rank, world, port, data, q):
    """One rank of the sharded compress: the product's own orchestration and its own backend."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import importlib
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)                 # both ranks share the one card of the test box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as e2
        mhc = e2.load_package()
        sharded = importlib.import_module("mhc_amd.sharded")
        lo, hi = sharded.shard_bounds(len(data), world)[rank]
        shard = torch.frombuffer(bytearray(data[lo:hi] + bytes(16)), dtype=torch.uint8)[:hi - lo].cuda()
        be = sharded.HipBackend(mhc, hi - lo, chunk_symbols=256)
        res = sharded.compress_shard(be, shard, data[hi - 1] if hi > lo else 0)     # compress_step: what bench.py times
        back = be.decode(res["model"])
        assert be.statuses() == (0, 0, 0)
        q.put((rank, lo, hi, res["prev0"], res["start_bit"], res["total_bits"], res["nbits"], res["model"].table_bytes(),
               res["payload"].cpu().numpy().tobytes(), back.cpu().numpy().tobytes(), be.paths()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [20_000_000 + 5, 40_000])
def test_hip_backend_two_ranks_stitch_to_the_reference_stream(mhc, oracle, n):
    """SURVEY.md 8(e) with the shipped code: two gloo ranks on the one device drive sharded.compress_shard(HipBackend),
    i.e. sharded.compress_step — the orchestration `bench.py --gpus N` times: region-mode local histograms, the
    all-reduce, identical models, the bit-offset all-gather, pre-shifted shard payloads from the region encoder —
    and sharded.stitch() of what they return IS the stream the reference writes for the whole input; every shard
    also decodes from its own buffers with the tile decoder.  Which kernels ran is asserted by path code."""
    import importlib
    import torch.multiprocessing as mp
    rng = np.random.default_rng(n)
    w = 1.0 / np.arange(1, 257) ** 1.1
    data = rng.choice(256, size=n, p=w / w.sum()).astype(np.uint8).tobytes()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hip_rank, args=(r, world, port, data, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sharded = importlib.import_module("mhc_amd.sharded")
    o = oracle.Model.from_data(data, 1)
    ref, ref_bits = o.compress(data)
    tables = {g[7] for g in got}
    assert tables == {o.table_bytes()}                      # every rank built the same (reference) tables
    assert got[0][5] == ref_bits and sum(g[6] for g in got) == ref_bits
    assert got[1][3] == data[got[1][1] - 1]                 # shard 1 starts in the context of shard 0's last byte
    assert sharded.stitch([(g[4], g[8]) for g in got], ref_bits) == ref[1:]
    for g in got:
        assert g[9] == data[g[1]:g[2]]
        if n >= 16 << 20:                                   # shards of 8 MiB and more: the fast flow, by path code
            # region encoder priced from the histogram (3: its escape variant ran too — 20 MB of Zipf has codes over 12 bits
            # in its rare contexts), tile decoder
            assert g[10] in ((1, 1), (3, 1)), g[10]
        else:
            assert g[10][0] in (1, 2, 3) and g[10][1] == 2, g[10]   # small shards: the chunk decoder (under 8 MiB)


# ------------------------------------------------------------------ histogram counter overflow, many times per workgroup

@pytest.mark.parametrize("with_ws", [True, False])
def test_histogram_guard_bit_fixups_many_per_workgroup(mhc, with_ws):
    """The order-1 histogram packs two 15-bit counters + guard bits per LDS word; a counter that passes
    32767 is credited to HBM by the lane whose add set the guard bit.  64 MiB of zeros then 64 MiB of 'ab'
    give every one of the 256 persistent workgroups ~262144 adds to ONE counter: the guard-bit path runs
    eight times per workgroup and counter.  Counts are known in closed form."""
    import ctypes as C
    z, ab = 64 << 20, 32 << 20
    data = np.concatenate([np.zeros(z, dtype=np.uint8), np.tile(np.frombuffer(b"ab", dtype=np.uint8), ab)])
    exp = np.zeros(65536, dtype=np.uint64)
    exp[0x20 * 256 + 0] = 1
    exp[0] = z - 1
    exp[0 * 256 + ord("a")] = 1
    exp[ord("a") * 256 + ord("b")] = ab
    exp[ord("b") * 256 + ord("a")] = ab - 1
    if with_ws:
        got = mhc.histogram_o1(data)             # host-buffer call: uses the slab workspace from 1 MiB on
    else:
        lib = mhc.lib()
        d = mhc.DeviceBuffer(data.size, init=data)
        dc = mhc.DeviceBuffer(65536 * 8)
        mhc._check(lib.mh_dev_histogram_o1(d.ptr, data.size, 0x20, dc.ptr, None, 0, None), "hist")
        got = dc.download(np.uint64)
    assert np.array_equal(got, exp)


# ------------------------------------------------------------------ index-free decode of fixed-length codes

@pytest.mark.parametrize("symbols,n", [(8, 3 << 20), (32, 1 << 20), (64, 1 << 20)])
def test_index_free_decode_fixed_length_codes(mhc, oracle, symbols, n):
    """A near-uniform alphabet of 8 / 32 / 64 symbols gives 3- / 5- / 6-bit codes in every context.  Such a
    stream never re-synchronises from a start guess that is off the code-length lattice; the index builder
    therefore puts its segment boundaries on multiples of the gcd of the code lengths.  The stream comes
    from the oracle (what the reference writes: no index)."""
    rng = np.random.default_rng(symbols)
    alphabet = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/", dtype=np.uint8)[:symbols]
    data = alphabet[rng.integers(0, symbols, n)].tobytes()
    o = oracle.Model.from_data(data, 1)
    lens, _ = o.codes()
    # every context of the alphabet has log2(symbols)-bit codes; the start context ' ' has one successor
    # (the first byte) and therefore the 1-bit code of src/huffman.cpp:154-162, which shifts the lattice by one bit
    assert set(np.unique(lens[lens > 0])) == {1, int(np.log2(symbols))}
    blob, _ = o.compress(data)
    m = mhc.Model.from_table(o.table_bytes())
    assert m.decompress(blob) == data


def test_index_free_decode_of_random_bytes_under_fixed_length_codes(mhc, oracle):
    """A stream without an index whose model has 8-bit codes in every one of 256 contexts, each context with its
    own code assignment (what ~1 GiB of random bytes gives: below that the counts still differ enough for mixed
    7/8/9-bit codes).  Code boundaries sit on a lattice of 8 bits and two decodes from different contexts only
    merge when they happen to produce the same symbol (1/256 per symbol), so the segment iteration needs about
    ten passes on the right residue class; the first version gave every class five passes and then walked the
    whole payload on one lane (1 GiB: minutes).  64 MiB here: the one-lane walk would take seconds, the
    segment iteration takes a few tens of milliseconds — the time bound tells them apart."""
    import time
    c = np.arange(256, dtype=np.uint64)
    counts = (100000 + ((c[None, :] * 7 + c[:, None] * 13) % 5)).astype(np.uint64).reshape(-1)
    om = oracle.Model.from_counts(counts, 1)
    lens = np.asarray(om.codes()[0]).reshape(256, 256)
    codes = np.asarray(om.codes()[1]).reshape(256, 256)
    assert (lens == 8).all() and (codes[0] != codes[1]).any()          # fixed length, context-dependent assignment
    n = 64 << 20
    data = np.random.default_rng(77).integers(0, 256, n, dtype=np.uint8).tobytes()
    blob, nbits = om.compress(data)
    assert nbits == 8 * n
    m = mhc.Model.from_table(om.table_bytes())
    assert m.decompress(blob[:1 + (1 << 20)]) == data[:1 << 20]          # warm-up (first HIP calls, allocations)
    t0 = time.perf_counter()
    out = m.decompress(blob)
    dt = time.perf_counter() - t0
    assert out == data
    assert mhc.lib().mh_last_index_path() == 1, "not the segment iteration"       # 1: it converged (4 would be the one-lane walk)
    assert dt < 10.0, "index-free decode of 64 MiB took %.1f s" % dt           # (generous: the path code above is the real check)


@pytest.mark.parametrize("kind", ["period3", "two_symbols_crossed", "runs"])
def test_index_free_decode_of_streams_whose_contexts_never_merge(mhc, oracle, kind):
    """Streams without an index on which two decodes from different contexts never agree again: "ABCABC..."
    (every context has one successor) and 0/1 data whose two contexts map the same bit to opposite symbols.
    The segment iteration repairs one segment per pass there; all codes have one length, so the index builder
    composes per-group context maps instead (positions are arithmetic).  "runs" (0...01...12...: each context
    is followed by itself or its successor, and the start context ' ' by a third symbol, so code lengths are
    mixed and positions depend on the history) takes the maps over (context, bit offset) states.  The one-lane
    walk these replace takes ~11-14 s for 64 Mi symbols: the time bound tells them apart."""
    import time
    n = 64 << 20
    if kind == "period3":
        data = np.tile(np.frombuffer(b"ABC", dtype=np.uint8), n // 3 + 1)[:n].tobytes()
        om = oracle.Model.from_data(data, 1)
    elif kind == "runs":
        data = (np.arange(n, dtype=np.int64) // 4096 % 256).astype(np.uint8).tobytes()
        om = oracle.Model.from_data(data, 1)
        assert np.asarray(om.codes()[0]).max() == 2              # mixed lengths: 1 bit, and 2 bits behind ' '
    else:
        counts = np.zeros((256, 256), dtype=np.uint64)
        counts[0x20, 48] = 1
        counts[48, 48], counts[48, 49] = 5, 3
        counts[49, 48], counts[49, 49] = 3, 5
        om = oracle.Model.from_counts(counts.reshape(-1), 1)
        l8, c64 = om.codes()
        c64 = np.asarray(c64).reshape(256, 256)
        assert c64[48, 48] != c64[49, 48]                      # the same symbol, opposite bits in the two contexts
        d = np.random.default_rng(5).integers(0, 2, n, dtype=np.uint8) + 48
        d[0] = 48                                                # the only successor the start context ' ' has in this model
        data = d.tobytes()
    blob, nbits = om.compress(data)
    assert nbits == n or kind == "runs"
    m = mhc.Model.from_table(om.table_bytes())
    if kind != "runs":
        assert m.decompress(blob[:1 + (1 << 17)]) == data[:1 << 20]      # warm-up
    t0 = time.perf_counter()
    out = m.decompress(blob)
    dt = time.perf_counter() - t0
    assert out == data
    # 2: per-group context maps (one code length), 3: per-group state maps (mixed lengths); 4 would be the one-lane walk
    assert mhc.lib().mh_last_index_path() == (3 if kind == "runs" else 2)
    assert dt < 20.0, "index-free decode of 64 Mi symbols took %.1f s" % dt    # (generous: the path code above is the real check)
