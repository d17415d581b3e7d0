"""ORDER 2 (context = the previous two bytes; SURVEY.md §8(f) N4, BASELINE config 5) on the GPU.

PARITY UNPINNED: the reference implements order 1 only (README.md:158-166 speculates about higher orders),
so there is no reference output to compare with.  The spec is the generalised oracle (oracle/mh_oracle.h,
order-2 section: the reference's per-context algorithm applied to 65536 two-byte contexts); these tests
show GPU == that oracle bit for bit (counts, table file, every codeword, the stream) plus round trips."""
import numpy as np
import pytest

import __graft_entry__ as entry
from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mhc():
    mod = entry.load_package()
    mod.lib()
    assert mod.device_count() >= 1
    return mod


def text_like(n, seed):
    rng = np.random.default_rng(seed)
    words = [b"lorem", b"ipsum", b"dolor", b"sit", b"amet", b"consectetur", b"adipiscing", b"elit", b"sed", b"do",
             b"eiusmod", b"tempor", b"incididunt", b"ut", b"labore", b"et", b"dolore", b"magna", b"aliqua"]
    out = bytearray()
    while len(out) < n:
        out += words[int(rng.integers(len(words)))] + (b". " if rng.random() < 0.1 else b" ")
    return bytes(out[:n])


def zipf_bytes(n, seed, s=1.1, k=256):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, k + 1) ** s
    return rng.choice(k, size=n, p=w / w.sum()).astype(np.uint8).tobytes()


CASES = {
    "ipsum": lambda: golden()["input_ipsum.txt"]["data"],
    "wiki_html": lambda: golden()["input_wiki_cpp.html"]["data"],
    "kat1": lambda: golden()["kat1"]["data"],
    "kat3": lambda: golden()["kat3"]["data"],
    "empty": lambda: b"",
    "one_Z": lambda: b"Z",
    "text_1m": lambda: text_like((1 << 20) + 7, 3),
    "zipf32_512k": lambda: zipf_bytes(1 << 19, 5, k=32),
    "zipf256_2m": lambda: zipf_bytes((1 << 21) + 3, 6),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_order2_histogram_parity_unpinned(mhc, oracle, name):
    data = CASES[name]()
    assert np.array_equal(mhc.histogram_o2(data), oracle.histogram_o2(data))


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 33, 4097, 65536 + 5])
def test_order2_histogram_ragged_sizes_parity_unpinned(mhc, oracle, n):
    data = zipf_bytes(n, n, k=16)
    assert np.array_equal(mhc.histogram_o2(data), oracle.histogram_o2(data))


def test_order2_histogram_repeated_pair_parity_unpinned(mhc):
    """A run of one repeated byte: every lane of every workgroup lands in ONE slot of the LDS counter cache."""
    n = 8 << 20
    data = b"\x00" * n
    h = mhc.histogram_o2(data)
    exp = np.zeros(1 << 24, dtype=np.uint64)
    exp[0x202000] = 1
    exp[0x200000] = 1
    exp[0] = n - 2
    assert np.array_equal(h, exp)


@pytest.mark.parametrize("name", sorted(CASES))
def test_order2_tables_stream_and_round_trip_parity_unpinned(mhc, oracle, name):
    data = CASES[name]()
    m = mhc.Model.from_data(data, 2)
    o = oracle.Model.from_data(data, 2)
    assert m.type == 2
    table = m.table_bytes()
    assert table == o.table_bytes()
    lg, cg = m.codes_o2()
    lo, co = o.codes_o2()
    assert np.array_equal(lg, lo) and np.array_equal(cg, co)
    blob, nbits, idx = m.compress(data, chunk_symbols=256)
    ref, ref_bits = o.compress(data)
    assert (nbits, blob) == (ref_bits, ref)
    assert blob[0] & 0xF8 == 0x40                                 # the extension's own magic nibble
    assert m.decompress(blob, index=idx, chunk_symbols=256, n_symbols=len(data)) == data
    assert o.decompress(blob) == data
    # a model loaded from the table file (tables derived on the host) against the device-built one: same
    # codewords, same first-level decode table and layout, same stream, and it decodes.  (The walk tree and
    # the node ids in second-level inner entries number the nodes in file order there, creation order here.)
    t = mhc.Model.from_table(table)
    assert t.type == 2 and t.table_bytes() == table
    for which in (1, 3, 4, 6):
        assert t.image(which) == m.image(which), "image %d differs" % which
    assert len(t.image(5)) == len(m.image(5))
    assert t.compress(data)[0] == blob
    assert t.decompress(blob, index=idx, chunk_symbols=256, n_symbols=len(data)) == data


@pytest.mark.parametrize("name", ["ipsum", "kat1", "one_Z", "zipf32_512k"])
def test_order2_decode_without_index_parity_unpinned(mhc, oracle, name):
    """No sidecar: the index builder's segment iteration with two-byte contexts (small streams: few segments)."""
    data = CASES[name]()
    o = oracle.Model.from_data(data, 2)
    blob, _ = o.compress(data)
    m = mhc.Model.from_table(o.table_bytes())
    assert m.decompress(blob) == data


def test_order2_stream_and_table_types_are_kept_apart(mhc, oracle):
    data = CASES["ipsum"]()
    m2 = mhc.Model.from_data(data, 2)
    m1 = mhc.Model.from_data(data, 1)
    b2, _, _ = m2.compress(data)
    b1, _, _ = m1.compress(data)
    with pytest.raises(mhc.MhError) as e:
        m1.decompress(b2)
    assert e.value.status == mhc.MH_ERR_TYPE
    with pytest.raises(mhc.MhError) as e:
        m2.decompress(b1)
    assert e.value.status == mhc.MH_ERR_TYPE
    # the reference's loader (here: the order-1 oracle path) takes an order-2 table file for an empty order-1 table
    assert len(b2) < len(b1)                                      # and order 2 does compress this text better


def test_order2_large_text_round_trip_parity_unpinned(mhc, oracle):
    n = (32 << 20) + 11
    data = text_like(n, 9)
    m = mhc.Model.from_data(data, 2)
    blob, nbits, idx = m.compress(data, chunk_symbols=1024)
    ref, ref_bits = oracle.Model.from_data(data, 2).compress(data)
    assert nbits == ref_bits and blob == ref
    assert m.decompress(blob, index=idx, chunk_symbols=1024, n_symbols=n) == data
    # and without the sidecar: the segment iteration of the order-1 index builder with two-byte contexts (the one-lane
    # walk it replaced takes ~10 s for these 32 Mi symbols)
    import time
    t0 = time.perf_counter()
    assert m.decompress(blob) == data
    dt = time.perf_counter() - t0
    assert mhc.lib().mh_last_index_path() == 1, "not the segment iteration"       # 4 would be the one-lane walk
    assert dt < 15.0, "order-2 decode without an index took %.1f s" % dt


def test_order2_codes_longer_than_the_packed_entry_parity_unpinned(mhc, oracle):
    """Fibonacci-weighted successors in every context: code lengths up to ~60 bits.  The encoder's packed
    `length << 56 | code` entries hold codes of at most 56 bits; longer ones carry the escape length and are
    fetched from the two full tables (and the decoder walks the tree for them)."""
    fib = [1, 1]
    while len(fib) < 60:
        fib.append(fib[-1] + fib[-2])
    row = np.zeros(256, dtype=np.uint64)
    row[:60] = np.array(fib, dtype=np.uint64)
    counts = np.tile(row, 65536)
    m = mhc.Model.from_counts(counts, 2)
    o = oracle.Model.from_counts(counts, 2)
    lo, co = o.codes_o2()
    lg, cg = m.codes_o2()
    assert np.array_equal(lg, lo) and np.array_equal(cg, co)
    lens = np.asarray(lo).reshape(65536, 256)
    assert lens.max() > 56 and lens.max() <= 64
    rng = np.random.default_rng(3)
    n = (1 << 20) + 5
    # mostly the frequent symbols (short codes), with the rare ones (longest codes) sprinkled in
    data = (59 - np.minimum(rng.geometric(0.5, size=n) - 1, 59)).astype(np.uint8)
    rare = rng.random(n) < 0.02
    data[rare] = rng.integers(0, 8, size=int(rare.sum()), dtype=np.uint8)
    data = data.tobytes()
    blob, nbits, idx = m.compress(data, chunk_symbols=256)
    ref, ref_bits = o.compress(data)
    assert (nbits, blob) == (ref_bits, ref)
    assert m.decompress(blob, index=idx, chunk_symbols=256, n_symbols=n) == data


def test_order2_config5_size_properties_parity_unpinned(mhc):
    """BASELINE config 5's size on one card: 16 GiB of Lorem-Ipsum-style text under order-2 contexts.  No oracle can
    run at this size (and the reference has no order 2 at all: parity unpinned), so the properties the domain gives:
    the counts add up to n, the payload length is the dot product of histogram and code lengths, the stream round-trips,
    and a model reloaded from the table file carries the same codewords."""
    import torch
    import bench
    bench.CHUNK = 1024
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = 16 << 30
    data = bench.generate("text", n, 1, 0, dev)
    codec = bench.Codec(mhc, n, dev, order=2)
    codec.histogram(data, 0x2020)
    assert int(codec.counts.sum().item()) == n
    model = codec.build_model()
    want = torch.zeros(1, dtype=torch.int64, device=dev)
    codec.payload_bits(model, codec.counts, want)
    codec.encode(model, data, 0x2020)
    codec.decode(model)
    torch.cuda.synchronize()
    assert codec.lib.mh_dev_status(codec.enc_ws.data_ptr(), codec.stream()) == 0
    assert codec.lib.mh_dev_status(codec.dec_ws.data_ptr(), codec.stream()) == 0
    nbits = int(codec.nbits[0].item())
    assert nbits == int(want.item())                              # histogram . lengths
    assert nbits < 0.3 * 8 * n                                    # order 2 on this text: ratio ~0.265 (order 1: 0.42)
    assert torch.equal(codec.decoded, data)
    table = model.table_bytes()
    t = mhc.Model.from_table(table)
    assert t.type == 2 and t.table_bytes() == table
    for which in (1, 3):                                          # code lengths and codewords of all 16.7 M pairs
        assert t.image(which) == model.image(which)


# ------------------------------------------------------------------ the model build shared by two ranks (reduce-scatter path)

def _o2_rank(rank, world, port, data, q, exchange):
    import os, sys, importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)                 # both ranks share the one card of the test box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as e2
        mhc = e2.load_package()
        lib = mhc.lib()
        sharded = importlib.import_module("mhc_amd.sharded")
        lo, hi = sharded.shard_bounds(len(data), world)[rank]
        shard = torch.frombuffer(bytearray(data[lo:hi] + bytes(32)), dtype=torch.uint8)[:hi - lo].cuda()
        ctx0 = 0x2020 if lo == 0 else (data[lo - 2] << 8 | data[lo - 1])
        counts = torch.zeros(1 << 24, dtype=torch.int64, device="cuda")
        assert lib.mh_dev_histogram_o2(shard.data_ptr(), hi - lo, ctx0, counts.data_ptr(), None) == 0
        model = sharded.order2_model(mhc, counts, None, exchange=exchange)
        q.put((rank, model.table_bytes(), bytes(model.image(1)), model.max_code_len))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["scatter", "compact", "allreduce"])
def test_order2_model_shared_by_two_ranks_parity_unpinned(mhc, oracle, exchange):
    """SURVEY.md 8e's order-2 exchange with the shipped code, two gloo ranks on the one card: local histograms, then
    scatter: reduce-scatter (staged: gloo), each rank builds the trees of ITS half of the contexts, the per-context arrays
    are all-gathered in place, mh_dev_model2_finish; compact: only the rows of the live contexts are summed; allreduce: all
    128 MiB — both ranks end with the table file and the code lengths the oracle derives from the whole input."""
    import socket
    import torch.multiprocessing as mp
    data = text_like((2 << 20) + 77, 21)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_o2_rank, args=(r, 2, port, data, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    o = oracle.Model.from_data(data, 2)
    lo, _ = o.codes_o2()
    for g in got:
        assert g[1] == o.table_bytes()
        assert np.array_equal(np.frombuffer(g[2], dtype=np.uint8), lo)
