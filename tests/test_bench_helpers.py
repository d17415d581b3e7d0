"""bench.py's synthetic inputs: byte i of a stream is a pure function of (kind, seed, i), the same on the host
(numpy) and where torch computes it — the bench's CPU-baseline sample and every rank's shard rest on that."""
import numpy as np
import pytest
import torch

import bench


@pytest.mark.parametrize("kind", ["zipf", "uniform"])
def test_a_slice_is_a_pure_function_of_its_position(kind):
    seed = {"zipf": 2, "uniform": 3}[kind]
    whole = bench.synth_slice(kind, seed, 1000, 50000)                      # numpy
    assert whole.dtype == np.uint8 and whole.size == 50000
    # any cut of the range gives the same bytes (shards regenerate their own part)
    parts = np.concatenate([bench.synth_slice(kind, seed, 1000, 123), bench.synth_slice(kind, seed, 1123, 49877)])
    assert np.array_equal(whole, parts)
    # torch (the device path of the bench, on the CPU here) = numpy
    t = bench.synth_slice(kind, seed, 1000, 50000, device=torch.device("cpu"))
    assert t.dtype == torch.uint8 and np.array_equal(t.numpy(), whole)
    # another seed is another stream
    assert not np.array_equal(whole, bench.synth_slice(kind, seed + 1, 1000, 50000))


def test_zipf_bytes_follow_the_exponent():
    """Rank k (byte value k - 1) with probability proportional to k^-1.1: the first symbol about 20.65 % of the stream, entropy about
    5.77 bits (SURVEY 8d's computed properties of config 3)."""
    x = bench.synth_slice("zipf", 2, 0, 1 << 20)
    p = np.bincount(x, minlength=256) / x.size
    assert abs(p[0] - 0.2065) < 0.003 and p[0] > p[1] > p[2] > p[5] > p[20]
    h = -(p[p > 0] * np.log2(p[p > 0])).sum()
    assert abs(h - 5.766) < 0.03
    thr = bench.zipf_thresholds()
    assert thr.size == 255 and np.all(np.diff(thr) >= 0) and thr[-1] <= 4294967295


def test_uniform_bytes_are_flat():
    x = bench.synth_slice("uniform", 3, 12345, 1 << 20)
    c = np.bincount(x, minlength=256)
    assert c.min() > 3600 and c.max() < 4600                              # 4096 expected per value


def test_lorem_block_is_seeded_ascii_text():
    a, b = bench.lorem_block(100000, 1), bench.lorem_block(100000, 1)
    assert a == b and len(a) == 100000 and bench.lorem_block(100000, 2) != a
    assert all(32 <= ch < 127 or ch == 10 for ch in a)
    assert b". " in a and b"\n" in a and len(set(a)) < 60               # a few dozen byte values, like test/input/input_ipsum.txt
    # generate() tiles 8 MiB of it; a shard that starts mid-tile continues the same stream
    g = bench.generate("text", 3000, 1, (8 << 20) - 1000, torch.device("cpu")).numpy().tobytes()
    base = bench.lorem_block(8 << 20, 1)
    assert g == base[-1000:] + base[:2000]
