#!/usr/bin/env python3
"""bench.py — encode+decode GB/s of the MI355X-native Markov-Huffman hot path (BASELINE.json metric).

One "step" = one full pass of the hot path over one HBM-resident synthetic stream:
    order-1 histogram -> [RCCL all-reduce of the 256x256 counts when N > 1] -> per-context tree build
    -> encode (payload + chunk index) -> decode (from payload + index)
N = 1 workload: BASELINE.json configs[2] — 16 GiB of Zipf(s=1.1) bytes (the configuration the metric
is quoted on; it fits one GPU).  N > 1, default: that ONE 16 GiB stream is split into N contiguous shards
(strong scaling: the metric's "16 GB byte stream at 1/2/4/8 GPUs"; `--total-size` for another total).
`--size` instead gives every GPU that many bytes (weak scaling), and `--config 4` is BASELINE.json
configs[3]: uniform random bytes, 8 GiB per GPU (64 GiB on 8).  Either way the only collective on the
data path is the histogram all-reduce, and `config.workload` says which mode ran.

Prints ONE JSON line (rank 0).  `value` = total uncompressed bytes of all ranks / step time (max over
ranks), inputs resident in HBM.  `roofline` describes the dominant kernel: algorithmic HBM bytes per
launch / that kernel's average launch duration (HIP events on the launch stream) against 8 TB/s.
`cpu_baseline` = the reference CPU path on the host cores (oracle/_ref binary when present, else the
oracle port) on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
COPY_CEILING_GBS = 6290.0  # same guide: what a float4 copy kernel reaches
CHUNK = int(os.environ.get("MH_BENCH_CHUNK", "0"))   # 0: 1024 symbols, 256 below 2 GiB per GPU (keeps every CU busy)


# ---- counter-based generator: byte i of a stream is a pure function of (seed, i), so any
# shard or slice can be regenerated independently, on the device (torch int64 ops) or on the host (numpy).
_M64 = (1 << 64) - 1
_GOLDEN = 0x9E3779B97F4A7C15


def _s64(v):
    """Python int -> the int64 with the same 64-bit pattern."""
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x, k, xp):
    """Logical right shift of int64 values (torch and numpy shift arithmetically)."""
    return (x >> k) & ((1 << (64 - k)) - 1)


def splitmix64(counter, seed, xp):
    """splitmix64 finaliser of (counter + 1) * golden + seed * golden^2; int64 in, int64 out (bit pattern)."""
    x = counter * _s64(_GOLDEN) + _s64((seed + 1) * _GOLDEN * _GOLDEN + _GOLDEN)
    x = (x ^ _lsr(x, 30, xp)) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27, xp)) * _s64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31, xp)


def zipf_thresholds(s=1.1):
    """T[k] = round(CDF(k+1) * 2^32), k = 0..254: a 32-bit uniform u maps to the number of T[k] <= u."""
    w = 1.0 / np.arange(1, 257, dtype=np.float64) ** s
    cdf = np.cumsum(w / w.sum())
    return np.minimum(np.rint(cdf[:255] * 4294967296.0), 4294967295.0).astype(np.int64)


def synth_slice(kind, seed, first, count, device=None):
    """Bytes [first, first + count) of the (kind, seed) stream: torch uint8 tensor on `device`, or a numpy
    array when device is None.  Both give the same bytes."""
    if device is None:
        with np.errstate(over="ignore"):
            z = splitmix64(np.arange(first, first + count, dtype=np.int64), seed, np)
        if kind == "uniform":
            return _lsr(z, 56, np).astype(np.uint8)
        return np.searchsorted(zipf_thresholds(), _lsr(z, 32, np), side="right").astype(np.uint8)
    z = splitmix64(torch.arange(first, first + count, dtype=torch.int64, device=device), seed, torch)
    if kind == "uniform":
        return _lsr(z, 56, torch).to(torch.uint8)
    thr = torch.from_numpy(zipf_thresholds()).to(device)
    return torch.searchsorted(thr, _lsr(z, 32, torch), right=True).to(torch.uint8)


LOREM = ("lorem ipsum dolor sit amet consectetur adipiscing elit sed do eiusmod tempor incididunt ut labore et dolore "
         "magna aliqua enim ad minim veniam quis nostrud exercitation ullamco laboris nisi aliquip ex ea commodo "
         "consequat duis aute irure in reprehenderit voluptate velit esse cillum eu fugiat nulla pariatur excepteur "
         "sint occaecat cupidatat non proident sunt culpa qui officia deserunt mollit anim id est laborum at vero eos "
         "accusamus iusto odio dignissimos ducimus blanditiis praesentium voluptatum deleniti atque corrupti quos "
         "dolores quas molestias excepturi occaecati cupiditate provident similique mollitia animi fuga harum quidem "
         "rerum facilis expedita distinctio nam libero tempore cum soluta nobis eligendi optio cumque nihil impedit "
         "quo minus quod maxime placeat facere possimus omnis assumenda repellendus temporibus autem quibusdam "
         "officiis debitis aut necessitatibus saepe eveniet voluptates repudiandae recusandae itaque earum hic "
         "tenetur sapiente delectus reiciendis voluptatibus maiores alias perferendis doloribus asperiores repellat").split()


def lorem_block(nbytes, seed):
    """Lorem-Ipsum-style ASCII: sentences of 4-16 words, capitalised, '. ' terminated,
    paragraphs of 3-8 sentences ending in a newline."""
    rng = np.random.default_rng(seed)
    out = bytearray()
    while len(out) < nbytes:
        for _ in range(int(rng.integers(3, 9))):
            words = [LOREM[i] for i in rng.integers(0, len(LOREM), int(rng.integers(4, 17)))]
            words[0] = words[0].capitalize()
            out += (" ".join(words) + ". ").encode()
        out[-1:] = b"\n"
    return bytes(out[:nbytes])


def generate(kind, n, seed, first_byte, device):
    """Bytes [first_byte, first_byte + n) of the seeded synthetic stream, resident on `device`."""
    out = torch.empty(n, dtype=torch.uint8, device=device)
    if kind == "text":
        # 8 MiB of generated text, tiled (the survey tiled the repo's ipsum file the same way); a shard starts
        # at its own phase of the tiling, so shards concatenate into the single tiled stream
        base = torch.frombuffer(bytearray(lorem_block(8 << 20, seed)), dtype=torch.uint8).to(device)
        phase = first_byte % base.numel()
        reps = (n + phase + base.numel() - 1) // base.numel()
        out.copy_(base.repeat(reps)[phase:phase + n])
        return out
    if kind not in ("zipf", "uniform"):
        raise ValueError(kind)
    sl = 1 << 26
    for off in range(0, n, sl):
        m = min(sl, n - off)
        out[off:off + m] = synth_slice(kind, seed, first_byte + off, m, device)
    return out


def Codec(mhc, n, device, order=1):
    """One rank's device buffers + C-ABI calls: sharded.HipBackend with this module's chunk size (tools/ and tests use it)."""
    import importlib
    sharded = importlib.import_module("mhc_amd.sharded")
    return sharded.HipBackend(mhc, n, device, order=order, chunk_symbols=CHUNK or 1024,
                              use_fine=not os.environ.get("MH_BENCH_NO_FINE"),
                              two_pass_encode=bool(os.environ.get("MH_BENCH_TWO_PASS_ENCODE")))


def cpu_baseline(mhc, table_bytes, sample, gpu_payload_prefix_check):
    """Reference CPU path on a bounded sample: round-trip GB/s = bytes / (compress + decompress wall)."""
    from oracle import mh_oracle
    n = len(sample)
    cores = 1
    if os.path.exists(mh_oracle.REF_BIN):
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
            src = os.path.join(tmp, "in.bin")
            with open(src, "wb") as f:
                f.write(sample)
            run = lambda a: subprocess.run([mh_oracle.REF_BIN] + a, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            t0 = time.perf_counter()
            run([src, "-o", os.path.join(tmp, "c"), "-d", os.path.join(tmp, "t")])
            t1 = time.perf_counter()
            run([os.path.join(tmp, "c"), "-o", os.path.join(tmp, "d"), "-x", "-e", os.path.join(tmp, "t")])
            t2 = time.perf_counter()
            ok = open(os.path.join(tmp, "d"), "rb").read() == sample
            # and the GPU stream vs the genuine reference with the GPU-built table on the same bytes
            with open(os.path.join(tmp, "gt"), "wb") as f:
                f.write(table_bytes)
            run([src, "-o", os.path.join(tmp, "gc"), "-e", os.path.join(tmp, "gt")])
            ref_stream = open(os.path.join(tmp, "gc"), "rb").read()[1:]
        kind = "reference"
    else:
        t0 = time.perf_counter()
        om = mh_oracle.Model.from_data(sample, 1)
        blob, _ = om.compress(sample)
        t1 = time.perf_counter()
        ok = om.decompress(blob) == sample
        t2 = time.perf_counter()
        ref_stream = mh_oracle.Model.from_table(table_bytes).compress(sample)[0][1:]
        kind = "port"
    stream_ok = gpu_payload_prefix_check(ref_stream)
    return {
        "value": round(n / (t2 - t0) / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": kind,
        "sample": "first %d MiB of rank 0's stream: compress (histogram+tree+encode) %.2f s, decompress %.2f s, 1 thread"
                  % (n >> 20, t1 - t0, t2 - t1),
        "compress_GBps": round(n / (t1 - t0) / 1e9, 4), "decompress_GBps": round(n / (t2 - t1) / 1e9, 4),
        "round_trip_ok": bool(ok), "gpu_stream_equals_cpu_stream_on_sample": bool(stream_ok),
    }


def index_free_decode(mhc, codec, model, data, nbits, prev0, reps=2):
    """The real drop-in decode: the reference's `.cm` carries no index (src/coding.cpp:35-59), so the
    payload the bench just wrote is decoded again with NO index handed in.  [r5] Two passes over the payload:
    mh_dev_decode_stream_states (every 352-bit segment's entry state and symbol count) and mh_dev_decode_stream_emit (the
    segment decoder writes the bytes) — no index is built.  `via_index` = the round-4 way beside it (mh_dev_build_index_fine
    rebuilds chunk index and fine index, mh_dev_decode_fine decodes from them: three passes).  Outside the timed loop, like
    cpu_baseline; HIP events on the launch stream.  (The states pass waits for the device between its sub-passes.)"""
    lib, n, dev = codec.lib, codec.n_now, codec.device
    wsb = int(lib.mh_dev_build_index_workspace(nbits))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    nsym = torch.zeros(1, dtype=torch.int64, device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    out = {}
    # ---- two passes, no index
    codec.decoded.zero_()
    t_a = t_b = 0.0
    path = 0
    for rep in range(reps + 1):
        e = [ev() for _ in range(3)]
        e[0].record()
        codec.check(lib.mh_dev_decode_stream_states(model.handle, codec.payload.data_ptr(), nbits, prev0, nsym.data_ptr(), ws.data_ptr(), wsb,
                                                    codec.stream()), "mh_dev_decode_stream_states")
        e[1].record()
        path = int(lib.mh_dev_index_path(ws.data_ptr(), codec.stream()))
        if path != 6:
            break
        codec.check(lib.mh_dev_decode_stream_emit(model.handle, codec.payload.data_ptr(), nbits, prev0, codec.decoded.data_ptr(), n, ws.data_ptr(), wsb,
                                                  codec.stream()), "mh_dev_decode_stream_emit")
        e[2].record()
        torch.cuda.synchronize()
        if rep:                                        # (the first repetition warms up)
            t_a += e[0].elapsed_time(e[1])
            t_b += e[1].elapsed_time(e[2])
    if path == 6:
        t_a, t_b = t_a / reps, t_b / reps
        ok = (int(nsym.item()) == n and lib.mh_dev_status(ws.data_ptr(), codec.stream()) == 0 and torch.equal(codec.decoded[:n], data))
        out = {"states_ms": round(t_a, 3), "emit_ms": round(t_b, 3), "total_ms": round(t_a + t_b, 3),
               "GBps": round(n / ((t_a + t_b) * 1e-3) / 1e9, 2) if t_a + t_b > 0 else None,
               "passes_over_the_payload": 2, "path": path, "workspace_bytes": wsb, "bit_exact": bool(ok)}
    else:
        out = {"path": path, "note": "this stream does not take the two-pass path (see include/mh.h); via_index is what decodes it"}
    # ---- the round-4 way: both indices, then the tile decoder
    index = torch.zeros(nbits // codec.chunk + 2, dtype=torch.int64, device=dev)   # sized as a caller that does not know n must
    fine_cap = nbits // max(model.min_code_len, 1) // 64 + 2     # (a code has at least min_code_len bits)
    fine = torch.zeros(fine_cap, dtype=torch.int32, device=dev)
    codec.decoded.zero_()
    t_idx = t_dec = 0.0
    for rep in range(reps + 1):
        e = [ev() for _ in range(3)]
        e[0].record()
        codec.check(lib.mh_dev_build_index_fine(model.handle, codec.payload.data_ptr(), nbits, prev0, index.data_ptr(), index.numel(), codec.chunk,
                                                fine.data_ptr(), fine_cap, nsym.data_ptr(), ws.data_ptr(), wsb, codec.stream()), "mh_dev_build_index_fine")
        e[1].record()
        codec.check(lib.mh_dev_decode_fine(model.handle, codec.payload.data_ptr(), nbits, None, codec.decoded.data_ptr(), n, index.data_ptr(), codec.chunk,
                                           fine.data_ptr(), codec.dec_ws.data_ptr(), codec.dec_ws_bytes, codec.stream()), "mh_dev_decode_fine")
        e[2].record()
        torch.cuda.synchronize()
        if rep:
            t_idx += e[0].elapsed_time(e[1])
            t_dec += e[1].elapsed_time(e[2])
    t_idx, t_dec = t_idx / reps, t_dec / reps
    ok = (int(nsym.item()) == n and lib.mh_dev_status(ws.data_ptr(), codec.stream()) == 0 and lib.mh_dev_status(codec.dec_ws.data_ptr(), codec.stream()) == 0
          and torch.equal(codec.decoded[:n], data) and torch.equal(index[:codec.nidx], codec.index[:codec.nidx]))
    out["via_index"] = {"index_build_ms": round(t_idx, 3), "decode_ms": round(t_dec, 3), "total_ms": round(t_idx + t_dec, 3),
                        "GBps": round(n / ((t_idx + t_dec) * 1e-3) / 1e9, 2) if t_idx + t_dec > 0 else None,
                        "passes_over_the_payload": 3,
                        "index_path": int(lib.mh_dev_index_path(ws.data_ptr(), codec.stream())),       # 5 = tiles (fast path), 1 = segment iteration, ...
                        "decode_path": int(lib.mh_dev_decode_path(codec.dec_ws.data_ptr(), codec.stream())),
                        "index_workspace_bytes": wsb, "fine_index_bytes": int(fine_cap * 4), "bit_exact": bool(ok)}
    if "bit_exact" not in out:
        out["bit_exact"] = bool(ok)
    return out


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: this process starts `python -m torch.distributed.run` with N ranks
    of this very command as a CHILD, relays what the ranks print and returns the child's exit code.  It runs before
    anything here touches a GPU (importing torch does not), and nothing is exec'ed."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:                         # rank 0's JSON line to stdout; whatever else the ranks say (gloo's
        dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr     # connection chatter) to stderr
        dst.write(line)
        dst.flush()
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=None, help="bytes per GPU, weak scaling (default 16 GiB)")
    ap.add_argument("--total-size", type=int, default=None,
                    help="strong scaling: ONE stream of this many bytes split into N contiguous shards (e.g. 17179869184)")
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[]: 3 (default) = 16 GiB Zipf(1.1) per GPU; 4 = uniform random, 8 GiB per GPU (64 GiB on 8); "
                         "2 = 256 MiB of Lorem-Ipsum-style ASCII on one GPU (the reference's own benchmark size, README.md:174); "
                         "5 = order 2 on 16 GiB of that text, ONE stream split over the GPUs (extension: parity unpinned)")
    ap.add_argument("--kind", default=None, choices=["zipf", "uniform", "text"])
    ap.add_argument("--order", type=int, default=1, choices=[1, 2],
                    help="2 = order-2 contexts (BASELINE configs[4]; extension the reference does not have: parity unpinned; 1 GPU)")
    ap.add_argument("--o2-exchange", default="compact", choices=["compact", "allreduce"],
                    help="order 2, N > 1: compact = all-reduce of the live contexts' rows only (default); allreduce = all-reduce all "
                         "128 MiB.  (SURVEY 8e's reduce-scatter + per-rank tree build + all-gather form moves MORE bytes than the "
                         "all-reduce it replaces — the dense node arrays travel — and is no longer offered here; "
                         "sharded.order2_model(exchange='scatter') keeps it for the API's test.)")
    ap.add_argument("--cpu-sample", type=int, default=256 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo (collectives staged through host memory) is only for rehearsing the N>1 path on one GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))                   # parent of N ranks: never touches a GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if "MH_BENCH_DEVICE" in os.environ:          # rehearsal: several ranks on one card
        local = int(os.environ["MH_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # MH_BENCH_DIST1=1 (rehearsal on the one-GPU box): a single rank goes through the whole N > 1 path — process group,
    # every collective on the real backend (RCCL), pre-shifted encode — so that its calls have run once before the
    # driver's multi-GPU bench
    multi = world > 1 or os.environ.get("MH_BENCH_DIST1") == "1"
    if multi and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if multi:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    mhc = entry.load_package()
    mhc.lib()                      # fails loudly if libmhc.so is missing: there is no fallback path
    import importlib
    sharded = importlib.import_module("mhc_amd.sharded")
    if args.config == 5:                                  # BASELINE configs[4]: order-2 contexts on 16 GiB of text (one stream, split N ways)
        args.order = 2
    kind = args.kind or {2: "text", 3: "zipf", 4: "uniform", 5: "text"}[args.config]
    # ---- what each rank holds: bytes [first, first + n) of ONE seeded stream
    if args.total_size is None and args.size is None and world > 1 and ((args.config == 3 and args.order == 1) or args.config == 5):
        # BASELINE.json's metric is ONE 16 GB stream at 1/2/4/8 GPUs: with no size given, N > 1 splits that stream
        # (strong scaling); --size keeps a fixed amount per GPU (weak scaling), --config 4 is 8 GiB per GPU by definition
        args.total_size = 16 << 30
    if args.total_size is not None:
        mode, total = "strong", args.total_size
        first, hi = sharded.shard_bounds(total, world)[rank]     # shards start on 16-byte boundaries (device loads are 16-byte vectors)
        n = hi - first
    else:
        mode = "weak"
        n = args.size if args.size is not None else {2: 256 << 20, 3: 16 << 30, 4: 8 << 30, 5: 16 << 30}[args.config]
        first, total = rank * n, n * world
    global CHUNK
    if CHUNK == 0:
        CHUNK = 1024 if n >= (2 << 30) else 256
    seed = {"zipf": 2, "uniform": 3, "text": 1}[kind]     # C2 seed 1, C3 seed 2, C4 seed 3
    data = generate(kind, n, seed, first, device)
    # context of each shard's first byte = last byte of the previous shard (' ' for rank 0)
    # (order 2: the last TWO bytes; prev0 then is the 16-bit context)
    prev0 = 0x20 if args.order == 1 else 0x2020
    if multi:
        k = args.order
        last = torch.zeros(world * k, dtype=torch.uint8, device=device)
        sharded.all_gather_flat(last, data[-k:].clone())
        if rank > 0:
            tail = last[(rank - 1) * k:rank * k].tolist()
            prev0 = tail[0] if k == 1 else (tail[0] << 8 | tail[1])
    # the rank's buffers and C-ABI calls (sharded.HipBackend); the step below is sharded.compress_step — the shipped
    # orchestration, the one the world-2 tests drive — followed by the decode of what it left in HBM
    codec = sharded.HipBackend(mhc, n, device, order=args.order, chunk_symbols=CHUNK, o2_exchange=args.o2_exchange,
                               use_fine=not os.environ.get("MH_BENCH_NO_FINE"),
                               two_pass_encode=bool(os.environ.get("MH_BENCH_TWO_PASS_ENCODE")))

    ev = lambda: torch.cuda.Event(enable_timing=True)
    stage_ms = {"hist": 0.0, "allreduce": 0.0, "tree": 0.0, "encode": 0.0, "decode": 0.0}
    model = None
    last_step = {}
    recorded = []                                     # the timed steps' events

    def step(record):
        nonlocal model, last_step
        marks = [ev()]
        marks[0].record()

        def mark(name):
            e = ev()
            e.record()
            marks.append(e)
        last_step = sharded.compress_step(codec, data, prev0, distributed=multi, mark=mark)   # hist, allreduce, tree, encode
        model = last_step["model"]
        codec.decode(model)                           # reads the payload length where the encoder left it (HBM)
        mark("decode")
        # no device-wide wait here: the step's only host wait is the model build's (16 KiB of table sizes); the next step is
        # enqueued behind this one on the same stream, and the stage times are read from the events after the timed loop
        if record:
            recorded.append(marks)
        else:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
        codec.nbits_hint = int(codec.nbits[0].item())
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    for marks in recorded:
        for i, k in enumerate(stage_ms):
            stage_ms[k] += marks[i].elapsed_time(marks[i + 1])
    def all_reduce(t, op=dist.ReduceOp.SUM):       # (after the timed region: timing and verdicts of the ranks)
        if t.is_cuda and dist.get_backend() != "nccl":
            c = t.cpu()
            dist.all_reduce(c, op=op)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op)

    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness of what was timed (outside the timed region)
    nbits = int(codec.nbits[0].item())
    rc, rc2, rc3 = codec.statuses()                  # encoder, decoder, histogram (its counts add up to n)
    enc_path, dec_path = codec.paths()
    round_trip = bool(rc == 0 and rc2 == 0 and rc3 == 0 and torch.equal(codec.decoded[:n], data))
    start_low = (int(last_step["start_bit"].item()) & 7) if multi else 0
    if multi:     # the shard ended where its histogram said it would: the ranks' payloads tile the global stream
        round_trip = round_trip and nbits == start_low + int(last_step["my_bits"].item())
    ok = torch.tensor([1 if round_trip else 0], device=device)
    tot_bits = torch.tensor([nbits - start_low], dtype=torch.int64, device=device)
    if multi:
        all_reduce(ok, op=dist.ReduceOp.MIN)
        all_reduce(tot_bits)
    round_trip_all = bool(ok.item())

    if rank == 0:
        K = args.steps
        ms = {k: v / K for k, v in stage_ms.items()}
        r = nbits / 8.0 / max(n, 1)
        total_bytes = float(total)
        gbps = total_bytes / (elapsed / K) / 1e9
        # algorithmic bytes per launch: hist 1, encode 1 + r, decode r + 1 per input byte
        # which decoder ran is on record in the workspace (mh_dev_decode_path): 1 the tile decoder, 2 the chunk decoder
        dec_name = "decode_tile_kernel" if dec_path == 1 else "decode_kernel"
        kernels = {
            "hist_o1_kernel": (1.0 * n, ms["hist"]),
            "enc_region_kernel": ((1.0 + r) * n, ms["encode"]),
            dec_name: ((1.0 + r) * n, ms["decode"]),
        }
        dom = max(kernels, key=lambda k: kernels[k][1])
        if args.order == 2:
            # which encoder ran is on record in the workspace (mh_dev_encode_path): 4 the one-pass enc_chain_kernel, else the pair
            enc2 = "enc_chain_kernel" if enc_path == 4 else "enc_emit_kernel<2>"
            kernels = {k.replace("hist_o1", "hist_o2").replace("enc_region_kernel", enc2).replace("decode_kernel", "decode2_kernel"): v
                       for k, v in kernels.items()}
            dom = max(kernels, key=lambda k: kernels[k][1])
        # counter-derived figures of the dominant kernel come from the committed --pmc summaries (profiles/): HBM
        # traffic per launch, and the secondary bounds — share of the kernel's time its vector ALUs issue, its LDS is
        # busy, its texture-address units are busy — because every kernel of this path is issue-/LDS-bound long before
        # HBM (tools/make_traffic_json.py writes both files from gpurun_out/<pmc run>)
        # ... and each figure is tied to the sources of its kernel: one collected on other sources is dropped (provenance.py)
        provenance = importlib.import_module("mhc_amd.provenance")
        traffic, secondary, counters_from = provenance.counters_for(dom, n, os.path.join(ROOT, "profiles"))
        ach = kernels[dom][0] / (kernels[dom][1] * 1e-3) / 1e9
        kname = {"zipf": "Zipf(s=1.1)", "uniform": "uniform", "text": "Lorem-Ipsum-style ASCII"}[kind]
        if mode == "weak":
            workload = ("order-%d Markov-Huffman round trip (histogram+tree+encode+decode), %s %s per GPU, "
                        "chunk index every %d symbols%s" % (args.order, ("%d GiB" % (n >> 30)) if n >= 1 << 30 else ("%d MiB" % (n >> 20)),
                                                            kname, CHUNK, " [order 2: extension, parity unpinned]" if args.order == 2 else ""))
        else:
            workload = ("order-%d Markov-Huffman round trip (histogram+tree+encode+decode), ONE %.3f GiB %s stream split into %d "
                        "contiguous shards (strong scaling), chunk index every %d symbols%s" %
                        (args.order, total / 2.0 ** 30, kname, world, CHUNK, " [order 2: extension, parity unpinned]" if args.order == 2 else ""))
        out = {
            "metric": "encode+decode GB/s on 16 GB byte stream, bit-exact round-trip",
            "value": round(gbps, 3), "unit": "GB/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(elapsed / K * 1e3, 3), "higher_is_better": True, "scaling": mode,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": args.config, "bytes_per_gpu": n, "total_bytes": int(total),
                       "generator": "counter-based splitmix64 (seed %d), byte i = f(seed, i): any shard regenerable alone" % seed
                                    if kind != "text" else "8 MiB of seeded Lorem-Ipsum-style text, tiled",
                       "sharding": ("contiguous byte ranges, histogram all-reduce (%s, %s)" %
                                    ("RCCL over xGMI" if args.backend == "nccl" else "gloo staged through host memory: REHEARSAL, not RCCL",
                                     "512 KiB" if args.order == 1 else (
                                         "order 2: the rows of the live contexts only" if args.o2_exchange == "compact" else "128 MiB"))) if multi else "single GPU",
                       "backend": (dist.get_backend() if multi else None), "ranks": (dist.get_world_size() if multi else 1)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_ratio": round(traffic / kernels[dom][0], 3) if traffic else None,
                         "secondary": secondary, "counters_from": counters_from or None,
                         "algorithmic_bytes_per_launch": int(kernels[dom][0])},
            "stages_ms": {k: round(v, 3) for k, v in ms.items()},
            "compress_ms": round(ms["hist"] + ms["allreduce"] + ms["tree"] + ms["encode"], 3),      # compress = hist + tree + encode
            "compress_GBps": round(n / ((ms["hist"] + ms["allreduce"] + ms["tree"] + ms["encode"]) * 1e-3) / 1e9, 2),
            "stage_GBps_input": {k: (round(n / (v * 1e-3) / 1e9, 2) if v > 0 else None) for k, v in ms.items()},
            "kernel_roofline_frac": {k: round(b / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if t > 0 else None for k, (b, t) in kernels.items()},
            "encode_read_roofline_frac": round(n / (ms["encode"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms["encode"] > 0 else None,
            # what that fraction can reach at all: the encoder moves (1 + r) bytes per input byte, and a plain copy reaches
            # 6.29 of the 8.0 TB/s (MI355X_MICROARCH.md): 1 / (1 + r) x 6.29 / 8.0 — the north star's 0.50 lies above it
            "encode_read_ceiling": round(1.0 / (1.0 + r) * COPY_CEILING_GBS / HBM_PEAK_GBS, 4),
            # what does not shrink with the stream: the tree stage (256 contexts whatever n is), the collective, and whatever of the
            # step is not inside a stage (launch gaps, the model build's one host wait).  At 16 GiB a few per cent; at config 2's
            # 256 MiB a large part of the step
            "fixed_cost_share": round((ms["tree"] + ms["allreduce"] + max(elapsed / K * 1e3 - sum(ms.values()), 0.0)) / (elapsed / K * 1e3), 4),
            "gap_ms": round(max(elapsed / K * 1e3 - sum(ms.values()), 0.0), 4),
            "compressed_ratio": round(r, 5), "total_payload_bits": int(tot_bits.item()), "round_trip_bit_exact": round_trip_all,
            # out-of-band bytes the decoder is handed beside the payload: the sidecar chunk index (8 B per chunk) and the
            # device-only fine index (4 B per 64 symbols); traffic of both kernels, never credit
            "index_bytes": {"chunk_index": int(codec.nidx * 8), "fine_index": int(((n + codec.fine_symbols - 1) // codec.fine_symbols) * 4) if codec.use_fine else 0},
            "max_code_len": model.max_code_len,
            "decode_tables": {"chunk_decoder": dict(zip(("primary_bits", "secondary_entries", "in_lds"), model.decode_layout())),
                              "tile_decoder": dict(zip(("primary_bits_in_lds", "secondary_bits", "secondary_entries_in_l2"),
                                                       model.tile_layout()))},
        }
        if args.order == 2:
            out["order"] = 2
            out["parity"] = "unpinned: the reference has no order 2; checked against the generalised oracle in tests/test_gpu_order2.py"
        if world == 1 and args.order == 1 and not os.environ.get("MH_BENCH_NO_INDEX_FREE"):
            out["decode_index_free"] = index_free_decode(mhc, codec, model, data, nbits, prev0)
        if world == 1 and not args.no_cpu_baseline and args.order == 1:
            sample_n = min(args.cpu_sample, n) & ~(CHUNK - 1)
            sample = data[:sample_n].cpu().numpy().tobytes()

            def prefix_check(ref_stream):
                # the GPU payload's first sample_n symbols end at the index entry of chunk sample_n / CHUNK
                if sample_n == n:
                    end_bits = nbits
                else:
                    end_bits = int(codec.index[sample_n // CHUNK].item()) & mhc.INDEX_BIT_MASK
                full = end_bits // 8
                gpu = codec.payload[:full].cpu().numpy().tobytes()
                return gpu == ref_stream[:full] and len(ref_stream) == (end_bits + 7) // 8
            table = model.table_bytes()               # before the workspace is reused for the sample's histogram model
            out["cpu_baseline"] = cpu_baseline(mhc, table, sample, prefix_check)
            # the histogram kernel against the oracle's on the same sample (its LDS counter overflow path runs here)
            from oracle import mh_oracle
            codec.histogram(data, 0x20, sample_n)
            torch.cuda.synchronize()
            out["cpu_baseline"]["gpu_histogram_equals_cpu_histogram_on_sample"] = bool(
                np.array_equal(codec.counts.cpu().numpy().astype(np.uint64), mh_oracle.histogram_o1(sample)))
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()
    if not round_trip_all:
        sys.exit(2)


if __name__ == "__main__":
    main()
