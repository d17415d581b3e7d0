"""Multi-GPU orchestration of the hot path: contiguous byte-range shards, one process per GPU.

The data path has exactly one exchange step: a sum-all-reduce of the 65 536-entry
conditional histogram (512 KiB of uint64, RCCL over xGMI on the GPU box), after which every rank
builds the identical model deterministically (integer-only tree build, no broadcast).  A second tiny
collective — an all-gather of one uint64 per rank — gives each shard payload's global bit offset
BEFORE the shard is encoded (a shard's payload length is its local histogram dotted with the model's
code lengths), so every rank emits its payload pre-shifted by (start bit mod 8) and the shards
concatenate at byte offset start // 8 with one OR-merged seam byte (`stitch`).

This module holds only the orchestration.  The per-shard compute comes from a `backend`:
`HipBackend` (below) drives libmhc.so on the rank's GPU and is the only backend the product ships;
tests inject their own backend to exercise the collective logic with gloo on CPU.
"""
import ctypes
import os

import torch
import torch.distributed as dist

PREV0 = 0x20


def shard_bounds(n, world):
    """Contiguous shards, sizes differing by at most one 16-byte unit; returns [(lo, hi)] * world."""
    unit = 16
    units = (n + unit - 1) // unit
    out = []
    for r in range(world):
        lo = min(n, (units * r // world) * unit)
        hi = min(n, (units * (r + 1) // world) * unit)
        out.append((lo, hi))
    return out


def exchange_prev0(last_byte, has_data, group=None):
    """Context of each shard's first byte = last byte of the nearest non-empty shard before it
    (src/main.cpp:32: ' ' for the very first)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.tensor([int(last_byte) if has_data else -1], dtype=torch.int64, device=_dev(group))
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    prev0 = PREV0
    for r in range(rank):
        v = int(allv[r].item())
        if v >= 0:
            prev0 = v
    return prev0


def _dev(group):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def merged_histogram(local_counts, group=None):
    """local_counts: int64[65536] tensor on the rank's device (counts < 2^63).  In-place sum over ranks.
    With RCCL the tensor is reduced where it lies (HBM, over xGMI); the gloo backend (CPU rehearsals of the
    N > 1 path, also with the shards on one GPU) is handed a host copy."""
    if local_counts.is_cuda and dist.get_backend(group) != "nccl":
        host = local_counts.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        local_counts.copy_(host)
    else:
        dist.all_reduce(local_counts, op=dist.ReduceOp.SUM, group=group)
    return local_counts


def merged_histogram_o2_compact(local_counts, group=None, dense_above=0.25):
    """Order 2: in-place sum over ranks of int64[1 << 24] counts, moving only the rows of contexts that are live somewhere.
    Text-like sources touch a few thousand of the 65536 two-byte contexts, so the 128 MiB all-reduce (or the scatter
    exchange's 420 MiB) shrinks to: one all-reduce (MAX) of a 65536-entry liveness vector, then one all-reduce (SUM) of
    the live rows gathered into a compact buffer (3000 contexts: 6 MiB), scattered back where they belong.  One host
    wait (the number of live contexts).  Falls back to the dense all-reduce when more than `dense_above` of the contexts
    are live (flat sources: nothing to gain).  Returns the number of live contexts."""
    nctx = 65536
    staged = local_counts.is_cuda and dist.get_backend(group) != "nccl"
    rows = local_counts.view(nctx, 256)
    live = (rows != 0).any(dim=1).to(torch.int32)
    if staged:
        h = live.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        live.copy_(h)
    else:
        dist.all_reduce(live, op=dist.ReduceOp.MAX, group=group)
    idx = live.nonzero(as_tuple=False).squeeze(1)               # (the host wait: its length)
    nlive = int(idx.numel())
    if nlive > dense_above * nctx:
        merged_histogram(local_counts, group)
        return nlive
    if nlive == 0:
        return 0
    compact = rows.index_select(0, idx).contiguous()
    if staged:
        h = compact.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        compact.copy_(h)
    else:
        dist.all_reduce(compact, op=dist.ReduceOp.SUM, group=group)
    rows.index_copy_(0, idx, compact)
    return nlive


def order2_model(mhc, local_counts, stream=None, group=None, exchange="scatter"):
    """Order-2 model (extension, parity unpinned) shared by the ranks of `group` from each rank's LOCAL counts
    (int64[1 << 24] on the device).  exchange:
      "scatter"    SURVEY.md 8e's alternative to all-reducing 128 MiB: reduce-scatter of the counts (each rank gets the
                   global counts of its 65536 / G contexts), every rank builds the trees of THOSE contexts
                   (mh_dev_model2_build_slice), the per-context arrays (code lengths, codewords, tree nodes, sizes) are
                   all-gathered in place, mh_dev_model2_finish derives the tables.  Needs G to divide 65536.
      "allreduce"  all-reduce the counts, every rank builds everything (local_counts becomes the global histogram).
      "compact"    the same, but only the rows of live contexts travel (merged_histogram_o2_compact).
    `stream` must be torch's CURRENT stream (the default when None is passed means the null stream: pass
    torch.cuda.current_stream().cuda_stream as HipBackend does): the torch collectives here order against the current stream
    only, so library kernels launched on any other stream would race them (checked below).  The in-place RCCL all-gather of
    the scatter form has run with one rank on the real backend and with two ranks on gloo, never with N > 1 on RCCL.
    Returns the Model; it borrows a workspace tensor that is kept alive on the object.  With gloo (rehearsals with the
    shards on one card) the collectives are staged through host memory and the reduce-scatter is an all-reduce + slice
    (gloo has none)."""
    import ctypes as C
    lib = mhc.lib()
    if local_counts.is_cuda:
        given = stream.value if isinstance(stream, C.c_void_p) else stream
        if (given or 0) != torch.cuda.current_stream().cuda_stream:
            raise ValueError("order2_model: `stream` must be torch's current stream (the collectives order against that one only)")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    staged = local_counts.is_cuda and dist.get_backend(group) != "nccl"
    nctx = 65536
    if exchange == "compact":
        merged_histogram_o2_compact(local_counts, group)
        return mhc.Model.from_device_counts(local_counts.data_ptr(), 2, stream)
    if exchange != "scatter" or nctx % world != 0:
        merged_histogram(local_counts, group)
        return mhc.Model.from_device_counts(local_counts.data_ptr(), 2, stream)
    per = nctx // world
    c0, c1 = rank * per, (rank + 1) * per
    mine = torch.empty(per * 256, dtype=torch.int64, device=local_counts.device)
    if staged:
        host = local_counts.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        mine.copy_(host[c0 * 256:c1 * 256])
    else:
        dist.reduce_scatter_tensor(mine, local_counts, op=dist.ReduceOp.SUM, group=group)
    wsb = int(lib.mh_dev_model2_workspace())
    ws = torch.empty(wsb, dtype=torch.uint8, device=local_counts.device)
    rc = lib.mh_dev_model2_build_slice(mine.data_ptr(), c0, c1, ws.data_ptr(), wsb, stream)
    if rc != 0:
        raise mhc.MhError(rc, "mh_dev_model2_build_slice")
    off, stride = C.c_size_t(), C.c_size_t()
    for which in range(7):                       # every array is laid out by context: a rank's share is one contiguous range
        rc = lib.mh_dev_model2_array(which, C.byref(off), C.byref(stride))
        if rc != 0:
            raise mhc.MhError(rc, "mh_dev_model2_array")
        whole = ws[off.value:off.value + nctx * stride.value]
        part = whole[c0 * stride.value:c1 * stride.value]
        if staged:
            hw, hp = whole.cpu(), part.cpu()
            dist.all_gather_into_tensor(hw, hp, group=group)
            whole.copy_(hw)
        else:
            dist.all_gather_into_tensor(whole, part, group=group)      # in place: the input is the rank's own range of the output
    h = C.c_void_p()
    rc = lib.mh_dev_model2_finish(ws.data_ptr(), wsb, stream, C.byref(h))
    if rc != 0:
        raise mhc.MhError(rc, "mh_dev_model2_finish")
    model = mhc.Model(h)
    model._workspace = ws                        # the model borrows it
    return model


def _staged(t, group):
    """True when a collective on tensor `t` has to go through host memory (gloo rehearsals with the shards on a GPU)."""
    return t.is_cuda and dist.get_backend(group) != "nccl"


def all_gather_flat(out, t, group=None):
    """all_gather_into_tensor where the tensors lie (RCCL), or through host copies (gloo)."""
    if _staged(t, group):
        co, ct = out.cpu(), t.cpu()
        dist.all_gather_into_tensor(co, ct, group=group)
        out.copy_(co)
    elif t.is_cuda:
        dist.all_gather_into_tensor(out, t, group=group)
    else:                                                        # gloo on CPU tensors: the list form
        parts = list(out.view(dist.get_world_size(group), -1).unbind(0))
        dist.all_gather(parts, t.view(-1), group=group)
    return out


def global_start_bit(my_bits, all_bits=None, start_bit=None, group=None):
    """All-gather of the shard payload lengths (one int64 per rank) and an exclusive sum: this rank's global start bit.
    Everything stays where `my_bits` lies (no host wait with RCCL).  Returns (start_bit[1], all_bits[world])."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if all_bits is None:
        all_bits = torch.zeros(world, dtype=torch.int64, device=my_bits.device)
    if start_bit is None:
        start_bit = torch.zeros(1, dtype=torch.int64, device=my_bits.device)
    all_gather_flat(all_bits, my_bits, group)
    torch.sum(all_bits[:rank], dim=0, keepdim=True, out=start_bit)
    return start_bit, all_bits


def compress_step(backend, shard, prev0, group=None, distributed=None, mark=None):
    """THE orchestration of one rank's part of a sharded compress — what `bench.py --gpus N` times, what
    `compress_shard` returns to callers, what the world-2/3 tests drive (gloo on CPU with an oracle-backed double, gloo
    and the HIP backend on one card):
        histogram of the shard (context of its first byte = the byte before it, src/main.cpp:32-37)
        -> sum over ranks (the ONE collective on the data path: 512 KiB of counts)
        -> the same model on every rank (integer-only tree build, src/huffman.cpp:131-164)
        -> shard payload bits = local histogram . code lengths -> all-gather -> exclusive sum = global start bit
        -> encode, pre-shifted by start bit % 8 so that the shards' payloads concatenate with one OR-merged seam byte.
    `backend` supplies the per-shard compute (HipBackend below; tests inject doubles): histogram, merge, build_model,
    payload_bits, encode.  `distributed` None = whenever a process group is up.  `mark(name)` is called after each stage
    (bench.py records an event there).  Nothing here waits for the device.
    Returns dict(model, local, my_bits, start_bit, all_bits, encoded)."""
    mark = mark or (lambda name: None)
    if distributed is None:
        distributed = dist.is_available() and dist.is_initialized()
    local = backend.histogram(shard, prev0)                    # int64 counts where the backend computes
    mark("hist")
    merged = local
    if distributed:
        merged = backend.merge(local, group)                   # a second buffer: the shard's own counts fix its payload length
    mark("allreduce")
    model = backend.build_model(merged)
    mark("tree")
    my_bits = start_bit = all_bits = None
    if distributed:
        my_bits = backend.payload_bits(model, local)           # known before encoding: placement first
        start_bit, all_bits = global_start_bit(my_bits, getattr(backend, "all_bits", None), getattr(backend, "start_bit", None), group)
    encoded = backend.encode(model, shard, prev0, start_bit)
    mark("encode")
    return {"model": model, "local": local, "my_bits": my_bits, "start_bit": start_bit, "all_bits": all_bits, "encoded": encoded}


def compress_shard(backend, shard, last_byte, group=None):
    """One rank's part of a sharded compress.  Returns dict(model, payload, nbits, index, prev0, start_bit, total_bits);
    `payload` and `index` are whatever backend.finish() hands out — for HipBackend VIEWS of its device buffers, overwritten
    by the backend's next step: copy them (`.cpu()`, `.clone()`) before calling again.  `payload` holds the shard's codes from bit (start_bit % 8) of its first
    byte on (zero bits before), `nbits` counts the shard's own payload bits."""
    n = backend.length(shard)
    prev0 = exchange_prev0(last_byte, n > 0, group)
    r = compress_step(backend, shard, prev0, group, distributed=True)
    nbits, start = int(r["my_bits"].item()), int(r["start_bit"].item())
    total = int(r["all_bits"].sum().item())
    payload, end_bits, index = backend.finish(r["encoded"])
    if end_bits != (start & 7) + nbits:
        raise RuntimeError("shard payload is %d bits, its histogram predicted %d" % (end_bits - (start & 7), nbits))
    # payload / index are views of the backend's buffers: valid until its next histogram() / encode() (HipBackend.finish)
    return {"model": r["model"], "payload": payload, "nbits": nbits, "index": index, "prev0": prev0,
            "start_bit": start, "total_bits": total}


def stitch(parts, total_bits):
    """parts: iterable of (start_bit, payload bytes as produced by compress_shard).  Returns the single
    stream's payload: every shard is OR-ed in at byte start_bit // 8 (its leading start_bit % 8 bits
    are zero, so the seam byte it shares with its predecessor merges without shifting anything)."""
    out = bytearray((total_bits + 7) // 8)
    for start, payload in sorted(parts, key=lambda p: p[0]):
        payload = bytes(payload)
        if not payload:
            continue
        off = start // 8
        out[off] |= payload[0]                                   # the seam byte shared with the previous shard
        out[off + 1:off + len(payload)] = payload[1:]            # everything after it belongs to this shard alone
    return bytes(out)


class HipBackend:
    """Per-shard compute on the rank's MI355X through the C ABI (device pointers, torch's current stream), for shards of
    at most `n` bytes.  Every buffer a step needs is allocated here, once: a step allocates nothing and waits for the
    device exactly once (the 16 KiB of table sizes the model build needs to pick the decode-table layout).
    The fast flow: the histogram runs in region mode and leaves the regions' pair counts in its
    workspace, the encoder prices its regions from them (no length pass) and writes the device-only fine index beside
    the payload, the tile decoder decodes from it."""

    def __init__(self, mhc, n, device=None, order=1, chunk_symbols=1024, o2_exchange="compact", use_fine=True, two_pass_encode=False):
        self.mhc, self.lib, self.n, self.order, self.chunk = mhc, mhc.lib(), int(n), order, int(chunk_symbols)
        if mhc.device_count() < 1:
            raise mhc.MhError(mhc.MH_ERR_NO_DEVICE, "HipBackend")
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.o2_exchange = o2_exchange
        self.two_pass_encode = two_pass_encode
        dev, n = self.device, self.n
        ncounts = 65536 if order == 1 else 1 << 24
        self.counts = torch.zeros(ncounts, dtype=torch.int64, device=dev)        # the shard's own histogram
        self.merged = None                                                       # the all-reduced one (allocated at first merge)
        self.cap = n + (64 << 20) if n >= (1 << 20) else n * 8 + 4096            # (tiny shards: codes of up to 64 bits)
        self.payload = torch.empty(self.cap, dtype=torch.uint8, device=dev)
        self.decoded = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        self.nidx = (n + self.chunk - 1) // self.chunk
        self.index = torch.empty(max(self.nidx, 1), dtype=torch.int64, device=dev)
        self.nbits = torch.zeros(2, dtype=torch.int64, device=dev)
        # device-only fine index (one uint32 per 64 symbols): what lets a wave decode 64 adjacent pieces from one
        # contiguous piece of the payload (mh_dev_encode_fine / mh_dev_decode_fine)
        self.use_fine = bool(use_fine) and not (order == 1 and two_pass_encode)
        self.fine_symbols = int(os.environ.get("MH_FINE_SYMBOLS", "64"))        # (an experimental library build may use 32)
        self.fine = torch.empty(max((n + self.fine_symbols - 1) // self.fine_symbols, 1), dtype=torch.int32, device=dev) if self.use_fine else None
        # order 1: the region histogram the encoder prices from; order 2: room for the partition path (flat sources)
        self.hist_ws_bytes = int(self.lib.mh_dev_histogram_o2_workspace(n) if order == 2 else self.lib.mh_dev_histogram_workspace(n))
        self.hist_ws = torch.empty(max(self.hist_ws_bytes, 64), dtype=torch.uint8, device=dev)
        self.enc_ws_bytes = int(self.lib.mh_dev_encode_workspace(n))
        self.enc_ws = torch.empty(self.enc_ws_bytes + 64, dtype=torch.uint8, device=dev)
        self.dec_ws_bytes = int(self.lib.mh_dev_decode_workspace(0, n, self.chunk))
        self.dec_ws = torch.empty(max(self.dec_ws_bytes, 64), dtype=torch.uint8, device=dev)
        self.model_ws_bytes = int(self.lib.mh_dev_model_workspace(1))
        self.model_ws = torch.empty(self.model_ws_bytes, dtype=torch.uint8, device=dev)
        self.my_bits = torch.zeros(1, dtype=torch.int64, device=dev)
        self.start_bit = torch.zeros(1, dtype=torch.int64, device=dev)
        self.all_bits = None                                                     # (global_start_bit allocates it: world entries)
        self.nbits_hint = 0          # last known payload length: steers the decode variant choice only
        self.n_now = n               # bytes of the shard the buffers currently describe

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _check(self, rc, what):
        if rc != 0:
            raise self.mhc.MhError(rc, what)

    stream, check = _stream, _check                            # (names the tools under tools/ use)

    def length(self, shard):
        return shard.numel()

    # ---- the five calls compress_step makes -------------------------------------------------------------------
    def histogram(self, shard, prev0, n=None):
        n = shard.numel() if n is None else n
        if n > self.n:
            raise ValueError("shard of %d bytes, buffers for %d" % (n, self.n))
        self.n_now = n
        if self.order == 2:         # extension (parity unpinned): 65536 two-byte contexts, counts in HBM; prev0: 16-bit context
            self._check(self.lib.mh_dev_histogram_o2_ws(shard.data_ptr(), n, prev0, self.counts.data_ptr(),
                                                        self.hist_ws.data_ptr(), self.hist_ws_bytes, self._stream()), "mh_dev_histogram_o2_ws")
        else:                       # region mode: the regions' own pair counts stay in hist_ws for the encoder
            self._check(self.lib.mh_dev_histogram_o1(shard.data_ptr(), n, prev0, self.counts.data_ptr(),
                                                     self.hist_ws.data_ptr(), self.hist_ws_bytes, self._stream()), "mh_dev_histogram_o1")
        return self.counts

    def merge(self, local, group=None):
        if self.merged is None:
            self.merged = torch.empty_like(local)
        self.merged.copy_(local)
        if self.order == 2 and self.o2_exchange == "compact":
            merged_histogram_o2_compact(self.merged, group)   # only the live contexts' rows travel (text: a few MiB of 128)
        elif self.order == 2 and self.o2_exchange == "scatter" and 65536 % dist.get_world_size(group) == 0:
            pass                                              # build_model reduce-scatters the counts itself
        else:
            merged_histogram(self.merged, group)              # the one collective: 512 KiB sum over xGMI (order 2: 128 MiB)
        return self.merged

    def build_model(self, counts=None):
        """Tables built on the device into the preallocated workspace: no allocation, one stream sync."""
        counts = self.counts if counts is None else counts
        if self.order == 2:
            if counts is self.merged and self.o2_exchange == "scatter" and 65536 % dist.get_world_size() == 0:
                return order2_model(self.mhc, counts, self._stream(), exchange="scatter")
            return self.mhc.Model.from_device_counts(counts.data_ptr(), 2, self._stream())   # (allocates its data-dependent tables)
        return self.mhc.Model.from_device_counts_ws(counts.data_ptr(), 1, self.model_ws.data_ptr(), self.model_ws_bytes, self._stream())

    def payload_bits(self, model, counts, out=None):
        out = self.my_bits if out is None else out
        self._check(self.lib.mh_dev_payload_bits(model.handle, counts.data_ptr(), out.data_ptr(), self._stream()), "mh_dev_payload_bits")
        return out

    def encode(self, model, shard, prev0, start_bit=None):
        """start_bit: device int64 tensor holding this shard's global start bit (the payload is emitted pre-shifted by
        its low 3 bits), or None.  Payload, its bit count (nbits[0]), the chunk index and the fine index stay on the device."""
        n = self.n_now
        sb = start_bit.data_ptr() if start_bit is not None else None
        fine = self.fine.data_ptr() if self.use_fine else None
        if self.order == 1 and not self.two_pass_encode:
            # the histogram of this very buffer is in hist_ws: the encoder prices its regions from it (no length pass)
            self._check(self.lib.mh_dev_encode_fine(model.handle, shard.data_ptr(), n, prev0, sb, self.payload.data_ptr(), self.cap,
                                                    self.nbits.data_ptr(), self.index.data_ptr(), self.chunk, fine,
                                                    self.hist_ws.data_ptr(), self.hist_ws_bytes,
                                                    self.enc_ws.data_ptr(), self.enc_ws_bytes, self._stream()), "mh_dev_encode_fine")
        else:
            self._check(self.lib.mh_dev_encode_ctx_fine(model.handle, shard.data_ptr(), n, prev0, sb, self.payload.data_ptr(), self.cap,
                                                        self.nbits.data_ptr(), self.index.data_ptr(), self.chunk, fine,
                                                        self.enc_ws.data_ptr(), self.enc_ws_bytes, self._stream()), "mh_dev_encode_ctx_fine")
        return self

    # ---- beside the orchestration --------------------------------------------------------------------------------
    def decode(self, model):
        """Decodes what encode() left (payload, indices; the payload length stays on the device) into self.decoded."""
        self._check(self.lib.mh_dev_decode_fine(model.handle, self.payload.data_ptr(), self.nbits_hint, self.nbits.data_ptr(),
                                                self.decoded.data_ptr(), self.n_now, self.index.data_ptr(), self.chunk,
                                                self.fine.data_ptr() if self.use_fine else None, self.dec_ws.data_ptr(),
                                                self.dec_ws_bytes, self._stream()), "mh_dev_decode_fine")
        return self.decoded[:self.n_now]

    def finish(self, encoded):
        """Host view of an encode: (payload tensor, bits in it counted from bit 0 of its first byte, chunk index).
        The tensors are VIEWS of this backend's buffers (as are the counts histogram() returns): they hold the step's
        results until the next histogram() / encode() on this backend overwrites them — clone what must outlive that."""
        self._check(self.lib.mh_dev_status(self.hist_ws.data_ptr(), self._stream()), "histogram status")   # counts add up to n
        self._check(self.lib.mh_dev_status(self.enc_ws.data_ptr(), self._stream()), "encode status")
        nb = int(self.nbits[0].item())
        self.nbits_hint = nb
        return self.payload[:(nb + 7) // 8], nb, self.index[:(self.n_now + self.chunk - 1) // self.chunk]

    def statuses(self):
        """(encode, decode, histogram) status words of the last step (0 = fine); synchronises."""
        st = self._stream()
        return (self.lib.mh_dev_status(self.enc_ws.data_ptr(), st), self.lib.mh_dev_status(self.dec_ws.data_ptr(), st),
                self.lib.mh_dev_status(self.hist_ws.data_ptr(), st))     # (order 2: hist2_total_kernel's verdict)

    def paths(self):
        """(encoder, decoder) that ran last on this backend's workspaces (codes of mh_dev_encode_path / mh_dev_decode_path)."""
        st = self._stream()
        return (self.lib.mh_dev_encode_path(self.enc_ws.data_ptr(), st), self.lib.mh_dev_decode_path(self.dec_ws.data_ptr(), st))
