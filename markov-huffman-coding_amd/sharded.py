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
    Returns the Model; it borrows a workspace tensor that is kept alive on the object.  With gloo (rehearsals with the
    shards on one card) the collectives are staged through host memory and the reduce-scatter is an all-reduce + slice
    (gloo has none)."""
    import ctypes as C
    lib = mhc.lib()
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


def global_bit_offsets(local_nbits, group=None):
    """All-gather of the shard payload lengths -> (this rank's global start bit, total bits, all lengths)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.tensor([int(local_nbits)], dtype=torch.int64, device=_dev(group))
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    lens = [int(v.item()) for v in allv]
    return sum(lens[:rank]), sum(lens), lens


def compress_shard(backend, shard, last_byte, group=None):
    """One rank's part of a sharded compress.  Returns dict(model, payload, nbits, index, prev0,
    start_bit, total_bits): `payload` holds the shard's codes from bit (start_bit % 8) of its first
    byte on (zero bits before), `nbits` counts the shard's own payload bits."""
    n = backend.length(shard)
    prev0 = exchange_prev0(last_byte, n > 0, group)
    local = backend.histogram(shard, prev0)
    merged = merged_histogram(local.clone(), group)
    model = backend.build_model(merged)
    nbits = backend.payload_bits(model, local)               # known before encoding: placement first
    start, total, _ = global_bit_offsets(nbits, group)
    payload, end_bits, index = backend.encode(model, shard, prev0, start)
    if end_bits != (start & 7) + nbits:
        raise RuntimeError("shard payload is %d bits, its histogram predicted %d" % (end_bits - (start & 7), nbits))
    return {"model": model, "payload": payload, "nbits": nbits, "index": index, "prev0": prev0,
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
    """Per-shard compute on the rank's MI355X through the C ABI (device pointers, current stream)."""

    def __init__(self, mhc, chunk_symbols=1024):
        self.mhc, self.lib, self.chunk = mhc, mhc.lib(), chunk_symbols
        if mhc.device_count() < 1:
            raise mhc.MhError(mhc.MH_ERR_NO_DEVICE, "HipBackend")

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _check(self, rc, what):
        if rc != 0:
            raise self.mhc.MhError(rc, what)

    def length(self, shard):
        return shard.numel()

    def histogram(self, shard, prev0):
        counts = torch.zeros(65536, dtype=torch.int64, device=shard.device)
        wsb = int(self.lib.mh_dev_histogram_workspace(shard.numel()))
        ws = torch.empty(wsb, dtype=torch.uint8, device=shard.device)
        self._check(self.lib.mh_dev_histogram_o1(shard.data_ptr(), shard.numel(), prev0, counts.data_ptr(), ws.data_ptr(), wsb,
                                                 self._stream()), "mh_dev_histogram_o1")
        return counts

    def build_model(self, counts):
        return self.mhc.Model.from_device_counts(counts.data_ptr(), 1, self._stream())

    def payload_bits(self, model, counts):
        out = torch.zeros(1, dtype=torch.int64, device=counts.device)
        self._check(self.lib.mh_dev_payload_bits(model.handle, counts.data_ptr(), out.data_ptr(), self._stream()),
                    "mh_dev_payload_bits")
        return int(out.item())

    def encode(self, model, shard, prev0, start_bit=0):
        n = shard.numel()
        cap = self.lib.mh_encode_bound(model.handle, n) + 16
        payload = torch.empty(cap, dtype=torch.uint8, device=shard.device)
        nbits = torch.zeros(1, dtype=torch.int64, device=shard.device)
        start = torch.tensor([int(start_bit)], dtype=torch.int64, device=shard.device)
        index = torch.empty(max((n + self.chunk - 1) // self.chunk, 1), dtype=torch.int64, device=shard.device)
        wsb = self.lib.mh_dev_encode_workspace(n)
        ws = torch.empty(wsb + 64, dtype=torch.uint8, device=shard.device)
        self._check(self.lib.mh_dev_encode_at(model.handle, shard.data_ptr(), n, prev0, start.data_ptr(), payload.data_ptr(), cap,
                                              nbits.data_ptr(), index.data_ptr(), self.chunk, ws.data_ptr(), wsb,
                                              self._stream()), "mh_dev_encode_at")
        self._check(self.lib.mh_dev_status(ws.data_ptr(), self._stream()), "encode status")
        nb = int(nbits.item())
        return payload[:(nb + 7) // 8], nb, index[:(n + self.chunk - 1) // self.chunk]

    def decode(self, model, payload, nbits, index, n):
        out = torch.empty(max(n, 1), dtype=torch.uint8, device=payload.device)
        wsb = int(self.lib.mh_dev_decode_workspace(nbits, n, self.chunk))
        ws = torch.empty(wsb, dtype=torch.uint8, device=payload.device)
        self._check(self.lib.mh_dev_decode(model.handle, payload.data_ptr(), nbits, out.data_ptr(), n, index.data_ptr(),
                                           self.chunk, ws.data_ptr(), wsb, self._stream()), "mh_dev_decode")
        self._check(self.lib.mh_dev_status(ws.data_ptr(), self._stream()), "decode status")
        return out[:n]
