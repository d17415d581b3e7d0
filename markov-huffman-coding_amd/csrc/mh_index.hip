// mh_index.hip — the index builder for streams that come without an index, i.e. the reference's own files
// (src/coding.cpp:35-59): the segment iteration and its map fallbacks, the bookkeeping of the fast path
// (the fast path's own kernels are mh_tile.hip's index_tile_kernel) and launch_build_index, which picks between them.
#include "mh_decode_dev.hpp"

namespace mhk {

// ---- index builder for streams that come without an index ---------------------------------------
// The reference's stream has no index (src/coding.cpp:35-59) and the decoder state is (bit position,
// previous byte).  Parallel reconstruction by fixed-point iteration over bit segments of seg_bits
// bits: segment i's start state is segment i-1's end state; every segment starts from a guess and is
// re-decoded whenever its predecessor's end state changes.  Segment 0 is exact after pass 0, and
// Huffman streams re-synchronise after a few symbols, so a handful of passes converge; a pass that
// recomputes nothing proves the fixed point (= the true decode).  Then a prefix sum of the symbol
// counts and one more pass emit the regular chunk index.
__device__ __forceinline__ uint64_t st_pack(uint32_t prev, uint64_t pos) { return (uint64_t(prev) << 56) | pos; }
// a state = context | bit position, packed like an index entry: order 1 context << 56, order 2 (two bytes) << 48
__device__ __forceinline__ uint32_t st_shift(const IdxParams &p) { return p.order == 2 ? 48u : 56u; }
__device__ __forceinline__ uint64_t st_make(const IdxParams &p, uint32_t ctx, uint64_t pos) { return (uint64_t(ctx) << st_shift(p)) | pos; }
__device__ __forceinline__ uint64_t st_pos(const IdxParams &p, uint64_t s) { return s & ((1ull << st_shift(p)) - 1ull); }
__device__ __forceinline__ uint32_t st_ctx(const IdxParams &p, uint64_t s) { return uint32_t(s >> st_shift(p)); }

// the fine index entry (mh_kernels.h, TileParams) of symbol number g, when g starts a 64-symbol sub-chunk: the fill
// passes know every symbol's context and position, so a stream that came without any index gets the tile decoder too
__device__ __forceinline__ void idx_fine_entry(const IdxParams &p, uint64_t g, uint32_t prev, uint64_t pos) {
    if (p.fine && p.order != 2 && (g & ((1u << T_SUB_SHIFT) - 1u)) == 0 && (g >> T_SUB_SHIFT) < p.fine_cap)
        p.fine[g >> T_SUB_SHIFT] = (prev << 24) | (uint32_t(pos) & FINE_POS_MASK);
}

// Decodes from `start` until the bit position reaches seg_end.  Returns the end state; *count = symbols
// whose code starts before seg_end.  ON_SYMBOL(k, prev_before, pos_before) is called per symbol.
// A null table entry stops the walk (*bad): speculative starts may run into one legitimately.
template <typename F>
__device__ __forceinline__ uint64_t walk_segment(const IdxParams &p, const DecTables &tabs, const BitSrc &src, uint64_t start,
                                                 uint64_t seg_end, uint32_t &count, bool &bad, F on_symbol, uint32_t *max_used = nullptr) {
    uint64_t pos = st_pos(p, start);
    uint32_t prev = st_ctx(p, start);
    count = 0;
    bad = false;
    if (max_used) *max_used = 0;
    if (pos >= seg_end) return start;
    GranuleCursor bc;                                       // (dword reads cost a cache line each here: lanes are a segment apart)
    bc.init(src, pos);
    while (pos < seg_end) {
        on_symbol(count, prev, pos);
        uint32_t used = 0;
        uint32_t sym = decode_one(p.prim, p.sec_base, tabs, src, bc, prev, used, bad);
        if (bad) break;
        if (max_used && used > *max_used) *max_used = used;     // (the longest code of the segment)
        prev = p.order == 2 ? (((prev << 8) | sym) & 0xFFFFu) : sym;
        pos += used;
        ++count;
    }
    return st_make(p, prev, pos);
}

// `first`: first pass of an instance (every segment starts from its guess, at bit i * seg_bits + phase);
// later passes re-decode only the segments whose predecessor's end state has changed.
__global__ __launch_bounds__(256) void index_sync_kernel(IdxParams p, uint32_t iter, uint32_t first, uint32_t phase) {
    if (!first && p.changed[iter - 1] == 0) return;                    // already at the fixed point
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= p.nseg) return;
    const uint64_t seg_end = ((i + 1) * p.seg_bits) < p.nbits ? ((i + 1) * p.seg_bits) : p.nbits;
    uint64_t start;
    if (i == 0) start = st_make(p, p.prev0, 0);
    else if (first) start = st_make(p, p.order == 2 ? 0x2020u : 0x20u, i * p.seg_bits + phase);
    else start = __hip_atomic_load(&p.seg_end_state[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!first && start == p.seg_used[i]) return;                      // same input as last time
    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    const DecTables tabs{p.sec, p.tree, p.P, p.direct, p.H};
    uint32_t count;
    bool bad;
    uint64_t end = walk_segment(p, tabs, src, start, seg_end, count, bad, [](uint32_t, uint32_t, uint64_t) {});
    if (bad) end = st_make(p, st_ctx(p, end), seg_end);                // a guess that ran into nothing: park it
    __hip_atomic_store(&p.seg_end_state[i], (unsigned long long)end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p.seg_used[i] = start;
    p.seg_count[i] = count;
    atomicAdd(&p.changed[iter], 1u);
}

// adds the block offsets of the two-level scan (seg_sym_start holds block-local prefixes)
__global__ __launch_bounds__(SCAN_THREADS) void index_scan_add_kernel(unsigned long long *seg_sym_start, const unsigned long long *blk_sum,
                                                                      uint64_t nseg, uint64_t nblk, unsigned long long *n_symbols) {
    const unsigned long long boff = blk_sum[blockIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_symbols = blk_sum[nblk];
    uint64_t i0 = uint64_t(blockIdx.x) * SCAN_BLOCK + uint64_t(threadIdx.x) * SCAN_PER_THREAD;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k)
        if (i0 + k < nseg) seg_sym_start[i0 + k] += boff;
}

// final pass: true start states are known; write one index entry per chunk_symbols symbols
__global__ __launch_bounds__(256) void index_fill_kernel(IdxParams p) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= p.nseg) return;
    const uint64_t seg_end = ((i + 1) * p.seg_bits) < p.nbits ? ((i + 1) * p.seg_bits) : p.nbits;
    const uint64_t start = i == 0 ? st_make(p, p.prev0, 0) : p.seg_end_state[i - 1];
    const uint64_t base = p.seg_sym_start[i];
    const uint64_t smask = (1ull << p.chunk_shift) - 1;
    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    const DecTables tabs{p.sec, p.tree, p.P, p.direct, p.H};
    uint32_t count;
    bool bad, overflow = false;
    uint64_t end = walk_segment(p, tabs, src, start, seg_end, count, bad, [&](uint32_t k, uint32_t prev, uint64_t pos) {
        const uint64_t g = base + k;
        if ((g & smask) == 0) {
            const uint64_t ci = g >> p.chunk_shift;
            if (ci < p.index_cap) p.index[ci] = st_make(p, prev, pos); else overflow = true;
        }
        idx_fine_entry(p, g, prev, pos);
    });
    if (overflow) atomicExch(p.status, MHK_STATUS_CAPACITY);
    // with true start states a null entry, a mismatch with the converged end state, or a stream that does
    // not end exactly at nbits (src/coding.cpp:124,158) means the stream does not belong to this table
    if (bad || end != p.seg_end_state[i] || count != p.seg_count[i] || (i + 1 == p.nseg && st_pos(p, end) != p.nbits))
        atomicExch(p.status, MHK_STATUS_CORRUPT);
}

// ---- the fast path's bookkeeping (mh_tile.hip, index_tile_kernel) ------------------------------------------------------
// A segment is in order when it was entered in the state its predecessor ended in.  The others are listed (one thread per
// segment; a wave appends its lanes' numbers with one atomic) ...
__global__ __launch_bounds__(256) void index_tile_dirty_kernel(IdxParams p) {
    // every wave owns a strided share of the segments, counts its share first and reserves room for all of it with ONE
    // atomic (an atomic per 64 segments on one address took 10 ms with one segment in seven to list), then lists it
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, nwaves = (uint64_t(gridDim.x) * blockDim.x) >> 6;
    auto is_dirty = [&](uint64_t i) -> bool {
        if (i >= p.nseg5) return false;
        const uint32_t pe = i ? uint32_t(p.e16[i - 1]) : (p.prev0 << 8);
        const uint32_t s = p.s16[i];
        return s != pe || s == IX_INVALID;
    };
    uint32_t mine = 0;
    for (uint64_t i0 = wave * 64u; i0 < p.nseg5; i0 += nwaves * 64u) mine += uint32_t(__popcll(__ballot(is_dirty(i0 + lane))));
    if (mine == 0) return;                                       // (wave-uniform)
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(&p.changed[p.iter], mine);
    at = uint32_t(__builtin_amdgcn_readfirstlane(int(at)));
    for (uint64_t i0 = wave * 64u; i0 < p.nseg5; i0 += nwaves * 64u) {
        const bool d = is_dirty(i0 + lane);
        const unsigned long long m = __ballot(d);
        if (d) {
            const uint32_t slot = at + uint32_t(__popcll(m & ((1ull << lane) - 1ull)));
            if (slot < p.dirty_cap) p.dirty_list[slot] = uint32_t(i0 + lane);
        }
        at += uint32_t(__popcll(m));
    }
}

// ... and decoded again from that state with the general tables from memory, one thread per listed segment (after the
// warm-up pass: one segment in 10^5 for an iid-like source, one in seven for text).  A pass that lists nothing proves the
// fixed point.
__global__ __launch_bounds__(256) void index_tile_repair_kernel(IdxParams p, uint32_t count) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const uint64_t i = p.dirty_list[j];
    const uint32_t pe = i ? uint32_t(__hip_atomic_load(&p.e16[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : (p.prev0 << 8);
    const uint64_t seg_end = (i + 1) * IX_SEG_BITS < p.nbits ? (i + 1) * IX_SEG_BITS : p.nbits;
    const uint64_t start = st_make(p, pe >> 8, i * IX_SEG_BITS + (pe & 255u));
    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    const DecTables tabs{p.sec, p.tree, p.P, p.direct, p.H};
    uint32_t count_sym, longest;
    bool bad;
    uint64_t end = walk_segment(p, tabs, src, start, seg_end, count_sym, bad, [](uint32_t, uint32_t, uint64_t) {}, &longest);
    if (bad) end = st_make(p, st_ctx(p, end), seg_end);                // (the fill pass reports it if the state was the true one)
    const uint64_t over = st_pos(p, end) - seg_end;
    __hip_atomic_store(&p.e16[i], uint16_t((st_ctx(p, end) << 8) | uint32_t(over > 254 ? 254 : over)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // [r5] bit 15 of the count: the segment holds a code the tile tables' two levels do not resolve (longer than tP + tH bits) —
    // such a segment is always listed (the states pass cannot decode it) and the segment decoder leaves it to
    // segment_walk_emit_kernel.  (A segment has at most IX_SEG_BITS symbols: the count itself needs nine bits.)
    p.c16[i] = uint16_t(count_sym | (longest > p.tP + p.tH ? IX_C16_WALK : 0u));
    p.s16[i] = uint16_t(pe == IX_INVALID ? 0xFFFEu : pe);              // (an end state is never IX_INVALID: its overshoot is under 255)
}

// [r5] ... and from the second pass on only where something can have changed: a segment is out of order after a pass only if it
// was repaired in that pass from a state that has changed meanwhile, or if its predecessor was (whose end state it must
// follow).  So the next list is made from the previous one — one thread per entry checks the segment itself and its successor —
// instead of reading all the segments' states again (1.1 GB per pass at 16 GiB: 0.39 ms each; a stream takes two or three
// such passes, text six).  A segment can be listed twice (as itself and as its predecessor's successor): it is then repaired
// twice to the same result.
__global__ __launch_bounds__(256) void index_tile_dirty_next_kernel(IdxParams p, const uint32_t *prev_list, uint32_t prev_count) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    auto is_dirty = [&](uint64_t i) -> bool {
        if (i >= p.nseg5) return false;
        const uint32_t pe = i ? uint32_t(p.e16[i - 1]) : (p.prev0 << 8);
        const uint32_t s = p.s16[i];
        return s != pe || s == IX_INVALID;
    };
    uint64_t i = 0;
    bool d0 = false, d1 = false;
    if (j < prev_count) {
        i = prev_list[j];
        d0 = is_dirty(i);
        d1 = is_dirty(i + 1);
    }
    const uint32_t mine = (d0 ? 1u : 0u) + (d1 ? 1u : 0u);
    uint32_t inc = mine;                                          // one atomic per wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if ((threadIdx.x & 63u) >= uint32_t(d)) inc += o; }
    const uint32_t total = __shfl(inc, 63);
    if (total == 0) return;
    uint32_t at = 0;
    if ((threadIdx.x & 63u) == 63u) at = atomicAdd(&p.changed[p.iter], total);
    at = __shfl(at, 63) + inc - mine;
    if (d0) { if (at < p.dirty_cap) p.dirty_list[at] = uint32_t(i); ++at; }
    if (d1 && at < p.dirty_cap) p.dirty_list[at] = uint32_t(i + 1);
}

// symbols per tile of IX_TILE_SEGS segments (the input of the prefix sum): sixteen threads per tile, eight counts each
__global__ __launch_bounds__(256) void index_tile_count_kernel(IdxParams p) {
    static_assert(IX_TILE_SEGS == 128, "sixteen 16-byte vectors of counts per tile");
    const uint32_t sub = threadIdx.x & 15u;
    for (uint64_t t = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 4; t < ((p.ntile5 + 15u) & ~uint64_t(15)); t += (uint64_t(gridDim.x) * blockDim.x) >> 4) {
        uint32_t sum = 0;
        if (t < p.ntile5) {
            const uint64_t s0 = t * IX_TILE_SEGS + sub * 8u;
            const uint4 v = *reinterpret_cast<const uint4 *>(p.c16 + s0);      // (the array is 64-byte aligned and padded: mh_index.hip idx_ws_layout)
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (s0 + uint32_t(k) < p.nseg5) sum += (w[k >> 1] >> (16 * (k & 1))) & IX_C16_COUNT;
        }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        if (sub == 0 && t < p.ntile5) p.tile_cnt[t] = sum;
    }
}

// ---- streams the segment iteration cannot synchronise: fixed-length codes with context-dependent assignment
// Two decodes that start in different contexts only ever agree again if they happen to produce the same symbol;
// with few symbols they may never (ABCABC...: every context has one successor; 0/1 data whose two contexts map
// the bit to opposite symbols), and the iteration then repairs one segment per pass.  But when EVERY code of
// every live context has the same length g (the first symbol's context aside), positions are known without
// decoding — symbol k >= 1 starts at l0 + (k - 1) g — and only the context chain is missing.  That chain is a
// composition of maps "context at the start of a group of 2^20 symbols -> context at its end":
//   index_ctx_scan_kernel   per context: is it live, its one code length (0: mixed or longer than P), does it
//                           emit the stream's start context
//   index_group_map_kernel  one thread per (group, live start context): the group's end context
//   index_group_chain_kernel one thread: the true start context of every group
//   index_group_fill_kernel one thread per group: the index entries of its chunks
// Work: (live contexts + 1) x one decode, instead of a one-lane walk of the whole payload (1 GiB of "ABC": 223 s).
__global__ __launch_bounds__(256) void index_ctx_scan_kernel(IdxParams p, uint32_t *info) {
    const uint32_t c = threadIdx.x, span = 1u << p.P;
    // a leaf entry carries its code length in bits 8..12; length 0 is the null entry of a context without codes
    uint32_t live = 0, len = 0, mixed = 0, emits = 0, all_leaf = 1;
    for (uint32_t w = 0; w < span; ++w) {
        const uint32_t e = p.prim[(c << p.P) | w];
        const uint32_t l = (e >> 8) & 31u;
        if ((e & DEC16_LEAF) && l != 0) {
            live = 1;
            if (len == 0) len = l; else if (l != len) mixed = 1;
            if ((e & 255u) == p.prev0) emits = 1;
        } else {
            all_leaf = 0;                                        // null, or an inner entry (codes longer than P)
        }
    }
    info[c] = live | ((live && !mixed && all_leaf) ? len << 8 : 0u) | (emits << 16);
}
struct IdxFixed { const uint8_t *live; uint32_t nlive, l0, g, shift; uint64_t nsym, ngroups; uint8_t *gmap, *gstart; };   // a group = 1 << shift symbols
__device__ __forceinline__ uint64_t idx_fixed_pos(const IdxFixed &f, uint64_t k) { return k == 0 ? 0 : f.l0 + (k - 1) * uint64_t(f.g); }
__global__ __launch_bounds__(256) void index_group_map_kernel(IdxParams p, IdxFixed f) {
    const uint64_t t = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= f.ngroups * f.nlive) return;
    const uint64_t grp = t / f.nlive;
    const uint32_t start_ctx = grp == 0 ? p.prev0 : f.live[t % f.nlive];
    if (grp == 0 && t % f.nlive != 0) return;                   // group 0 starts in the stream's own context only
    const uint64_t k0 = grp << f.shift;
    const uint64_t k1 = (k0 + (1ull << f.shift)) < f.nsym ? k0 + (1ull << f.shift) : f.nsym;
    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    const DecTables tabs{p.sec, p.tree, p.P, p.direct, p.H};
    GranuleCursor bc;
    bc.init(src, idx_fixed_pos(f, k0));
    uint32_t prev = start_ctx;
    bool bad = false;
    for (uint64_t k = k0; k < k1 && !bad; ++k) {
        uint32_t used = 0;
        prev = decode_one(p.prim, p.sec_base, tabs, src, bc, prev, used, bad);
    }
    f.gmap[grp * 256 + start_ctx] = uint8_t(prev);              // a start that runs into a null entry is never the true one
}
__global__ void index_group_chain_kernel(IdxParams p, IdxFixed f) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t s = p.prev0;
    for (uint64_t grp = 0; grp < f.ngroups; ++grp) { f.gstart[grp] = uint8_t(s); s = f.gmap[grp * 256 + s]; }
}
__global__ __launch_bounds__(64) void index_group_fill_kernel(IdxParams p, IdxFixed f) {
    const uint64_t grp = uint64_t(blockIdx.x) * 64 + threadIdx.x;
    if (grp >= f.ngroups) return;
    const uint64_t k0 = grp << f.shift;
    const uint64_t k1 = (k0 + (1ull << f.shift)) < f.nsym ? k0 + (1ull << f.shift) : f.nsym;
    const uint64_t smask = (1ull << p.chunk_shift) - 1;
    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    const DecTables tabs{p.sec, p.tree, p.P, p.direct, p.H};
    GranuleCursor bc;
    uint64_t pos = idx_fixed_pos(f, k0);
    bc.init(src, pos);
    uint32_t prev = f.gstart[grp];
    bool bad = false, overflow = false;
    for (uint64_t k = k0; k < k1 && !bad; ++k) {
        if ((k & smask) == 0) {
            const uint64_t ci = k >> p.chunk_shift;
            if (ci < p.index_cap) p.index[ci] = st_pack(prev, pos); else overflow = true;
        }
        idx_fine_entry(p, k, prev, pos);
        uint32_t used = 0;
        prev = decode_one(p.prim, p.sec_base, tabs, src, bc, prev, used, bad);
        pos += used;
    }
    if (overflow) atomicExch(p.status, MHK_STATUS_CAPACITY);
    // every code had the length the positions assumed, and the stream ends where its last code ends
    if (bad || pos != idx_fixed_pos(f, k1) || (k1 == f.nsym && pos != p.nbits)) atomicExch(p.status, MHK_STATUS_CORRUPT);
    if (k1 == f.nsym) *p.n_symbols = f.nsym;
}

// ---- the same for mixed code lengths: maps over BIT groups from (context, offset of the first code behind the
// group's start) to (context, overshoot past its end, symbols decoded).  A run-structured stream (0...01...12...:
// two successors per context, so decodes from different contexts never merge; the start context ' ' adds a third
// and 2-bit codes, so positions are not arithmetic) costs (live contexts x longest code) decodes here instead of
// the one-lane walk.  Map entry: end context | overshoot << 8 | symbols << 16, bit 63 = ran into a null entry.
struct IdxState { const uint8_t *live; const uint8_t *inv; uint32_t nlive, maxlen, shift; uint64_t ngroups;
                  unsigned long long *map; uint16_t *gstart; unsigned long long *gbase; };   // a group = 1 << shift bits
constexpr unsigned long long IDX_STATE_BAD = 1ull << 63;
__device__ __forceinline__ unsigned long long idx_state_walk(const IdxParams &p, uint64_t pos, uint64_t end, uint32_t ctx, bool last) {
    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    const DecTables tabs{p.sec, p.tree, p.P, p.direct, p.H};
    GranuleCursor bc;
    bc.init(src, pos);
    uint32_t nsym = 0;
    bool bad = false;
    while (pos < end) {
        uint32_t used = 0;
        ctx = decode_one(p.prim, p.sec_base, tabs, src, bc, ctx, used, bad);
        if (bad) return IDX_STATE_BAD;
        pos += used;
        ++nsym;
    }
    if (last && pos != end) return IDX_STATE_BAD;                // the stream must end with its last code
    return (unsigned long long)(ctx) | ((unsigned long long)(pos - end) << 8) | ((unsigned long long)(nsym) << 16);
}
__global__ __launch_bounds__(256) void index_state_map_kernel(IdxParams p, IdxState f) {
    const uint64_t per = uint64_t(f.nlive) * f.maxlen;
    const uint64_t t = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= f.ngroups * per) return;
    const uint64_t grp = t / per;
    const uint32_t ci = uint32_t((t % per) / f.maxlen), o = uint32_t(t % f.maxlen);
    const uint64_t begin = (grp << f.shift) + o;
    const bool last = grp + 1 == f.ngroups;
    const uint64_t end = last ? p.nbits : (grp + 1) << f.shift;
    f.map[t] = begin < end ? idx_state_walk(p, begin, end, f.live[ci], last) : IDX_STATE_BAD;
}
__global__ void index_state_chain_kernel(IdxParams p, IdxState f) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t ci = f.inv[p.prev0 & 255u], o = 0;
    unsigned long long base = 0;
    for (uint64_t grp = 0; grp < f.ngroups; ++grp) {
        f.gstart[grp] = uint16_t(ci | (o << 8));
        f.gbase[grp] = base;
        const unsigned long long e = (ci < f.nlive && o < f.maxlen) ? f.map[(grp * f.nlive + ci) * f.maxlen + o] : IDX_STATE_BAD;
        if (e & IDX_STATE_BAD) {                                 // the true chain itself runs into nothing: not this table's stream
            atomicExch(p.status, MHK_STATUS_CORRUPT);
            for (uint64_t r = grp + 1; r < f.ngroups; ++r) { f.gstart[r] = 0xFFFFu; f.gbase[r] = base; }
            break;
        }
        base += (e >> 16) & 0xFFFFFFFFull;
        ci = f.inv[e & 255u];                                    // 255 for a symbol without codes of its own: fine at the very end only
        o = uint32_t(e >> 8) & 255u;
    }
    *p.n_symbols = base;
}
__global__ __launch_bounds__(64) void index_state_fill_kernel(IdxParams p, IdxState f) {
    const uint64_t grp = uint64_t(blockIdx.x) * 64 + threadIdx.x;
    if (grp >= f.ngroups || f.gstart[grp] == 0xFFFFu) return;
    const uint32_t ci = f.gstart[grp] & 255u, o = f.gstart[grp] >> 8;
    const bool last = grp + 1 == f.ngroups;
    const uint64_t end = last ? p.nbits : (grp + 1) << f.shift;
    const uint64_t smask = (1ull << p.chunk_shift) - 1;
    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    const DecTables tabs{p.sec, p.tree, p.P, p.direct, p.H};
    GranuleCursor bc;
    uint64_t pos = (grp << f.shift) + o, k = f.gbase[grp];
    bc.init(src, pos);
    uint32_t prev = f.live[ci];
    bool bad = false, overflow = false;
    while (pos < end && !bad) {
        if ((k & smask) == 0) {
            const uint64_t cidx = k >> p.chunk_shift;
            if (cidx < p.index_cap) p.index[cidx] = st_pack(prev, pos); else overflow = true;
        }
        idx_fine_entry(p, k, prev, pos);
        uint32_t used = 0;
        prev = decode_one(p.prim, p.sec_base, tabs, src, bc, prev, used, bad);
        pos += used;
        ++k;
    }
    if (overflow) atomicExch(p.status, MHK_STATUS_CAPACITY);
    if (bad || (last && pos != end)) atomicExch(p.status, MHK_STATUS_CORRUPT);
}

// Sequential fallback (one lane) for streams whose segments refuse to synchronise: walks the whole
// payload once.  The loop condition is the reference's `while(bi < length)` (src/coding.cpp:124).
template <int ORDER>
__global__ __launch_bounds__(64) void build_index_kernel(IdxParams p) {
    // order 1: the first-level table goes to LDS first (the walk is one dependent lookup per symbol: 208 ns from
    // L2, a third of that from LDS), the payload comes through the 32-byte-granule FIFO
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint16_t *prim = p.prim;
    if (ORDER == 1) {
        uint16_t *lut = reinterpret_cast<uint16_t *>(smem);
        for (uint32_t k = threadIdx.x; k < (256u << p.P) / 8u; k += 64u)
            reinterpret_cast<uint4 *>(lut)[k] = reinterpret_cast<const uint4 *>(p.prim)[k];
        __syncthreads();
        prim = lut;
    }
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    const DecTables tabs{p.sec, p.tree, p.P, p.direct, p.H};
    GranuleCursor bc;
    bc.init(src, 0);
    uint64_t bi = 0, nsym = 0;
    uint32_t prev = p.prev0;                                // order 2: the 16-bit context
    const uint64_t S = 1ull << p.chunk_shift;
    bool bad = false;
    while (bi < p.nbits) {
        if ((nsym & (S - 1)) == 0) {
            uint64_t ci = nsym >> p.chunk_shift;
            if (ci >= p.index_cap) { atomicExch(p.status, MHK_STATUS_CAPACITY); break; }
            p.index[ci] = (uint64_t(prev) << (ORDER == 2 ? 48 : 56)) | bi;
        }
        idx_fine_entry(p, nsym, prev, bi);
        uint32_t used = 0;
        const uint32_t sym = decode_one(prim, p.sec_base, tabs, src, bc, prev, used, bad);
        if (bad) break;
        prev = ORDER == 2 ? (((prev << 8) | sym) & 0xFFFFu) : sym;
        bi += used;
        ++nsym;
    }
    if (bad || bi > p.nbits) atomicExch(p.status, MHK_STATUS_CORRUPT);
    *p.n_symbols = nsym;
}


// workspace: [0,64) status | changed u32[IDX_MAX_PASSES] | end_state u64[nseg] | used u64[nseg] |
//            count u32[nseg] | sym_start u64[nseg] | blk_sum u64[nblk + 1]
// Segments are ~4096 bits (512 bytes) long, rounded DOWN to a multiple of the gcd g of the model's code
// lengths: when every code length is a multiple of g > 1 (fixed-length codes of 3, 5, 6, 7 bits: a
// near-uniform 8-, 32-, 64- or 128-symbol alphabet such as base64 text), code boundaries only occur at
// multiples of g, and a guessed start that is off that lattice can never re-synchronise: each pass would
// then fix a single segment.  With the guesses on the lattice such streams synchronise at once.
constexpr uint32_t IDX_SEG_BITS = 4096;
constexpr uint32_t IDX_SEG_BITS_MIN = IDX_SEG_BITS - 64;       // smallest segment any gcd <= 64 gives (workspace sizing)
struct IdxWs { size_t off_changed, off_end, off_used, off_count, off_start, off_blk, total; uint64_t nseg, nblk;
               size_t off_e16, off_s16, off_c16, off_dirty, off_dirty2, off_tcnt, off_tbase, off_tblk; uint64_t nseg5, ntile5, ntblk, dirty_cap; };
static IdxWs idx_ws_layout(uint64_t nbits) {
    IdxWs w;
    w.nseg = (nbits + IDX_SEG_BITS_MIN - 1) / IDX_SEG_BITS_MIN;  // capacity; the launch uses the model's segment length
    w.nblk = (w.nseg + SCAN_BLOCK - 1) / SCAN_BLOCK;
    auto up = [](size_t v) { return (v + 63) & ~size_t(63); };
    w.off_changed = 64;
    w.off_end = up(w.off_changed + IDX_MAX_PASSES * 4);
    w.off_used = up(w.off_end + size_t(w.nseg) * 8);
    w.off_count = up(w.off_used + size_t(w.nseg) * 8);
    w.off_start = up(w.off_count + size_t(w.nseg) * 4);
    w.off_blk = up(w.off_start + size_t(w.nseg) * 8);
    w.total = up(w.off_blk + size_t(w.nblk + 1) * 8);
    // the fast path (index_tile_kernel): 6 bytes per IX_SEG_BITS-bit segment, a list of the segments to repair (a quarter of them at
    // most: beyond that the stream does not synchronise this way) and 12 bytes per tile, in the same space (one path runs at a time)
    w.nseg5 = (nbits + IX_SEG_BITS - 1) / IX_SEG_BITS;
    w.ntile5 = (w.nseg5 + IX_TILE_SEGS - 1) / IX_TILE_SEGS;
    w.ntblk = (w.ntile5 + SCAN_BLOCK - 1) / SCAN_BLOCK;
    w.dirty_cap = w.nseg5 / 4 + 64;
    w.off_e16 = w.off_end;
    w.off_s16 = up(w.off_e16 + size_t(w.nseg5) * 2);
    w.off_c16 = up(w.off_s16 + size_t(w.nseg5) * 2);
    w.off_dirty = up(w.off_c16 + size_t(w.nseg5) * 2 + 64);       // (+ 64: the count kernel reads whole 16-byte vectors of counts)
    w.off_dirty2 = up(w.off_dirty + size_t(w.dirty_cap) * 4);     // [r5] the lists of two consecutive passes
    w.off_tcnt = up(w.off_dirty2 + size_t(w.dirty_cap) * 4);
    w.off_tbase = up(w.off_tcnt + size_t(w.ntile5) * 4);
    w.off_tblk = up(w.off_tbase + size_t(w.ntile5) * 8);
    const size_t total5 = up(w.off_tblk + size_t(w.ntblk + 1) * 8);
    if (total5 > w.total) w.total = total5;
    return w;
}
size_t build_index_workspace_bytes(uint64_t nbits) { return idx_ws_layout(nbits).total; }

// which way the index was built (status block of the workspace, bytes 8..11; mh_dev_index_path): tests tell the
// fallbacks apart by this, not by the clock
static void note_index_path(unsigned char *ws, uint32_t path, hipStream_t st) {
    (void)launch_set_word(reinterpret_cast<uint32_t *>(ws + 8), path, st);
}

// ---- the fast path's first pass (index_tile_kernel mode 0 + repairs + prefix sums), shared by the index builder and the
// segment decoder [r5]: on success (*ok) q holds the workspace's arrays — e16 / c16 = every segment's converged end state
// and symbol count, tile_base = the number of the first symbol of every tile, *p.n_symbols = the stream's symbol count.
// Eligible: order 1, tile tables with P = 7, no code-length lattice (g == 1), a stream worth a launch of 256 workgroups.
// Given up (*ok false, nothing else changed but the status block) when more than a quarter of the segments did not
// synchronise within their warm-up, or the repairs do not die out.  Synchronises `st` between passes.
// allow_long: codes longer than the tile tables resolve are the caller's business (the segment decoder hands their segments
// to a walk; the index fill pass of the tiles cannot)
static bool index_tiles_eligible(const IdxParams &p, const IdxWs &L, uint32_t g, bool allow_long = false) {
    return p.order != 2 && p.tprim && p.tP == 7 && (allow_long || p.max_len <= p.tP + p.tH) && g == 1 && p.nbits >= (1ull << 20) && L.nseg5 < 0xFFFFFFFFull && !getenv("MH_INDEX_NO_TILES");
}
static hipError_t index_tile_states(const IdxParams &p, const IdxWs &L, unsigned char *ws, hipStream_t st, IdxParams &q, bool *ok_out) {
    hipError_t e = hipSuccess;
    q = p;
    q.e16 = reinterpret_cast<uint16_t *>(ws + L.off_e16);
    q.s16 = reinterpret_cast<uint16_t *>(ws + L.off_s16);
    q.c16 = reinterpret_cast<uint16_t *>(ws + L.off_c16);
    q.tile_cnt = reinterpret_cast<uint32_t *>(ws + L.off_tcnt);
    q.tile_base = reinterpret_cast<unsigned long long *>(ws + L.off_tbase);
    q.nseg5 = L.nseg5; q.ntile5 = L.ntile5;
    // The warm-up is short: streams of an iid-like source re-synchronise within a few symbols (4 GiB of Zipf: 256 bits leave
    // one segment in 10^5 for the repairs), text, whose decode depends on the context at every step, leaves one in seven —
    // listed and repaired one thread each, which costs a fraction of a pass either way.  More than a quarter to repair:
    // once more with the longest warm-up; still more: the stream does not synchronise this way.
    q.dirty_list = reinterpret_cast<uint32_t *>(ws + L.off_dirty);
    q.dirty_cap = uint32_t(L.dirty_cap < 0xFFFFFFFFull ? L.dirty_cap : 0xFFFFFFFFull);
    const uint64_t dwant = (q.nseg5 + 255) / 256;
    const unsigned dgrid = unsigned(dwant < 2048 ? dwant : 2048);
    bool ok = false;
    uint32_t it = 0;
    q.warm_bits = 256;
    if (const char *wv = getenv("MH_INDEX_WARM_BITS")) { const int v = atoi(wv); if (v >= 16 && v <= int(IX_WARM_BITS_MAX)) q.warm_bits = uint32_t(v); }
    else if (q.ntile5 >= 16384) {
        // how long a warm-up this stream needs is a property of the source: a sample (the first 1/256 of the tiles, one tile
        // per wave of the card at least) with 96 bits tells — under 2.5 % of the segments to repair: 96 bits serve the whole
        // stream (the pass decodes warm-up + segment); under 10 %: 128; else the default 256 (text, whose decode depends on the
        // context at every step).  [r5] 4 GiB of Zipf(1.1), ms per states pass: 64 bits 8.4 (5 % to repair), 96 7.95, 128 8.1, 256 9.4
        IdxParams sq = q;
        sq.ntile5 = q.ntile5 / 256 > 4096 ? q.ntile5 / 256 : 4096;
        sq.nseg5 = sq.ntile5 * IX_TILE_SEGS;
        sq.warm_bits = 96;
        sq.iter = it;
        e = launch_index_tile(sq, 0, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(index_tile_dirty_kernel, dim3(256), dim3(256), 0, st, sq);
        unsigned int dirty = ~0u;
        e = hipMemcpyAsync(&dirty, q.changed + it, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
        ++it;
        if (uint64_t(dirty) * 40u < sq.nseg5) q.warm_bits = 96;
        else if (uint64_t(dirty) * 10u < sq.nseg5) q.warm_bits = 128;
    }
    for (int attempt = 0; attempt < 2 && !ok; ++attempt) {
        e = launch_index_tile(q, 0, st);
        if (e != hipSuccess) return e;
        unsigned int prev_dirty = ~0u;
        bool hopeless = false;
        uint32_t *lists[2] = {reinterpret_cast<uint32_t *>(ws + L.off_dirty), reinterpret_cast<uint32_t *>(ws + L.off_dirty2)};
        uint32_t which = 0;
        for (const uint32_t it_end = it + 24u; it < it_end && !ok && !hopeless; ++it) {
            q.iter = it;
            q.dirty_list = lists[which];
            // the first pass reads every segment's state; so do the later ones while the list is long (runs of listed neighbours,
            // as text has them, would be listed twice by the incremental form: 4 GiB of text 9.95 against 9.5 ms)
            if (prev_dirty == ~0u || uint64_t(prev_dirty) * 64u > q.nseg5) hipLaunchKernelGGL(index_tile_dirty_kernel, dim3(dgrid), dim3(256), 0, st, q);
            else hipLaunchKernelGGL(index_tile_dirty_next_kernel, dim3((prev_dirty + 255u) / 256u), dim3(256), 0, st, q, lists[which ^ 1u], prev_dirty);
            which ^= 1u;
            unsigned int dirty = 1;
            e = hipMemcpyAsync(&dirty, q.changed + it, 4, hipMemcpyDeviceToHost, st);
            if (e != hipSuccess) return e;
            e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
            ok = dirty == 0;
            // too many to list, or repairs that do not die out (fewer than a quarter fewer per pass)
            hopeless = dirty > q.dirty_cap || (dirty > 4096u && prev_dirty != ~0u && uint64_t(dirty) * 4u > uint64_t(prev_dirty) * 3u);
            prev_dirty = dirty;
            if (!ok && !hopeless) hipLaunchKernelGGL(index_tile_repair_kernel, dim3((dirty + 255u) / 256u), dim3(256), 0, st, q, dirty);
        }
        if (!ok && q.warm_bits < IX_WARM_BITS_MAX) q.warm_bits = IX_WARM_BITS_MAX; else break;
    }
    *ok_out = ok;
    if (ok) {
        unsigned long long *tblk = reinterpret_cast<unsigned long long *>(ws + L.off_tblk);
        { const uint64_t cw = (q.ntile5 + 15) / 16; hipLaunchKernelGGL(index_tile_count_kernel, dim3(unsigned(cw < 1 ? 1 : (cw > 8192 ? 8192 : cw))), dim3(256), 0, st, q); }
        (void)launch_scan_local(q.tile_cnt, q.ntile5, q.tile_base, tblk, st);
        (void)launch_scan_top(tblk, L.ntblk, nullptr, st);
        hipLaunchKernelGGL(index_scan_add_kernel, dim3(unsigned(L.ntblk)), dim3(SCAN_THREADS), 0, st, q.tile_base, tblk, q.ntile5, L.ntblk, q.n_symbols);
    }
    return hipGetLastError();
}

// Synchronises `st` between batches of passes (the pass count depends on the data).
hipError_t launch_build_index(IdxParams p, void *d_ws, hipStream_t st) {
    unsigned char *ws = static_cast<unsigned char *>(d_ws);
    const IdxWs L = idx_ws_layout(p.nbits);
    p.status = reinterpret_cast<int *>(ws);
    p.changed = reinterpret_cast<unsigned int *>(ws + L.off_changed);
    p.seg_end_state = reinterpret_cast<unsigned long long *>(ws + L.off_end);
    p.seg_used = reinterpret_cast<unsigned long long *>(ws + L.off_used);
    p.seg_count = reinterpret_cast<uint32_t *>(ws + L.off_count);
    p.seg_sym_start = reinterpret_cast<unsigned long long *>(ws + L.off_start);
    const uint32_t g = p.len_gcd >= 1 && p.len_gcd <= 64 ? p.len_gcd : 1;
    // [r5] On a code-length lattice (g > 1: fixed-length codes assigned by context — uniform bytes, base64 text) two decodes from
    // different contexts merge only when they happen to produce the same symbol: 1/256 per step for random bytes, so a 512-symbol
    // segment's end state still depends on its guess one time in seven and the iteration needs about ten passes.  Eight times
    // the segment: the end state of 4096 symbols is independent of the guess but for one segment in 10^7.
#ifndef MH_IDX_LATTICE_SEG_MULT
#define MH_IDX_LATTICE_SEG_MULT 8
#endif
    const uint32_t seg_target = g > 1 ? IDX_SEG_BITS * MH_IDX_LATTICE_SEG_MULT : IDX_SEG_BITS;
    p.seg_bits = seg_target - seg_target % g;
    p.nseg = (p.nbits + p.seg_bits - 1) / p.seg_bits;          // <= L.nseg
    hipError_t e = hipMemsetAsync(ws, 0, L.off_end, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(p.n_symbols, 0, 8, st);
    if (e != hipSuccess || p.nbits == 0) return e;
    e = once_per_device(&DeviceState::index_ready, [] { return allow_lds(reinterpret_cast<const void *>(build_index_kernel<1>), 131072); });
    if (e != hipSuccess) return e;
    if (p.P > 8) return hipErrorInvalidValue;
    // ---- the fast path (index_tile_states above); when it gives up the segment iteration below starts from scratch
    if (index_tiles_eligible(p, L, g)) {
        IdxParams q;
        bool ok = false;
        e = index_tile_states(p, L, ws, st, q, &ok);
        if (e != hipSuccess) return e;
        if (ok) {
            note_index_path(ws, IDX_PATH_TILES, st);
            return launch_index_tile(q, 1, st);
        }
        e = hipMemsetAsync(ws, 0, L.off_end, st);                     // (status and the pass counters: the iteration starts clean)
        if (e != hipSuccess) return e;
    }
    const unsigned grid = unsigned((p.nseg + 255) / 256);
    const uint64_t nblk = (p.nseg + SCAN_BLOCK - 1) / SCAN_BLOCK;
    // One instance of the iteration = a first pass from guessed starts + passes that chase the changes.
    // g == 1: a single instance with the whole pass budget.  g > 1 (every code length a multiple of g): the
    // guesses of an instance all sit on one residue class i * seg_bits + phase (seg_bits is a multiple of g);
    // the stream's own class is set by whatever came before (e.g. a 1-bit code for the very first symbol,
    // whose context ' ' has a single successor), so the classes are tried in turn with a short budget each;
    // the right one converges in two or three passes.  If none does, the last instance runs on to the pass
    // cap and the sequential walk below is the last resort.
    // An instance is given up when it stops making progress, not after a fixed number of passes: on the wrong
    // residue class every pass re-decodes (nearly) every segment, on the right one the count of changed segments
    // falls geometrically — but how fast depends on the model: 8-bit codes in 256 contexts (random bytes) merge
    // two trajectories with probability 1/256 per symbol, i.e. 86 % per 512-symbol segment, and need ten passes
    // where text needs three (a fixed budget of five sent exactly that case, 1 GiB of random bytes, through all
    // eight classes and then to the one-lane walk: minutes).
    bool converged = false;
    const uint32_t nphase = g > 1 ? (g < 16u ? g : 16u) : 1u;
    uint32_t it = 0, best_phase = 0;
    unsigned int best_changed = ~0u;
    // instances 0 .. nphase - 1 are given up when they stall; instance nphase re-runs the class that got furthest
    // with whatever is left of the pass budget (nphase == 1: the only class runs to the end at once)
    for (uint32_t inst = 0; inst <= nphase && !converged && it < IDX_MAX_PASSES; ++inst) {
        if (inst == nphase && nphase == 1) break;
        const bool to_the_end = nphase == 1 || inst == nphase;
        const uint32_t phase = inst == nphase ? best_phase : inst;
        bool first = true;
        unsigned int prev_changed = 0;                           // changed segments at the end of the previous batch
        while (it < IDX_MAX_PASSES && !converged) {
            const uint32_t batch = first ? 3u : 4u;
            const uint32_t batch_end = it + batch < IDX_MAX_PASSES ? it + batch : IDX_MAX_PASSES;
            for (; it < batch_end; ++it) {
                hipLaunchKernelGGL(index_sync_kernel, dim3(grid), dim3(256), 0, st, p, it, first ? 1u : 0u, phase);
                first = false;
            }
            unsigned int last = 1;
            e = hipMemcpyAsync(&last, p.changed + (it - 1), 4, hipMemcpyDeviceToHost, st);
            if (e != hipSuccess) return e;
            e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
            converged = last == 0;
            // no progress over a whole batch (less than a quarter fewer changes), or after the first three passes
            // still every second segment changing: the wrong class
            const bool stalled = prev_changed != 0 && uint64_t(last) * 4u > uint64_t(prev_changed) * 3u;
            const bool hopeless = prev_changed == 0 && uint64_t(last) * 2u > p.nseg;
            prev_changed = last;
            if (!converged && !to_the_end && (stalled || hopeless)) {
                if (last < best_changed) { best_changed = last; best_phase = phase; }
                break;
            }
        }
    }
    if (!converged && (L.nseg < 256 || p.order == 2)) {          // a workspace too small to hold the maps (a small stream); order 2
        note_index_path(ws, IDX_PATH_WALK, st);
        if (p.order == 2) hipLaunchKernelGGL(build_index_kernel<2>, dim3(1), dim3(64), 0, st, p);
        else hipLaunchKernelGGL(build_index_kernel<1>, dim3(1), dim3(64), (size_t(256) << p.P) * 2, st, p);
        return hipGetLastError();
    }
    if (!converged) {      // segments that never re-synchronise
        // fixed-length codes (see index_ctx_scan_kernel): positions are arithmetic, the context chain is composed
        // from per-group maps.  The per-segment arrays of the workspace are free again and hold the maps.
        uint32_t *info = reinterpret_cast<uint32_t *>(p.seg_sym_start);
        hipLaunchKernelGGL(index_ctx_scan_kernel, dim3(1), dim3(256), 0, st, p, info);
        uint32_t h[256];
        e = hipMemcpyAsync(h, info, sizeof h, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
        uint8_t live[256];
        uint32_t nlive = 0, gfix = 0;
        bool fixed = true, emits_prev0 = false;
        for (uint32_t c = 0; c < 256; ++c) {
            if (!(h[c] & 1u)) continue;
            live[nlive++] = uint8_t(c);
            const uint32_t l = (h[c] >> 8) & 255u;
            if (c == p.prev0) continue;                          // the first symbol's context: looked at below
            if (l == 0 || (gfix != 0 && l != gfix)) fixed = false;
            gfix = gfix ? gfix : l;
            emits_prev0 = emits_prev0 || ((h[c] >> 16) & 1u);
        }
        const uint32_t l0 = (h[p.prev0 & 255u] & 1u) ? (h[p.prev0 & 255u] >> 8) & 255u : 0u;
        if (gfix == 0) gfix = l0;                                // the start context is the only live one
        // the start context may have its own length if the stream never comes back to it
        if (l0 == 0 || (l0 != gfix && (emits_prev0 || ((h[p.prev0 & 255u] >> 16) & 1u)))) fixed = false;
        const uint64_t nsym = fixed && gfix && p.nbits >= l0 && (p.nbits - l0) % gfix == 0 ? (p.nbits - l0) / gfix + 1 : 0;
        // groups as small as the workspace allows (256 bytes of map each), but not below one chunk or 16 Ki symbols:
        // more groups = more threads for the two passes, and the chain over the groups is a single thread
        uint32_t shift = p.chunk_shift > 14u ? p.chunk_shift : 14u;
        uint64_t ngroups = (nsym + (1ull << shift) - 1) >> shift;
        while (shift < 30u && (ngroups * 256 > L.off_used - L.off_end || ngroups > L.off_count - L.off_used)) {
            ++shift;
            ngroups = (nsym + (1ull << shift) - 1) >> shift;
        }
        const bool room = ngroups * 256 <= L.off_used - L.off_end && ngroups <= L.off_count - L.off_used;
        if (p.order != 2 && fixed && nsym && room) {
            uint8_t *d_live = reinterpret_cast<uint8_t *>(p.seg_count);
            e = hipMemcpyAsync(d_live, live, nlive, hipMemcpyHostToDevice, st);
            if (e != hipSuccess) return e;
            e = hipStreamSynchronize(st);                        // `live` is on this stack frame
            if (e != hipSuccess) return e;
            IdxFixed f{d_live, nlive, l0, gfix, shift, nsym, ngroups, reinterpret_cast<uint8_t *>(p.seg_end_state), reinterpret_cast<uint8_t *>(p.seg_used)};
            const uint64_t nthreads = ngroups * nlive;
            note_index_path(ws, IDX_PATH_GROUP_MAPS, st);
            hipLaunchKernelGGL(index_group_map_kernel, dim3(unsigned((nthreads + 255) / 256)), dim3(256), 0, st, p, f);
            hipLaunchKernelGGL(index_group_chain_kernel, dim3(1), dim3(64), 0, st, p, f);
            hipLaunchKernelGGL(index_group_fill_kernel, dim3(unsigned((ngroups + 63) / 64)), dim3(64), 0, st, p, f);
            return hipGetLastError();
        }
        // mixed code lengths: maps over bit groups from (context, offset) states (index_state_map_kernel)
        if (p.order != 2 && nlive > 0 && p.max_len >= 1 && p.max_len <= 32) {
            const uint32_t maxlen = p.max_len;
            const size_t map_room = L.off_count - L.off_end;         // the per-segment end / used arrays
            const size_t aux_room = L.off_blk - L.off_count;         // counts and symbol starts: live lists, group starts, bases
            uint32_t shift = 16;
            uint64_t ngroups = (p.nbits + (1ull << shift) - 1) >> shift;
            auto fits = [&](uint64_t ng) { return ng * nlive * maxlen * 8 <= map_room && 1024 + ng * 2 + 64 + ng * 8 <= aux_room; };
            while (shift < 40u && !fits(ngroups)) { ++shift; ngroups = (p.nbits + (1ull << shift) - 1) >> shift; }
            if (fits(ngroups) && (1ull << shift) > maxlen) {
                uint8_t inv[256];
                for (uint32_t c = 0; c < 256; ++c) inv[c] = 255;
                for (uint32_t i = 0; i < nlive; ++i) inv[live[i]] = uint8_t(i);
                unsigned char *aux = ws + L.off_count;
                e = hipMemcpyAsync(aux, live, 256, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipMemcpyAsync(aux + 256, inv, 256, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);   // the two arrays are on this stack frame
                if (e != hipSuccess) return e;
                const size_t gstart_off = 1024, gbase_off = (gstart_off + ngroups * 2 + 63) & ~size_t(63);
                IdxState f{aux, aux + 256, nlive, maxlen, shift, ngroups, reinterpret_cast<unsigned long long *>(ws + L.off_end),
                           reinterpret_cast<uint16_t *>(aux + gstart_off), reinterpret_cast<unsigned long long *>(aux + gbase_off)};
                const uint64_t nthreads = ngroups * nlive * maxlen;
                note_index_path(ws, IDX_PATH_STATE_MAPS, st);
                hipLaunchKernelGGL(index_state_map_kernel, dim3(unsigned((nthreads + 255) / 256)), dim3(256), 0, st, p, f);
                hipLaunchKernelGGL(index_state_chain_kernel, dim3(1), dim3(64), 0, st, p, f);
                hipLaunchKernelGGL(index_state_fill_kernel, dim3(unsigned((ngroups + 63) / 64)), dim3(64), 0, st, p, f);
                return hipGetLastError();
            }
        }
        // the slow, certain way: one lane walks the payload
        note_index_path(ws, IDX_PATH_WALK, st);
        if (p.order == 2) hipLaunchKernelGGL(build_index_kernel<2>, dim3(1), dim3(64), 0, st, p);
        else hipLaunchKernelGGL(build_index_kernel<1>, dim3(1), dim3(64), (size_t(256) << p.P) * 2, st, p);
        return hipGetLastError();
    }
    note_index_path(ws, IDX_PATH_SEGMENTS, st);
    unsigned long long *blk_sum = reinterpret_cast<unsigned long long *>(ws + L.off_blk);
    (void)launch_scan_local(p.seg_count, p.nseg, p.seg_sym_start, blk_sum, st);
    (void)launch_scan_top(blk_sum, nblk, nullptr, st);
    hipLaunchKernelGGL(index_scan_add_kernel, dim3(unsigned(nblk)), dim3(SCAN_THREADS), 0, st, p.seg_sym_start, blk_sum, p.nseg, nblk, p.n_symbols);
    hipLaunchKernelGGL(index_fill_kernel, dim3(grid), dim3(256), 0, st, p);
    return hipGetLastError();
}


// ---- streams without an index in TWO passes over the payload [r5]: states (above), then the segment decoder (mh_tile.hip)
// launch_stream_states: status block bytes 8..11 = IDX_PATH_STATES when the workspace is ready for launch_stream_emit,
// IDX_PATH_NONE when this stream / model does not take the fast path (the caller then builds an index as before).
static void stream_params(IdxParams &p, unsigned char *ws, const IdxWs &L) {
    p.status = reinterpret_cast<int *>(ws);
    p.changed = reinterpret_cast<unsigned int *>(ws + L.off_changed);
}
hipError_t launch_stream_states(IdxParams p, void *d_ws, hipStream_t st) {
    unsigned char *ws = static_cast<unsigned char *>(d_ws);
    const IdxWs L = idx_ws_layout(p.nbits);
    stream_params(p, ws, L);
    hipError_t e = hipMemsetAsync(ws, 0, L.off_end, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(p.n_symbols, 0, 8, st);
    if (e != hipSuccess || p.nbits == 0) return e;
    const uint32_t g = p.len_gcd >= 1 && p.len_gcd <= 64 ? p.len_gcd : 1;
    if (!index_tiles_eligible(p, L, g, true)) return hipSuccess;
    IdxParams q;
    bool ok = false;
    e = index_tile_states(p, L, ws, st, q, &ok);
    if (e != hipSuccess) return e;
    if (ok) note_index_path(ws, IDX_PATH_STATES, st);
    else e = hipMemsetAsync(ws, 0, L.off_end, st);
    return e;
}
// [r5] the segments the segment decoder left alone (bit 15 of their count: a code longer than the tile tables resolve), listed by
// it in dirty_list (counter: the last word of `changed`): one thread each, general tables with the tree walk
// (src/coding.cpp:129-149), bytes written one by one.  Rare by construction: such a code has a probability under 2^-15 in its
// context.
__global__ __launch_bounds__(256) void segment_walk_emit_kernel(IdxParams p, uint8_t *out, uint64_t out_cap) {
    const uint32_t count = p.changed[IDX_MAX_PASSES - 1] < p.dirty_cap ? p.changed[IDX_MAX_PASSES - 1] : p.dirty_cap;
    if (p.changed[IDX_MAX_PASSES - 1] > p.dirty_cap && blockIdx.x == 0 && threadIdx.x == 0) atomicExch(p.status, MHK_STATUS_CAPACITY);
    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    const DecTables tabs{p.sec, p.tree, p.P, p.direct, p.H};
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < count; j += gridDim.x * blockDim.x) {
        const uint64_t i = p.dirty_list[j];
        const uint32_t pe = i ? uint32_t(p.e16[i - 1]) : (p.prev0 << 8);
        const uint64_t seg_end = (i + 1) * IX_SEG_BITS < p.nbits ? (i + 1) * IX_SEG_BITS : p.nbits;
        uint64_t pos = i * IX_SEG_BITS + (pe & 255u);
        uint32_t prev = pe >> 8;
        uint64_t base = p.tile_base[i / IX_TILE_SEGS];
        for (uint64_t s = (i / IX_TILE_SEGS) * IX_TILE_SEGS; s < i; ++s) base += p.c16[s] & IX_C16_COUNT;
        const uint32_t want = p.c16[i] & IX_C16_COUNT;
        GranuleCursor bc;
        bc.init(src, pos);
        uint32_t k = 0;
        bool bad = false;
        while (pos < seg_end && k < want) {
            uint32_t used = 0;
            const uint32_t sym = decode_one(p.prim, p.sec_base, tabs, src, bc, prev, used, bad);
            if (bad) break;
            if (base + k < out_cap) out[base + k] = uint8_t(sym); else atomicExch(p.status, MHK_STATUS_CAPACITY);
            prev = sym;
            pos += used;
            ++k;
        }
        const uint64_t over = pos - seg_end;
        if (bad || pos < seg_end || k != want || uint32_t(p.e16[i]) != ((prev << 8) | uint32_t(over > 254 ? 254 : over))) atomicExch(p.status, MHK_STATUS_CORRUPT);
    }
}

hipError_t launch_stream_emit(IdxParams p, void *d_ws, uint8_t *d_out, uint64_t out_cap, hipStream_t st) {
    unsigned char *ws = static_cast<unsigned char *>(d_ws);
    const IdxWs L = idx_ws_layout(p.nbits);
    stream_params(p, ws, L);
    p.e16 = reinterpret_cast<uint16_t *>(ws + L.off_e16);
    p.s16 = reinterpret_cast<uint16_t *>(ws + L.off_s16);
    p.c16 = reinterpret_cast<uint16_t *>(ws + L.off_c16);
    p.tile_cnt = reinterpret_cast<uint32_t *>(ws + L.off_tcnt);
    p.tile_base = reinterpret_cast<unsigned long long *>(ws + L.off_tbase);
    p.nseg5 = L.nseg5; p.ntile5 = L.ntile5;
    p.dirty_list = reinterpret_cast<uint32_t *>(ws + L.off_dirty);
    p.dirty_cap = uint32_t(L.dirty_cap < 0xFFFFFFFFull ? L.dirty_cap : 0xFFFFFFFFull);
    const bool walks = p.max_len > p.tP + p.tH;                   // only such a model has segments for the walk
    hipError_t e = walks ? hipMemsetAsync(p.changed + (IDX_MAX_PASSES - 1), 0, 4, st) : hipSuccess;
    if (e != hipSuccess) return e;
    e = launch_segment_decode(p, d_out, out_cap, st);
    if (e != hipSuccess || !walks) return e;
    hipLaunchKernelGGL(segment_walk_emit_kernel, dim3(64), dim3(256), 0, st, p, d_out, out_cap);
    return hipGetLastError();
}

}  // namespace mhk
