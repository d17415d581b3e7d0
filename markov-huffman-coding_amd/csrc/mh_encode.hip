// mh_encode.hip — the encoders of the Markov-Huffman hot path for gfx950 (order 2: N4).
//   enc_len_kernel / scan_* / enc_emit_kernel   length pass, prefix, emit (inputs under 4 MiB, order-2 fallback)
//   enc_chain_kernel                            one-pass order-2 encoder (chained scan)
//   region_bits / region_scan / enc_region_kernel  the region encoder priced from a region-mode histogram: one read of the input
//   enc2_len_kernel / enc2_emit_kernel          order 2 with every codeword gathered from the full tables
#include "mh_dev.hpp"

namespace mhk {

// ------------------------------------------------------------------------------------------------
// encode: three dependency-free steps
// ------------------------------------------------------------------------------------------------
//   enc_len_kernel   per wave-tile (4 KiB of input) sum of code lengths          reads n
//   scan_*           exclusive prefix over the wave-tile sums -> absolute bit offsets; zeroes the
//                    one output dword at every wave-tile seam
//   enc_emit_kernel  every WAVE encodes its wave-tiles on its own: LDS codeword table, wave
//                    prefix-sum of bit lengths, bits OR-ed into a wave-private LDS image that is
//                    already aligned to the absolute output dwords, coalesced dword stores; the
//                    two seam dwords of a wave-tile are merged with global atomic OR.  No barrier,
//                    no inter-workgroup hand-off, nothing to wait for.
// A single-pass variant with decoupled look-back across tiles was measured first (round 1): its
// prefix chain advances <= 64 tiles per ~2 us descriptor hop across XCDs, i.e. ~260 GB/s; the extra
// read of the length pass costs far less than that chain.

// ---- order 2 with the live contexts' tables in LDS (BASELINE config 5: "LDS codeword-table staging") ----
// Text-like sources use a few hundred two-byte contexts over a few dozen byte values.  The model builder (mh_api.cpp,
// dev_model_build2) ranks the byte values (the 63 most frequent get ids 0..62, every other byte id 63) and gives the
// heaviest contexts whose two bytes both have an id < 63 a slot; the image `o2hot` it hands over is
//     symid[256] u8 | ctxmap[64 * 64] u16 (id of the byte before the previous << 6 | (id of the previous ^ that id) -> slot) |
//     hot[(nslots + 1) * 64] u16 (slot << 6 | (id of the symbol ^ id of the previous byte) -> len << 12 | code, as the order-1 table)
// with the last slot all ENC16_ESCAPE (what ctxmap gives for every other context) and column 63 all ENC16_ESCAPE.  An
// escape sends the wave's sub-step through the symbol-by-symbol path with the full tables in L2 (emit_substep_slow<2>), so
// the image is only handed over when the slots cover (nearly) the whole input (the builder knows every context's weight).
constexpr uint32_t O2H_MAP_OFF = 256, O2H_HOT_OFF = 256 + 64 * 64 * 2;
__device__ __forceinline__ void o2hot_lookup16(const unsigned char *img, const uint4 &x4, uint32_t ctx, uint32_t (&e)[16]) {
    const uint16_t *ctxmap = reinterpret_cast<const uint16_t *>(img + O2H_MAP_OFF);
    const uint16_t *hot = reinterpret_cast<const uint16_t *>(img + O2H_HOT_OFF);
    const uint32_t x[4] = {x4.x, x4.y, x4.z, x4.w};
    uint32_t id[18];
    id[0] = img[ctx >> 8];
    id[1] = img[ctx & 255u];
#pragma unroll
    for (int j = 0; j < 16; ++j) id[2 + j] = img[(x[j >> 2] >> (8 * (j & 3))) & 255u];
    // Both tables are read at a column XOR-ed with the id of the byte in front: with a few dozen byte values, and
    // rows of 64 two-byte entries = 32 banks, the bank of a plain [row][id] access is id / 2 whatever the row — every
    // lane that looks at a frequent letter lands on the same bank (the builder stores the rows permuted accordingly)
    uint32_t cs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) cs[j] = ctxmap[(id[j] << 6) | (id[j + 1] ^ id[j])];
#pragma unroll
    for (int j = 0; j < 16; ++j) e[j] = hot[(cs[j] << 6) | (id[j + 2] ^ id[j + 1])];
}

// eight symbols (two dwords) — enc_chain_kernel holds a whole wave-tile's results in registers and has no room for sixteen in flight
__device__ __forceinline__ void o2hot_lookup8(const unsigned char *img, uint32_t xa, uint32_t xb, uint32_t ctx, uint32_t (&e)[8]) {
    const uint16_t *ctxmap = reinterpret_cast<const uint16_t *>(img + O2H_MAP_OFF);
    const uint16_t *hot = reinterpret_cast<const uint16_t *>(img + O2H_HOT_OFF);
    const uint32_t x[2] = {xa, xb};
    uint32_t id[10];
    id[0] = img[ctx >> 8];
    id[1] = img[ctx & 255u];
#pragma unroll
    for (int j = 0; j < 8; ++j) id[2 + j] = img[(x[j >> 2] >> (8 * (j & 3))) & 255u];
    uint32_t cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = ctxmap[(id[j] << 6) | (id[j + 1] ^ id[j])];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = hot[(cs[j] << 6) | (id[j + 2] ^ id[j + 1])];
}

// ---- pass 1 ------------------------------------------------------------------------------------
// ORDER 2: the hot order-2 image in LDS (o2hot_lookup16); lengths of escapes come from p.len_slot = len8[ctx * 256 + sym]
template <int ORDER>
__global__ __launch_bounds__(E_THREADS, ORDER == 1 ? 8 : 4) void enc_len_kernel(LenParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t *ltab = smem;   // code length per slot (0..64)
    if (ORDER == 1) {
        for (int i = threadIdx.x; i < 4096; i += E_THREADS)
            reinterpret_cast<uint4 *>(ltab)[i] = reinterpret_cast<const uint4 *>(p.len_slot)[i];
    } else {
        for (uint32_t i = threadIdx.x; i < (p.o2hot_bytes + 15u) / 16u; i += E_THREADS)
            reinterpret_cast<uint4 *>(smem)[i] = reinterpret_cast<const uint4 *>(p.o2hot)[i];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave0 = uint64_t(blockIdx.x) * E_WAVES + (threadIdx.x >> 6);
    const uint64_t nwaves = uint64_t(gridDim.x) * E_WAVES;
    LaneIn ahead = ORDER == 1 ? load_raw(p.data, p.n, wave0 * E_WT + lane * E_VEC, p.prev0) : load_raw2(p.data, p.n, wave0 * E_WT + lane * E_VEC, p.prev0);
    for (uint64_t wt = wave0; wt < p.nwt; wt += nwaves) {
        uint32_t sum = 0;
#pragma unroll 1
        for (int k = 0; k < E_SUBSTEPS; ++k) {
            const LaneIn in = ahead;
            {
                const uint64_t nwt_ = (k + 1 < E_SUBSTEPS) ? wt : wt + nwaves;
                const uint64_t noff = nwt_ * E_WT + uint64_t((k + 1) % E_SUBSTEPS) * E_SUB + lane * E_VEC;
                ahead = ORDER == 1 ? load_raw(p.data, p.n, noff, p.prev0) : load_raw2(p.data, p.n, noff, p.prev0);
            }
            if (ORDER == 2) {
                uint32_t e[16];
                uint32_t ctx = head_ctx(in);
                o2hot_lookup16(smem, in.x, ctx, e);
                const uint32_t x[4] = {in.x.x, in.x.y, in.x.z, in.x.w};
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t sym = (x[j >> 2] >> (8 * (j & 3))) & 255u;
                    uint32_t l = e[j] >> 12;
                    if (e[j] >= 0xD000u) {                     // escape: the full table in L2 (rare by construction)
                        l = uint32_t(j) < in.nvalid ? uint32_t(p.len_slot[(ctx << 8) | sym]) : 0u;
                        if (l > 64u) l = 0;
                    }
                    sum += uint32_t(j) < in.nvalid ? l : 0u;
                    ctx = ((ctx << 8) | sym) & 0xFFFFu;
                }
                continue;
            }
            uint32_t w[16];
            slots16(in.x, head_byte(in), w);
            if (__all(in.nvalid == E_VEC)) {             // wave-uniform: everything but the stream's last vectors
#pragma unroll
                for (int j = 0; j < 16; ++j) sum += ltab[w[j]];
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    uint32_t l = ltab[w[j]];
                    sum += (uint32_t(j) < in.nvalid) ? l : 0u;
                }
            }
        }
        sum = wave_sum(sum);
        if (lane == 0) p.wt_bits[wt] = sum;
    }
}


// per block of 4096 wave-tiles: exclusive prefix within the block + the block's total
__global__ __launch_bounds__(SCAN_THREADS) void scan_local_kernel(const uint32_t *wt_bits, uint64_t nwt,
                                                                  unsigned long long *wt_start, unsigned long long *blk_sum) {
    __shared__ uint64_t lds[SCAN_THREADS / 64];
    uint64_t i0 = uint64_t(blockIdx.x) * SCAN_BLOCK + uint64_t(threadIdx.x) * SCAN_PER_THREAD;
    uint64_t v[SCAN_PER_THREAD], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) { v[k] = (i0 + k < nwt) ? wt_bits[i0 + k] : 0; s += v[k]; }
    uint64_t total;
    uint64_t ex = block_excl_scan(s, lds, total);
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) { if (i0 + k < nwt) wt_start[i0 + k] = ex; ex += v[k]; }
    if (threadIdx.x == 0) blk_sum[blockIdx.x] = total;
}

// one block: exclusive scan of the block totals in place; writes the grand total after the last entry
// carry0: device pointer to the global bit position this payload starts at (only its low 3 bits are
// used: the payload is emitted pre-shifted so that shards concatenate with one OR-merged seam byte), or
// nullptr.
__global__ __launch_bounds__(SCAN_THREADS) void scan_top_kernel(unsigned long long *blk_sum, uint64_t nblk,
                                                                const unsigned long long *carry0) {
    __shared__ uint64_t lds[SCAN_THREADS / 64];
    uint64_t carry = carry0 ? (*carry0 & 7ull) : 0;
    for (uint64_t base = 0; base < nblk; base += SCAN_THREADS) {
        uint64_t i = base + threadIdx.x;
        uint64_t v = i < nblk ? blk_sum[i] : 0;
        uint64_t total;
        uint64_t ex = block_excl_scan(v, lds, total);
        if (i < nblk) blk_sum[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) blk_sum[nblk] = carry;
}

// adds the block offsets, publishes the total, zeroes the output dword under every wave-tile seam
// (those dwords are completed by global atomic OR from two neighbouring waves)
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(ScanParams p) {
    const unsigned long long boff = p.blk_sum[blockIdx.x];
    const uint64_t total = p.blk_sum[p.nblk];
    const uint64_t cap_bits = p.cap * 8;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *p.nbits = total;
        if (total > cap_bits) atomicExch(p.status, MHK_STATUS_CAPACITY);
        uint64_t endw = total >> 5;
        if ((total & 31u) && (endw + 1) * 4 <= p.cap) reinterpret_cast<uint32_t *>(p.out)[endw] = 0;
        else if (total & 31u) for (uint64_t b = endw * 4; b < p.cap; ++b) p.out[b] = 0;
    }
    uint64_t i0 = uint64_t(blockIdx.x) * SCAN_BLOCK + uint64_t(threadIdx.x) * SCAN_PER_THREAD;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        uint64_t i = i0 + k;
        if (i < p.nwt) {
            uint64_t s = p.wt_start[i] + boff;
            p.wt_start[i] = s;
            if ((s & 31u) && ((s >> 5) + 1) * 4 <= p.cap) reinterpret_cast<uint32_t *>(p.out)[s >> 5] = 0;
        }
    }
}

// ---- pass 2 ------------------------------------------------------------------------------------
// OR a left-aligned string (first bit at bit 63 of `vl`) into the wave's staging image at image bit
// offset `o` (image word j <-> output dword base + j).  CLIP: only words in [wbase, wbase + nwords).
template <bool CLIP>
__device__ __forceinline__ void deposit(uint32_t *stage, uint64_t vl, uint32_t o, uint32_t wbase, uint32_t nwords) {
    uint32_t hi = uint32_t(vl >> 32), lo = uint32_t(vl);
    uint32_t sh = o & 31u;
    uint32_t w0 = hi >> sh;
    uint32_t w1 = __builtin_amdgcn_alignbit(hi, lo, sh);
    uint32_t w2 = __builtin_amdgcn_alignbit(lo, 0u, sh);
    uint32_t wi = (o >> 5) - wbase;   // wraps when below the window; the unsigned compares reject it
    if (CLIP) {
        if (w0 && wi < nwords) atomicOr(&stage[wi], w0);
        if (w1 && wi + 1u < nwords) atomicOr(&stage[wi + 1u], w1);
        if (w2 && wi + 2u < nwords) atomicOr(&stage[wi + 2u], w2);
    } else {
        atomicOr(&stage[wi], w0);                 // OR-ing a zero is harmless and cheaper than testing for it
        atomicOr(&stage[wi + 1u], w1);
        if (w2) atomicOr(&stage[wi + 2u], w2);    // only groups that straddle two word boundaries
    }
}

// Stores image words [0, nfull) to output dwords gbase + j (MSB-first bytes), clears them, and moves
// image word `nfull` (the partial tail) to word 0.  Word 0 is the seam with the previous wave-tile when `seam0` is
// set: SEAM_OR = it goes out with an atomic OR (both neighbours write their part of a zeroed dword), SEAM_DROP = it is
// not written at all (enc_chain_kernel: the previous wave-tile writes that dword whole).
// Wave-synchronous: LDS ops of one wave execute in order.
constexpr uint32_t SEAM_NONE = 0, SEAM_OR = 1, SEAM_DROP = 2;
__device__ __forceinline__ void flush_words(uint32_t *stage, uint32_t *out32, uint64_t gbase, uint32_t nfull,
                                            uint32_t seam0, uint32_t lane) {
    const uint32_t tail = stage[nfull];
    for (uint32_t j = lane; j < nfull; j += 64u) {
        uint32_t v = __builtin_bswap32(stage[j]);
        stage[j] = 0;
        if (j == 0 && seam0 != SEAM_NONE) { if (seam0 == SEAM_OR) atomicOr(&out32[gbase], v); }
        else out32[gbase + j] = v;
    }
    if (lane == 0) { stage[nfull] = 0; stage[0] = tail; }
}

// Order-2 fine index entry of the lane's sub-chunk (every fourth lane): two context bytes << 16 | bits from the chunk's
// index entry to the sub-chunk (exc = bits of the wave's 1 KiB sub-step in front of the lane; a chunk of S <= 1024
// symbols starts inside the sub-step, at the lane whose offset is a multiple of S).  0xFFFF: does not fit 16 bits.
__device__ __forceinline__ void fine2_entry(const EmitParams &p, uint32_t S, uint32_t lane, uint64_t off, uint32_t nvalid, uint32_t ctx, uint32_t exc) {
    if (!p.fine || S > uint32_t(E_SUB)) return;                  // (wave-uniform)
    const uint32_t first = lane & ~((S >> 4) - 1u);              // the lane that starts this lane's chunk
    const uint32_t d = exc - uint32_t(__shfl(int(exc), int(first)));
    if (nvalid && (lane & (uint32_t(1u << T_SUB_SHIFT) / 16u - 1u)) == 0u) p.fine[off >> T_SUB_SHIFT] = (ctx << 16) | (d > 0xFFFFu ? 0xFFFFu : d);
}

// Escape path of one sub-step (some code in the wave is longer than 12 bits): everything is recomputed
// from the lane's 16 input bytes so that the hot path keeps no per-symbol state alive.  The sub-step
// may carry up to 64 bits per symbol, so the image is filled and flushed in rounds.
// ORDER 2 (extension, see the order-2 section below): pb holds the TWO bytes before the lane's vector,
// (byte before previous) << 8 | previous byte, every codeword comes from the full tables in HBM/L2
// (len8 / code64 indexed ctx * 256 + sym), and index entries carry the 16-bit context in bits 48..63.
template <int ORDER>
__device__ __forceinline__ void emit_substep_slow(const EmitParams &p, const uint16_t *tab, uint32_t *stage, uint32_t *out32,
                                                  uint4 x, uint32_t pb, uint32_t nvalid, uint32_t lane, uint64_t off,
                                                  uint64_t abs_bits, uint64_t &gbase, uint32_t &cur, uint32_t &seam0,
                                                  uint32_t &sub_bits_out) {
    // opaque to the optimiser, so that nothing of the hot path is kept alive for this rare branch
    asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w), "+v"(pb));
    // rolling walk over the lane's bytes: no per-symbol arrays, a handful of registers
    struct Roll {
        uint4 x; uint32_t prev;
        // order 1: sym << 8 | prev (the raw 16-bit field of the stream); order 2: ctx16 << 8 | sym
        __device__ __forceinline__ uint32_t next_window() {
            uint32_t sym = x.x & 255u;
            uint32_t win = ORDER == 2 ? ((prev << 8) | sym) : ((sym << 8) | prev);
            prev = ORDER == 2 ? (((prev << 8) | sym) & 0xFFFFu) : sym;
            x.x = __builtin_amdgcn_alignbyte(x.y, x.x, 1);
            x.y = __builtin_amdgcn_alignbyte(x.z, x.y, 1);
            x.z = __builtin_amdgcn_alignbyte(x.w, x.z, 1);
            x.w >>= 8;
            return win;
        }
    };
    auto code_of = [&](uint32_t win, bool valid, uint32_t &l, uint64_t &c) {
        if (ORDER == 2) {
            l = valid ? uint32_t(p.len8[win]) : 0u;
            c = valid ? p.code64[win] : 0ull;
            if (l > 64u) { l = 0; c = 0; }
            return;
        }
        uint32_t e = valid ? uint32_t(tab[mh::enc_slot(win)]) : 0u;
        l = e >> 12;
        c = e & 0xFFFu;
        if (e >= 0xD000u) {
            uint32_t nat = ((win & 255u) << 8) | (win >> 8);     // prev * 256 + sym
            l = p.len8[nat];
            c = p.code64[nat];
            if (l > 64u) { l = 0; c = 0; }                       // rejected on the host
        }
    };
    uint32_t L = 0;
    {
        Roll r{x, pb};
#pragma unroll 1
        for (uint32_t j = 0; j < 16; ++j) { uint32_t l; uint64_t c; code_of(r.next_window(), j < nvalid, l, c); L += l; }
    }
    uint32_t inc = L;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d);
        if (lane >= uint32_t(d)) inc += t;
    }
    const uint32_t sub_bits = __shfl(inc, 63);
    const uint32_t exc = inc - L;
    const uint32_t S = 1u << p.chunk_shift;
    if (p.index && nvalid && ((uint32_t(off) & (S - 1u)) == 0u))
        p.index[off >> p.chunk_shift] = (uint64_t(pb) << (ORDER == 2 ? 48 : 56)) | (abs_bits + exc);
    if (ORDER != 2 && p.fine && nvalid && (lane & (uint32_t(1u << T_SUB_SHIFT) / 16u - 1u)) == 0u)         // every fourth lane starts a 64-symbol sub-chunk
        p.fine[off >> T_SUB_SHIFT] = (pb << 24) | (uint32_t(abs_bits + exc) & FINE_POS_MASK);
    if (ORDER == 2) fine2_entry(p, S, lane, off, nvalid, pb, exc);

    const uint32_t end = cur + sub_bits;     // image bit one past the sub-step (frame of this sub-step)
    const uint32_t nwords = uint32_t(E_STAGE_WORDS - 2);
    uint32_t wbase = 0;                       // frame word that stage[0] currently holds
    for (;;) {
        uint32_t o = cur + exc;
        Roll r{x, pb};
#pragma unroll 1
        for (uint32_t j = 0; j < 16; ++j) {
            uint32_t l; uint64_t c;
            code_of(r.next_window(), j < nvalid, l, c);
            if (l) deposit<true>(stage, c << (64u - l), o, wbase, nwords);
            o += l;
        }
        uint32_t nfull = (end >> 5) - wbase;
        const bool more = nfull > nwords - 1u;
        if (more) nfull = nwords - 1u;        // keep one word as the moving tail
        flush_words(stage, out32, gbase, nfull, seam0, lane);
        if (nfull) seam0 = SEAM_NONE;
        gbase += nfull;
        wbase += nfull;
        if (!more) break;
    }
    cur = end & 31u;
    sub_bits_out = sub_bits;
}

// ORDER 2: the emit loop over the hot order-2 image (o2hot_lookup16) instead of the order-1 codeword table; the context of a
// lane is the two bytes before its vector, index entries carry it in bits 48..63, and the fine index entry is
// context << 16 | bit offset relative to the chunk's index entry (0xFFFF: does not fit; chunks of at most 1024 symbols)
template <int ORDER>
__global__ __launch_bounds__(E_THREADS) void enc_emit_kernel(EmitParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tab_bytes = ORDER == 1 ? 131072u : ((p.o2hot_bytes + 15u) & ~15u);
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem + tab_bytes) + wave * E_STAGE_WORDS;
    if (ORDER == 1) {
        for (int i = threadIdx.x; i < 8192; i += E_THREADS)
            reinterpret_cast<uint4 *>(tab)[i] = reinterpret_cast<const uint4 *>(p.enc16)[i];
    } else {
        for (uint32_t i = threadIdx.x; i < tab_bytes / 16u; i += E_THREADS)
            reinterpret_cast<uint4 *>(smem)[i] = reinterpret_cast<const uint4 *>(p.o2hot)[i];
    }
    for (int i = lane; i < E_STAGE_WORDS; i += 64) stage[i] = 0;
    __syncthreads();
    if (*p.status != MHK_STATUS_OK) return;     // capacity overrun found by the scan: write nothing

    uint32_t *out32 = reinterpret_cast<uint32_t *>(p.out);
    const uint32_t S = 1u << p.chunk_shift;
    const uint64_t wave0 = uint64_t(blockIdx.x) * E_WAVES + wave;
    const uint64_t nwaves = uint64_t(gridDim.x) * E_WAVES;
    // Software pipeline over the wave's sub-steps i = 0, 1, ... (sub-step i = piece i % 4 of wave-tile
    // wave0 + (i / 4) * nwaves): while sub-step i is packed, scanned, deposited and flushed, the 16 codeword
    // lookups of sub-step i + 1 are already in the LDS queue and the inputs of sub-steps i + 2 and i + 3 are
    // on their way from HBM.  (With the lookups issued at the top of their own sub-step the wave sat through the LDS
    // round trip three times per sub-step: lookups, tail word, flush reads.)
    auto offset_of = [&](uint64_t i) -> uint64_t {
        return (wave0 + (i >> 2) * nwaves) * E_WT + (i & 3u) * E_SUB + lane * E_VEC;
    };
    auto lookup16 = [&](const LaneIn &in, uint32_t pb, uint32_t (&e)[16]) {
        if (ORDER == 2) { o2hot_lookup16(smem, in.x, pb, e); return; }
        uint32_t w[16];
        slots16(in.x, pb, w);
#pragma unroll
        for (int j = 0; j < 16; ++j) e[j] = uint32_t(tab[w[j]]);
    };
    auto load = [&](uint64_t off) -> LaneIn { return ORDER == 1 ? load_raw(p.data, p.n, off, p.prev0) : load_raw2(p.data, p.n, off, p.prev0); };
    auto head = [&](const LaneIn &in) -> uint32_t { return ORDER == 1 ? head_byte(in) : head_ctx(in); };
    LaneIn cur_in = load(offset_of(0));
    LaneIn next_in = load(offset_of(1));
    LaneIn next2_in = load(offset_of(2));
    uint32_t cur_pb = head(cur_in);
    uint32_t E[16];
    lookup16(cur_in, cur_pb, E);
    uint64_t gbase = 0, abs_bits = 0;
    uint32_t cur = 0;
    uint32_t seam0 = SEAM_NONE;
#pragma unroll 1
    for (uint64_t i = 0;; ++i) {
        const uint64_t wt = wave0 + (i >> 2) * nwaves;
        if (wt >= p.nwt) break;
        const uint32_t k = uint32_t(i) & 3u;
        if (k == 0) {
            const uint64_t s = p.wt_start[wt];
            gbase = s >> 5;                      // output dword under image word 0
            cur = uint32_t(s & 31u);             // image bit where the next code goes
            abs_bits = s;                        // absolute bit offset of image bit `cur`
            seam0 = cur != 0 ? SEAM_OR : SEAM_NONE;   // word 0 is shared with the previous wave-tile
        }
        const uint64_t off = wt * E_WT + uint64_t(k) * E_SUB + lane * E_VEC;
        // the next sub-steps: input three ahead (its first use, the lookups, comes two sub-steps from now), lookups one ahead
        const LaneIn in3 = load(offset_of(i + 3));      // past the end: zeros, nothing is read
        const uint32_t next_pb = head(next_in);
        uint32_t En[16];
        lookup16(next_in, next_pb, En);
        // ---- this sub-step
        const uint4 x = cur_in.x;
        const uint32_t nvalid = cur_in.nvalid;
        const uint32_t pb = cur_pb;
        uint32_t L = 0;
        uint64_t g[4]; uint32_t gl[4];
        uint32_t emax = nvalid == E_VEC ? 0u : 0xFFFFu;   // ragged vectors take the symbol-by-symbol path
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t e0 = E[4 * q], e1 = E[4 * q + 1], e2 = E[4 * q + 2], e3 = E[4 * q + 3];
            uint32_t m01 = e0 > e1 ? e0 : e1, m23 = e2 > e3 ? e2 : e3;
            m01 = m01 > m23 ? m01 : m23;
            emax = m01 > emax ? m01 : emax;
            const uint32_t l0 = e0 >> 12, l1 = e1 >> 12, l2 = e2 >> 12, l3 = e3 >> 12;
            const uint32_t p01 = ((e0 & 0xFFFu) << l1) | (e1 & 0xFFFu);
            const uint32_t p23 = ((e2 & 0xFFFu) << l3) | (e3 & 0xFFFu);
            g[q] = (uint64_t(p01) << (l2 + l3)) | p23;
            gl[q] = l0 + l1 + l2 + l3;
            L += gl[q];
        }
        uint32_t sub_bits;
        if (__any(emax >= 0xD000u)) {        // wave-uniform: an escape code or a ragged vector somewhere
            emit_substep_slow<ORDER>(p, tab, stage, out32, x, pb, nvalid, lane, off, abs_bits, gbase, cur, seam0, sub_bits);
        } else {
            // exclusive wave scan of the lane totals
            const uint32_t inc = wave_inclusive_sum(L);
            sub_bits = __builtin_amdgcn_readlane(inc, 63);
            const uint32_t exc = inc - L;
            // chunk index: the lane whose first byte starts a chunk records (context, bit offset)
            if (p.index && nvalid && ((uint32_t(off) & (S - 1u)) == 0u))
                p.index[off >> p.chunk_shift] = (uint64_t(pb) << (ORDER == 2 ? 48 : 56)) | (abs_bits + exc);
            if (ORDER == 1) {
                if (p.fine && nvalid && (lane & (uint32_t(1u << T_SUB_SHIFT) / 16u - 1u)) == 0u)         // fine index (mh_kernels.h, TileParams): every fourth lane
                    p.fine[off >> T_SUB_SHIFT] = (pb << 24) | (uint32_t(abs_bits + exc) & FINE_POS_MASK);
            } else {
                fine2_entry(p, S, lane, off, nvalid, pb, exc);
            }
            uint32_t o = cur + exc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (gl[q]) deposit<false>(stage, g[q] << (64u - gl[q]), o, 0, 0);
                o += gl[q];
            }
            const uint32_t nfull = (cur + sub_bits) >> 5;
            flush_words(stage, out32, gbase, nfull, seam0, lane);
            if (nfull) seam0 = SEAM_NONE;
            gbase += nfull;
            cur = (cur + sub_bits) & 31u;
        }
        abs_bits += sub_bits;
        // last partial dword of the wave-tile: seam with the next wave-tile (or the stream's end)
        if (k == 3u && cur != 0 && lane == 0) {
            atomicOr(&out32[gbase], __builtin_bswap32(stage[0]));
            stage[0] = 0;
        }
        cur_in = next_in; cur_pb = next_pb;
        next_in = next2_in;
        next2_in = in3;
#pragma unroll
        for (int j = 0; j < 16; ++j) E[j] = En[j];
    }
}

// ------------------------------------------------------------------------------------------------
// ORDER 2 in ONE pass: a chained scan over the wave-tiles
// ------------------------------------------------------------------------------------------------
// The length pass + emit pair looks every symbol up twice (three dependent LDS gathers per symbol each time: byte
// ids, context slot, codeword), and the lookups are what both kernels spend their time on.  Here a wave looks its
// 4 KiB wave-tile up ONCE, keeps the packed codeword groups in registers (rotating through four held sets so that the
// loops over the sub-steps stay rolled), and the start bits come from a chained scan in the manner of Merrill &
// Garland's decoupled look-back: a state word per GROUP of 16 wave-tiles (one round of one workgroup) is empty, then
// AGGREGATE | the group's bits, then PREFIX | bits up to and including it; a look-back adds aggregates down to the
// nearest prefix.  Groups are handed out in order through a ticket counter, one round of one workgroup at a time, so
// every group a look-back can wait for has been taken by a workgroup that is running; the wait is bounded anyway
// (CH_SPIN_MAX polls, then MHK_STATUS_TIMEOUT, a bogus prefix so that nobody else hangs, and the wave leaves).
// (Measured on the way, 4 GiB of text: four rounds per ticket ran the launch in sequence, 938 ms — the first tiles of a
// ticket wait for the aggregates of the previous ticket's LAST round; a state word per wave-tile instead of per group
// 6.2-6.8 ms; this 5.6 ms.)
// No dword of the output has two writers, so nothing needs zeroing and no global atomic is spent on seams: the dword
// that holds a tile's last bits is written by THAT tile, which encodes the next few symbols of the input itself to
// fill it (at most 31 bits, codes have at least one bit), and a tile never writes the part of its first dword that
// lies before its first dword boundary (SEAM_DROP) — tile 0 excepted, which starts the stream.
struct ChainParams {
    EmitParams e;                            // (wt_start unused)
    unsigned long long *state;               // one word per group of E_WAVES wave-tiles, zeroed
    const unsigned long long *start_bit;     // nullptr or the global bit position the payload starts at (low 3 bits used)
    uint64_t cap;                            // bytes, a multiple of 4
    unsigned long long *nbits;
    int *status;
    uint32_t probe;
    uint32_t *sync;                          // the ticket counter (zeroed)
};
constexpr unsigned long long CH_AGG = 1ull << 62, CH_PFX = 2ull << 62, CH_VAL = (1ull << 62) - 1ull;
constexpr uint32_t CH_SPIN_MAX = 1u << 20;                   // polls of one wait (~ a second) before giving up

// bits of all tiles before `tile` (tile >= 1); false: gave up waiting
__device__ __forceinline__ bool chain_lookback(const unsigned long long *state, uint64_t tile, uint32_t lane, uint64_t &excl) {
    uint64_t sum = 0;
    long long top = (long long)tile - 1;                      // nearest tile not yet accounted for
    uint32_t spins = 0;
    while (top >= 0) {
        const long long my = top - (long long)lane;
        unsigned long long st = CH_PFX;                       // in front of tile 0: an inclusive prefix of zero
        if (my >= 0) st = __hip_atomic_load(&state[my], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t flag = uint32_t(st >> 62);
        const unsigned long long pfx = __ballot(flag == 2u), emp = __ballot(flag == 0u);
        const uint32_t fp = pfx ? uint32_t(__builtin_ctzll(pfx)) : 64u;     // nearest lane that holds an inclusive prefix
        const unsigned long long need = fp >= 63u ? ~0ull : ((2ull << fp) - 1ull);   // lanes 0 .. fp
        if (emp & need) {                                     // a tile in reach has not published anything yet
            if (++spins > CH_SPIN_MAX) return false;
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        uint64_t v = lane <= fp ? (st & CH_VAL) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        sum += v;
        if (fp < 64u) break;
        top -= 64;
    }
    excl = sum;
    return true;
}

// one symbol through the hot image; escapes (and anything the image does not hold) through the full tables
__device__ __forceinline__ void o2_code_of(const EmitParams &p, const unsigned char *img, uint32_t b2, uint32_t b1, uint32_t sym,
                                           uint32_t &l, uint64_t &c) {
    const uint16_t *ctxmap = reinterpret_cast<const uint16_t *>(img + O2H_MAP_OFF);
    const uint16_t *hot = reinterpret_cast<const uint16_t *>(img + O2H_HOT_OFF);
    const uint32_t i2 = img[b2], i1 = img[b1], i0 = img[sym];
    const uint32_t cs = ctxmap[(i2 << 6) | (i1 ^ i2)];
    const uint32_t e = hot[(cs << 6) | (i0 ^ i1)];
    l = e >> 12;
    c = e & 0xFFFu;
    if (e >= 0xD000u) {
        const uint32_t key = (b2 << 16) | (b1 << 8) | sym;
        l = p.len8[key];
        c = p.code64[key];
        if (l > 64u) { l = 0; c = 0; }
    }
}

__global__ __launch_bounds__(E_THREADS) void enc_chain_kernel(ChainParams cp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_tile[2][E_WAVES], s_base[2];
    __shared__ uint32_t s_done[2], s_tag[2], s_bad[2], s_ticket;
    const EmitParams &p = cp.e;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tab_bytes = (p.o2hot_bytes + 15u) & ~15u;
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem + tab_bytes) + wave * E_STAGE_WORDS;
    if (threadIdx.x < 2) { s_done[threadIdx.x] = 0; s_tag[threadIdx.x] = 0; }
    for (uint32_t i = threadIdx.x; i < tab_bytes / 16u; i += E_THREADS)
        reinterpret_cast<uint4 *>(smem)[i] = reinterpret_cast<const uint4 *>(p.o2hot)[i];
    for (int i = lane; i < E_STAGE_WORDS; i += 64) stage[i] = 0;
    uint32_t *out32 = reinterpret_cast<uint32_t *>(p.out);
    const uint32_t S = 1u << p.chunk_shift;
    const uint64_t cap_bits = cp.cap * 8;
    const uint64_t carry = cp.start_bit ? (*cp.start_bit & 7ull) : 0ull;
    // ---- the groups are handed out in order by a ticket counter: every group a look-back can wait for has been taken by a
    // workgroup that is RUNNING, whatever else holds CUs of the device.  (Dealing them round-robin over a grid assumed to be
    // resident all at once is no faster — 5.60 against 5.58 ms per 4 GiB — and two ranks rehearsing on one card deadlocked
    // each other that way until the bounded waits ran out.)
    if (threadIdx.x == 0) s_ticket = atomicAdd(cp.sync, 1u);
    __syncthreads();
    uint64_t group = s_ticket;
    uint64_t wt = group * E_WAVES + wave;                      // neighbouring tiles run side by side
    LaneIn ahead = load_raw2(p.data, p.n, wt * E_WT + lane * E_VEC, p.prev0);   // (past the end: zeros, nothing read)
#pragma unroll 1
    for (uint32_t round = 0; wt - wave < p.nwt; ++round) {     // (every wave of the workgroup takes part in every group)
        const uint32_t par = round & 1u;
        struct Held { uint64_t g[4]; uint32_t gl, pb; };      // gl: the four group lengths, a byte each (escape sub-step: the lane's bits)
        Held h0{}, h1{}, h2{}, h3{};
        uint32_t escmask = 0;
        uint64_t s = carry, tile_bits = 0;
        bool emit = false;
        if (wt < p.nwt) {
            // ---- everything looked up once: per sub-step four packed groups of four codes, their lengths, the lane's bits.
            // (Both loops over the sub-steps stay rolled — the slow path is inlined once, the registers hold one sub-step's
            // working set beside the four held ones — so the held sets rotate: slot 0 is the oldest.)
            uint32_t lane_bits = 0;
#pragma unroll 1
            for (int k = 0; k < E_SUBSTEPS; ++k) {
                const LaneIn in = ahead;
                if (k + 1 < E_SUBSTEPS) ahead = load_raw2(p.data, p.n, wt * E_WT + uint64_t(k + 1) * E_SUB + lane * E_VEC, p.prev0);
                const uint32_t pb = head_ctx(in);
                h0 = h1; h1 = h2; h2 = h3;
                uint32_t L = 0, glk = 0;
                uint32_t emax = in.nvalid == E_VEC ? 0u : 0xFFFFu;     // ragged vectors take the symbol-by-symbol path
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    uint32_t E[8];
                    const uint32_t c8 = half == 0 ? pb : (((in.x.y >> 16) & 255u) << 8) | (in.x.y >> 24);
                    o2hot_lookup8(smem, half == 0 ? in.x.x : in.x.z, half == 0 ? in.x.y : in.x.w, c8, E);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const uint32_t e0 = E[4 * q], e1 = E[4 * q + 1], e2 = E[4 * q + 2], e3 = E[4 * q + 3];
                        uint32_t m01 = e0 > e1 ? e0 : e1, m23 = e2 > e3 ? e2 : e3;
                        m01 = m01 > m23 ? m01 : m23;
                        emax = m01 > emax ? m01 : emax;
                        const uint32_t l0 = e0 >> 12, l1 = e1 >> 12, l2 = e2 >> 12, l3 = e3 >> 12;
                        const uint32_t p01 = ((e0 & 0xFFFu) << l1) | (e1 & 0xFFFu);
                        const uint32_t p23 = ((e2 & 0xFFFu) << l3) | (e3 & 0xFFFu);
                        h3.g[2 * half + q] = (uint64_t(p01) << (l2 + l3)) | p23;
                        const uint32_t glq = l0 + l1 + l2 + l3;
                        glk |= glq << (8 * (2 * half + q));
                        L += glq;
                    }
                }
                if (__any(emax >= 0xD000u)) {                  // wave-uniform: the lengths symbol by symbol, escapes from the full table
                    escmask |= 1u << k;
                    uint4 x = in.x;                              // rolled, one symbol at a time: rare, and the registers are taken
                    uint32_t ctx = pb;
                    L = 0;
#pragma unroll 1
                    for (uint32_t j = 0; j < 16; ++j) {
                        const uint32_t sym = x.x & 255u;
                        x.x = __builtin_amdgcn_alignbyte(x.y, x.x, 1);
                        x.y = __builtin_amdgcn_alignbyte(x.z, x.y, 1);
                        x.z = __builtin_amdgcn_alignbyte(x.w, x.z, 1);
                        x.w >>= 8;
                        uint32_t l = 0;
                        uint64_t c;
                        if (j < in.nvalid) o2_code_of(p, smem, ctx >> 8, ctx & 255u, sym, l, c);
                        L += l;
                        ctx = ((ctx << 8) | sym) & 0xFFFFu;
                    }
                }
                h3.gl = ((escmask >> k) & 1u) ? L : glk; h3.pb = pb;
                lane_bits += L;
            }
            tile_bits = wave_sum(lane_bits);
        }
        // ---- the tile's place in the stream.  The 16 tiles of the workgroup are one GROUP in the chained scan (sixteen times
        // fewer state words in memory, and the look-backs stay short: at most 256 groups are in flight).  The waves leave their
        // bit counts in LDS; the last one to arrive adds them up, publishes the group's aggregate, looks back over the groups
        // before it and leaves the group's start bit in LDS for the others, who poll LDS, not memory.  Nobody passes a round
        // before its last wave has arrived, so a wave is at most one round ahead of another: two sets of slots.
        // (Measured, 4 GiB of text: one state word per wave-tile 6.2 ms; this 5.65 ms; the first wave to arrive looking back
        // while the others are still looking up 5.9 ms; no look-back at all, wrong output, 5.15 ms.)
        bool waited_out = false;
        {
            const uint32_t tag = round + 1u;
            uint32_t arrived = 0;
            if (lane == 0) {
                s_tile[par][wave] = tile_bits;
                arrived = __hip_atomic_fetch_add(&s_done[par], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            arrived = uint32_t(__builtin_amdgcn_readfirstlane(int(arrived)));
            if (arrived == uint32_t(E_WAVES) - 1u) {
                uint64_t total = lane < uint32_t(E_WAVES) ? s_tile[par][lane] : 0ull;
#pragma unroll
                for (int d = 8; d >= 1; d >>= 1) total += __shfl_xor(total, d);
                total = __shfl(total, 0);
                uint64_t base = carry;
                bool bad = false;
                if (group != 0) {
                    if (lane == 0) __hip_atomic_store(&cp.state[group], CH_AGG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cp.probe == 2u) bad = true;                            // (test hook: as if the wait had run out)
#ifdef MH_EXP_PROBES                                                           /* diagnostic builds only: no look-back, output wrong */
                    else if (cp.probe == 1u) base = group * 320000ull;
#endif
                    else bad = !chain_lookback(cp.state, group, lane, base);  // (group 0's prefix carries the start offset)
                }
                if (lane == 0) {
                    const uint64_t gend = base + total;
                    __hip_atomic_store(&cp.state[group], CH_PFX | (gend & CH_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((group + 1) * E_WAVES >= p.nwt) *cp.nbits = gend;
                    if (bad) atomicExch(cp.status, MHK_STATUS_TIMEOUT);
                    s_done[par] = 0;
                    s_base[par] = base;
                    s_bad[par] = bad ? 1u : 0u;
                    __hip_atomic_store(&s_tag[par], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            // The group's last arrival always publishes s_tag (its own look-back is bounded), so this wait ends; its bound is a
            // second line of defence only, longer than the look-back's (shorter sleeps, hence the factor), and a wave that does
            // run out of it says so: its tile stays unwritten, and nobody may take the payload for valid.
            uint32_t spins = 0;
            const bool follower_hook = cp.probe == 3u && group != 0 && arrived != uint32_t(E_WAVES) - 1u;   // (test hook)
            while (__hip_atomic_load(&s_tag[par], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != tag) {
                if (++spins > CH_SPIN_MAX * 8u || follower_hook) {
                    waited_out = true;
                    if (lane == 0) atomicExch(cp.status, MHK_STATUS_TIMEOUT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (follower_hook && !waited_out) {                    // (the tag was already there: the hook still reports)
                waited_out = true;
                if (lane == 0) atomicExch(cp.status, MHK_STATUS_TIMEOUT);
            }
            uint64_t mine = lane < wave ? s_tile[par][lane] : 0ull;          // the tiles of the group before this wave's
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
            s = s_base[par] + __shfl(mine, 0);
            if (s_bad[par]) waited_out = true;
        }
        if (wt < p.nwt) {
            const uint64_t end = s + tile_bits;
            if (lane == 0 && !waited_out && end > cap_bits) atomicExch(cp.status, MHK_STATUS_CAPACITY);
            emit = !(waited_out || end > cap_bits);            // (wave-uniform) else nothing of this tile is written
        }
        if (waited_out) break;                                 // (the others of the workgroup run into their own bound)
        const uint64_t wt_now = wt;
        __syncthreads();                                       // (everybody has read the ticket before it is replaced)
        if (threadIdx.x == 0) s_ticket = atomicAdd(cp.sync, 1u);
        __syncthreads();
        group = s_ticket;                                      // the next group: whichever is next in line
        wt = group * E_WAVES + wave;
        ahead = load_raw2(p.data, p.n, wt * E_WT + lane * E_VEC, p.prev0);   // the next round's first vectors (past the end: zeros)
        if (emit) {
            // ---- emit from the registers
            uint64_t gbase = s >> 5;                             // output dword under image word 0
            uint32_t cur = uint32_t(s & 31u);                    // image bit where the next code goes
            uint64_t abs_bits = s;
            uint32_t seam0 = (cur != 0 && wt_now != 0) ? SEAM_DROP : SEAM_NONE;   // the tile before this one writes that dword
#pragma unroll 1
            for (int k = 0; k < E_SUBSTEPS; ++k) {
                const uint64_t off = wt_now * E_WT + uint64_t(k) * E_SUB + lane * E_VEC;
                const uint32_t nvalid = off + E_VEC <= p.n ? uint32_t(E_VEC) : off < p.n ? uint32_t(p.n - off) : 0u;
                uint32_t sub_bits;
                if ((escmask >> k) & 1u) {
                    const LaneIn again = load_raw2(p.data, p.n, off, p.prev0);   // (kept out of the registers: rare)
                    emit_substep_slow<2>(p, nullptr, stage, out32, again.x, h0.pb, nvalid, lane, off, abs_bits, gbase, cur, seam0, sub_bits);
                } else {
                    const uint32_t L = (h0.gl & 255u) + ((h0.gl >> 8) & 255u) + ((h0.gl >> 16) & 255u) + (h0.gl >> 24);
                    const uint32_t inc = wave_inclusive_sum(L);
                    sub_bits = __builtin_amdgcn_readlane(inc, 63);
                    const uint32_t exc = inc - L;
                    if (p.index && nvalid && ((uint32_t(off) & (S - 1u)) == 0u))
                        p.index[off >> p.chunk_shift] = (uint64_t(h0.pb) << 48) | (abs_bits + exc);
                    fine2_entry(p, S, lane, off, nvalid, h0.pb, exc);
                    uint32_t o = cur + exc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t glq = (h0.gl >> (8 * q)) & 255u;
                        if (glq) deposit<false>(stage, h0.g[q] << (64u - glq), o, 0, 0);
                        o += glq;
                    }
                    const uint32_t nfull = (cur + sub_bits) >> 5;
                    flush_words(stage, out32, gbase, nfull, seam0, lane);
                    if (nfull) seam0 = SEAM_NONE;
                    gbase += nfull;
                    cur = (cur + sub_bits) & 31u;
                }
                abs_bits += sub_bits;
                h0 = h1; h1 = h2; h2 = h3;
            }
            // ---- the tile's last, partial dword
            if (cur != 0 && seam0 == SEAM_NONE) {               // this tile's to write: filled up with the first bits of what follows
                const uint64_t next = (wt_now + 1) * uint64_t(E_WT);   // (a ragged tile is the last one: nothing follows)
                const uint64_t pos = next + lane;
                const bool valid = lane < 32u && pos < p.n;
                uint32_t l = 0;
                uint64_t c = 0;
                if (valid) o2_code_of(p, smem, p.data[pos - 2], p.data[pos - 1], p.data[pos], l, c);
                const uint32_t inc = wave_inclusive_sum(l);
                const uint32_t exc = inc - l;
                if (l && exc < 32u - cur) deposit<true>(stage, c << (64u - l), cur + exc, 0u, 1u);
                if (lane == 0) {
                    out32[gbase] = __builtin_bswap32(stage[0]);
                    stage[0] = 0;
                }
            } else if (cur != 0 && lane == 0) {                  // the whole tile lies inside a dword of the tile before it
                stage[0] = 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// encode without a length pass (after a region-mode histogram of the same data)
// ------------------------------------------------------------------------------------------------
// The compress path takes a histogram anyway.  When hist_o1_kernel ran in region mode, every workgroup's slab
// plus its crossing list ARE the exact pair counts of its contiguous region, so the region's payload length is
// a dot product with the code lengths (region_bits_kernel), and an exclusive scan over the <= 256 regions gives
// every region its absolute start bit (region_scan_kernel).  enc_region_kernel then gives each workgroup the
// same region: it walks it in rounds of 16 KiB (one 1 KiB piece per wave), and inside a round the waves only
// need each other's bit counts — one LDS exchange — because all 16 deposit into ONE image shared by the
// workgroup (LDS atomics merge the seams between waves exactly as they merge them between lanes), which the
// whole workgroup then flushes with coalesced stores.  No second read of the input, no per-tile offsets in
// HBM: traffic is the algorithmic (1 + r) n.  Two workgroup barriers per round are the price.
// Only for models without escape codes (max length <= 12); others take the three-kernel path.
// the image: all the LDS the codeword table leaves.  A round whose bits fit HALF of it alternates between the halves with its
// neighbours (one barrier per round, see enc_region_kernel); any other round takes the whole image (two barriers).
constexpr int R_IMG_WORDS = ((163840 - 131072 - 128) / 4) & ~7;             // 8160 words >= 16 pieces of <= 12288 bits + carry + slack
constexpr int R_HALF_WORDS = R_IMG_WORDS / 2;
constexpr uint32_t R_HALF_CAP_BITS = uint32_t(R_HALF_WORDS - 8) * 32u;    // (the carried partial word and the flush's 16-byte groups stay inside)
static_assert(R_IMG_WORDS >= E_WAVES * (E_STAGE_BITS / 32) + 16 && R_HALF_WORDS % 4 == 0, "image size");
constexpr int REGION_LDS_BYTES = 131072 + R_IMG_WORDS * 4 + 128;           // + the waves' piece counts, two rounds' worth



// one workgroup per region: bits = sum over pairs of (slab field + 16384 x crossings) x code length
__global__ __launch_bounds__(1024) void region_bits_kernel(const HistHeader *hdr, HistHeader expect, const uint32_t *slab,
                                                           const uint32_t *cross_all, const uint8_t *len8,
                                                           unsigned long long *region_bits, uint32_t *region_esc, int *status) {
    __shared__ unsigned long long part[16];
    __shared__ uint32_t any_esc;
    // [r4] the 64 KiB of code lengths go through LDS (dynamic: REGION_BITS_LDS): in slab order the pairs' lengths lie 256 bytes
    // apart in len8, and two such byte gathers per word made this little kernel take 60 us (1.2 % of a 2 GiB shard's step)
    extern __shared__ __attribute__((aligned(16))) unsigned char lens[];
    for (uint32_t i = threadIdx.x; i < 65536u / 16u; i += 1024u) reinterpret_cast<uint4 *>(lens)[i] = reinterpret_cast<const uint4 *>(len8)[i];
    if (threadIdx.x == 0) any_esc = 0;
    __syncthreads();
    const uint32_t w = blockIdx.x, tid = threadIdx.x;
    if (hdr->magic != HIST_WS_MAGIC || hdr->n != expect.n || hdr->data != expect.data || hdr->region_vecs != expect.region_vecs ||
        hdr->grid != expect.grid || hdr->prev0 != expect.prev0 || hdr->cross_cap != expect.cross_cap) {
        if (tid == 0) { atomicExch(status, MHK_STATUS_CORRUPT); region_bits[w] = 0; region_esc[w] = 0; }   // not the histogram of this input
        return;
    }
    unsigned long long acc = 0;
    bool esc = false;                            // a pair of this region has a code the 12-bit table does not hold
    const uint32_t *sl = slab + size_t(w) * 32768u;
    for (uint32_t i = tid; i < 32768u; i += 1024u) {
        const uint32_t v = sl[i];
        const uint32_t s0 = i, s1 = i | 0x8000u;
        const uint32_t l0 = lens[hist_slot_prev(s0) * 256u + (s0 >> 8)], l1 = lens[hist_slot_prev(s1) * 256u + (s1 >> 8)];
        acc += (unsigned long long)(v & 0xFFFFu) * l0;
        acc += (unsigned long long)(v >> 16) * l1;
        esc |= ((v & 0xFFFFu) && l0 > uint32_t(mh::ENC16_MAX_LEN)) || ((v >> 16) && l1 > uint32_t(mh::ENC16_MAX_LEN));
    }
    const uint32_t *cross = cross_all + size_t(w) * (expect.cross_cap + 1u);
    const uint32_t nc = cross[0];
    if (nc > expect.cross_cap && tid == 0) atomicExch(status, MHK_STATUS_CAPACITY);
    for (uint32_t i = tid; i < (nc < expect.cross_cap ? nc : expect.cross_cap); i += 1024u) {
        const uint32_t sl2 = cross[1u + i];
        const uint32_t l2 = lens[hist_slot_prev(sl2) * 256u + (sl2 >> 8)];
        acc += 16384ull * l2;
        esc |= l2 > uint32_t(mh::ENC16_MAX_LEN);
    }
    if (esc) atomicOr(&any_esc, 1u);
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((tid & 63u) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < 16; ++i) t += part[i];
        region_bits[w] = t;
        region_esc[w] = any_esc;
    }
}

// one block: exclusive scan of the region lengths (<= 1024 regions); total, capacity check, and the dwords that
// two regions share (or that end the stream) are zeroed: they are completed with atomic ORs
__global__ __launch_bounds__(SCAN_THREADS) void region_scan_kernel(const unsigned long long *region_bits, uint32_t nregion,
                                                                   unsigned long long *region_start, const unsigned long long *carry0,
                                                                   uint8_t *out, uint64_t cap, unsigned long long *nbits, int *status) {
    __shared__ uint64_t lds[SCAN_THREADS / 64];
    const uint64_t c0 = carry0 ? (*carry0 & 7ull) : 0;
    const uint32_t i = threadIdx.x;
    const uint64_t v = i < nregion ? region_bits[i] : 0;
    uint64_t total;
    const uint64_t ex = block_excl_scan(v, lds, total);
    const uint64_t s = c0 + ex;
    if (i < nregion) {
        region_start[i] = s;
        if (i > 0 && (s & 31u) && ((s >> 5) + 1) * 4 <= cap) reinterpret_cast<uint32_t *>(out)[s >> 5] = 0;
    }
    if (i == 0) {
        const uint64_t end = c0 + total;
        *nbits = end;
        if (end > cap * 8) atomicExch(status, MHK_STATUS_CAPACITY);
        const uint64_t endw = end >> 5;
        if ((end & 31u) && (endw + 1) * 4 <= cap) reinterpret_cast<uint32_t *>(out)[endw] = 0;
        else if (end & 31u) for (uint64_t b = endw * 4; b < cap; ++b) out[b] = 0;
        if ((c0 & 31u) && 4 <= cap) reinterpret_cast<uint32_t *>(out)[0] = 0;      // the shard's own first dword (pre-shift)
    }
}

struct RegionParams {
    const unsigned long long *region_start;
    const unsigned long long *region_bits;   // what region_bits_kernel priced each region at
    const uint32_t *region_esc;   // per region: != 0 when its histogram has pairs whose codes exceed 12 bits
    uint64_t region_vecs;         // vectors (16 bytes) per region, a multiple of 1024
    uint64_t nvec_up;             // ceil(n / 16)
    uint64_t cap_words;           // output dwords that may be stored (capacity / 4)
    int *status;                  // writable: a region that emits something else than it was priced at reports MHK_STATUS_CORRUPT
};
constexpr uint32_t R_IMG_CAP_BITS = E_WAVES * E_STAGE_BITS;      // what one round may deposit beside the carried partial word

// One symbol of a lane's vector at a time (the escape path: codes over 12 bits come from the full tables in L2).
struct Roll1 {
    uint4 x; uint32_t prev;
    __device__ __forceinline__ uint32_t next_window() {         // sym << 8 | prev, the raw 16-bit field of the stream
        const uint32_t sym = x.x & 255u;
        const uint32_t win = (sym << 8) | prev;
        prev = sym;
        x.x = __builtin_amdgcn_alignbyte(x.y, x.x, 1);
        x.y = __builtin_amdgcn_alignbyte(x.z, x.y, 1);
        x.z = __builtin_amdgcn_alignbyte(x.w, x.z, 1);
        x.w >>= 8;
        return win;
    }
};
__device__ __forceinline__ void code_of1(const uint8_t *len8, const uint64_t *code64, const uint16_t *tab, uint32_t win, bool valid, uint32_t &l, uint64_t &c) {
    const uint32_t e = valid ? uint32_t(tab[mh::enc_slot(win)]) : 0u;
    l = e >> 12;
    c = e & 0xFFFu;
    if (e >= 0xD000u) {                                          // ENC16_ESCAPE: longer than 12 bits
        const uint32_t nat = ((win & 255u) << 8) | (win >> 8);   // prev * 256 + sym
        l = len8[nat];
        c = code64[nat];
        if (l > 64u) { l = 0; c = 0; }                           // rejected on the host
    }
}
// bits of the lane's vector / its codes OR-ed into the image from bit `o` on, symbol by symbol
__device__ __forceinline__ uint32_t region_escape_bits(const uint8_t *len8, const uint64_t *code64, const uint16_t *tab, uint4 x, uint32_t pb, uint32_t nvalid) {
    Roll1 r{x, pb};
    uint32_t L = 0;
#pragma unroll 1
    for (uint32_t j = 0; j < 16; ++j) { uint32_t l; uint64_t c; code_of1(len8, code64, tab, r.next_window(), j < nvalid, l, c); L += l; }
    return L;
}
__device__ __forceinline__ void region_escape_deposit(const uint8_t *len8, const uint64_t *code64, const uint16_t *tab, uint32_t *img, uint4 x,
                                                   uint32_t pb, uint32_t nvalid, uint32_t o) {
    Roll1 r{x, pb};
#pragma unroll 1
    for (uint32_t j = 0; j < 16; ++j) {
        uint32_t l; uint64_t c;
        code_of1(len8, code64, tab, r.next_window(), j < nvalid, l, c);
        if (l) deposit<false>(img, c << (64u - l), o, 0, 0);
        o += l;
    }
}

// [r5] lanes per round of a region of `nsym` symbols priced at `bits`: 1024, or — where the mean round of 1024 vectors does not fit
// half the image — the largest of 1008 / 992 / 960 whose mean round fits with half a per cent to spare; 1024 when none does
// (more than ~8.4 bits per symbol: both barriers, all lanes)
__device__ __forceinline__ uint32_t region_round_lanes(uint64_t nsym, uint64_t bits) {
    const uint32_t cand[4] = {1024u, 1008u, 992u, 960u};
    uint32_t pick = 0;
#pragma unroll
    for (int c = 3; c >= 0; --c)                                  // bits / nsym * (cand * 16) * 1.005 <= cap - 32
        if (bits * cand[c] * E_VEC * 201u <= uint64_t(R_HALF_CAP_BITS - 32u) * nsym * 200u) pick = cand[c];
    return pick ? pick : 1024u;
}

// ESCK: the kernel is launched twice; a workgroup takes its region in the launch that matches the region's escape flag
// (one function with both round bodies spilled registers; a workgroup of the other kind leaves at once)
#ifdef MH_ENC_STAMP
// diagnostic build only (make exp EXPFLAGS=-DMH_ENC_STAMP): shader-clock stamps around the round's phases
#define ENC_STAMP(i) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
        stamp_acc[i] += t_ - stamp_last; stamp_last = t_; } while (0)
#else
#define ENC_STAMP(i) do { } while (0)
#endif
// WIDE [r5]: the launch for the regions whose rounds run with fewer than 1024 lanes (below); the other two instantiations keep
// the round length a compile-time constant
template <bool ESCK, bool WIDE = false>
__global__ __launch_bounds__(E_THREADS) void enc_region_kernel(EmitParams p, RegionParams rp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((rp.region_esc[blockIdx.x] != 0) != ESCK) return;
    {   // which launch takes this region (workgroup-uniform; before anything is staged)
        const uint64_t v0 = uint64_t(blockIdx.x) * rp.region_vecs;
        const uint64_t v1 = v0 + rp.region_vecs < rp.nvec_up ? v0 + rp.region_vecs : rp.nvec_up;
        if (v0 < v1) {
            const uint32_t rs = ESCK ? uint32_t(E_THREADS) : region_round_lanes((v1 * E_VEC < p.n ? v1 * E_VEC : p.n) - v0 * E_VEC, rp.region_bits[blockIdx.x]);
            if ((rs != uint32_t(E_THREADS)) != WIDE) return;
        } else if (WIDE) return;
    }
    uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
    uint32_t *img = reinterpret_cast<uint32_t *>(smem + 131072);
    uint32_t *sb = img + R_IMG_WORDS;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (int i = tid; i < 8192; i += E_THREADS)
        reinterpret_cast<uint4 *>(tab)[i] = reinterpret_cast<const uint4 *>(p.enc16)[i];
    for (int i = tid; i < R_IMG_WORDS; i += E_THREADS) img[i] = 0;
    __syncthreads();
    if (*p.status != MHK_STATUS_OK) return;     // capacity overrun or a foreign histogram: write nothing

    const uint64_t v0 = uint64_t(blockIdx.x) * rp.region_vecs;
    const uint64_t v1 = v0 + rp.region_vecs < rp.nvec_up ? v0 + rp.region_vecs : rp.nvec_up;
    if (v0 >= v1) return;                        // empty region (uniform for the workgroup)
    uint32_t *out32 = reinterpret_cast<uint32_t *>(p.out);
    const uint32_t S = 1u << p.chunk_shift;
    // [r5] vectors per round, a choice of the region (workgroup-uniform).  A round of all 1024 lanes is 16 KiB of input; at 8 bits
    // per symbol that is 131 072 bits, 768 more than half the image holds (R_HALF_CAP_BITS), so uniform bytes — BASELINE config 4
    // — kept both barriers of every round.  A region whose mean round does not fit a half with 1024 lanes runs its rounds with
    // the first 1008, 992 or 960 lanes (the largest count whose mean round fits with half a per cent to spare: 129 024 bits at 8
    // bits per symbol); the other lanes of the last wave idle (zero-length entries, no index entries).  Multiples of 4: a fine
    // index entry belongs to every fourth vector, which must stay every fourth lane.
    // (region_round_lanes above; this launch was chosen by it.)  The launches with WIDE = false keep RS a constant.
    const uint32_t RS = WIDE ? region_round_lanes((v1 * E_VEC < p.n ? v1 * E_VEC : p.n) - v0 * E_VEC, rp.region_bits[blockIdx.x]) : uint32_t(E_THREADS);
    const bool act = !WIDE || tid < RS;                           // this lane takes a vector of every round
    const bool wave_idles = WIDE && uint32_t(wave) * 64u + 63u >= RS;     // (wave-uniform) some lane of this wave does not
    const uint64_t rounds = (v1 - v0 + RS - 1) / RS;
    const uint64_t s0 = rp.region_start[blockIdx.x];
    uint64_t gbase = s0 >> 5, abs_round = s0;
    uint32_t cur = uint32_t(s0 & 31u);
    bool seam_first = cur != 0;                  // the region's first dword is shared with its predecessor
    // The image's partial last word travels from round to round in a REGISTER of the thread that read (and
    // cleared) its 16-byte group during the flush, and is OR-ed back into word 0 behind the next round's first
    // barrier: no thread ever reads a word that another thread's clear may touch in the same phase.
    uint32_t carry = 0;
    bool prev_half = false;                      // the previous round used a half of the image (see round())

    auto fetch = [&](uint64_t r) -> LaneIn {
        const uint64_t v = v0 + r * RS + tid;
        LaneIn in = load_raw(p.data, p.n, (act && v < v1) ? v * E_VEC : ~0ull >> 1, p.prev0);   // beyond the region, an idle lane: nothing
        return in;
    };
    auto lookup16 = [&](const LaneIn &in, uint32_t pb, uint32_t (&e)[16]) {
        uint32_t w[16];
        slots16(in.x, pb, w);
#pragma unroll
        for (int j = 0; j < 16; ++j) e[j] = uint32_t(tab[w[j]]);
    };
    // the workgroup stores image words [0, nfull) to output dwords gbase + j (coalesced, MSB-first bytes) and
    // clears them; four words per lane: one 16-byte LDS read, one 16-byte clear, one 16-byte store.  The group
    // that holds word nfull (the partial tail) is visited too: its reader returns that word.
    uint32_t *imgr = img;                        // the image of the round at hand: the whole one, or one of its halves
    auto flush = [&](uint32_t nfull) -> uint32_t {
        uint32_t tail = 0;
        for (uint32_t j = tid * 4u; j <= nfull; j += E_THREADS * 4u) {
            const uint4 w = *reinterpret_cast<const uint4 *>(imgr + j);
            *reinterpret_cast<uint4 *>(imgr + j) = make_uint4(0, 0, 0, 0);
            const uint32_t v[4] = {__builtin_bswap32(w.x), __builtin_bswap32(w.y), __builtin_bswap32(w.z), __builtin_bswap32(w.w)};
            if (j + 4u <= nfull && !(j == 0 && seam_first) && gbase + j + 4u <= rp.cap_words) {
                struct __attribute__((packed, aligned(4))) Q4 { uint32_t a, b, c, d; };
                *reinterpret_cast<Q4 *>(out32 + gbase + j) = Q4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    if (j + k >= nfull || gbase + j + k >= rp.cap_words) break;      // (beyond the capacity: only when the histogram was not this input's)
                    if (j + k == 0 && seam_first) atomicOr(&out32[gbase], v[k]);
                    else out32[gbase + j + k] = v[k];
                }
            }
            if (nfull - j < 4u) {
                const uint32_t k = nfull - j;
                tail = k == 0 ? w.x : k == 1 ? w.y : k == 2 ? w.z : w.w;
            }
        }
        return tail;
    };
#ifdef MH_ENC_STAMP
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_last = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last) :: "memory");
#endif
    // ---- the round pipeline ------------------------------------------------------------------------------------
    // Stamps of the two-phase version (profiles/r03/enc_stamps_*.txt): a round spent 1800 cycles packing (vector
    // ALU, LDS idle), 900 at the first barrier, 1700 behind its deposits (the LDS working through ~160 atomic
    // wave-instructions, vector ALU idle), 600 at the second barrier and 850 flushing.  Deposits return nothing, so a
    // wave can issue them and go on: round r + 1 is therefore PACKED between round r's deposits and the barrier
    // that ends them — the vector ALU packs while the LDS ORs.
    //   top of round r:   P = round r packed (groups, lengths, inclusive scan), its bit count in sb[r & 1];
    //                     E1 = codeword entries of round r + 1; D1, D2, D3 = input of rounds r + 1 .. r + 3
    //   barrier 1         counts of round r visible, image free (every wave has flushed round r - 1)
    //   exchange, index entries, deposits of round r (issued, not awaited)
    //   fetch r + 4, lookups of round r + 2, pack + scan of round r + 1, its count to sb[(r + 1) & 1]
    //   barrier 2         deposits of round r done
    //   flush round r
    struct Packed { uint64_t g[4]; uint32_t gl[4]; uint32_t L, inc; bool esc; };
    auto pack = [&](const uint32_t (&E)[16], const LaneIn &in, uint32_t pb, auto full_c, auto esc_c) __attribute__((always_inline)) -> Packed {
        constexpr bool FULL = decltype(full_c)::value, ESC = decltype(esc_c)::value;
        Packed P;
        uint32_t e[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) e[j] = (FULL || uint32_t(j) < in.nvalid) ? E[j] : 0u;   // the stream's ragged last vector, lanes past the region's end
        if (FULL && wave_idles) {                                 // (wave-uniform: the last wave of a region with fewer than 1024 lanes per round)
#pragma unroll
            for (int j = 0; j < 16; ++j) e[j] = act ? e[j] : 0u;
        }
        P.L = 0;
        uint32_t emax = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t e0 = e[4 * q], e1 = e[4 * q + 1], e2 = e[4 * q + 2], e3 = e[4 * q + 3];
            if (ESC) {
                const uint32_t m01 = e0 > e1 ? e0 : e1, m23 = e2 > e3 ? e2 : e3;
                emax = emax > m01 ? emax : m01;
                emax = emax > m23 ? emax : m23;
            }
            const uint32_t l0 = e0 >> 12, l1 = e1 >> 12, l2 = e2 >> 12, l3 = e3 >> 12;
            const uint32_t p01 = ((e0 & 0xFFFu) << l1) | (e1 & 0xFFFu);
            const uint32_t p23 = ((e2 & 0xFFFu) << l3) | (e3 & 0xFFFu);
            P.g[q] = (uint64_t(p01) << (l2 + l3)) | p23;
            P.gl[q] = l0 + l1 + l2 + l3;
            P.L += P.gl[q];
        }
        // a code of more than 12 bits among the lane's 16 (entry ENC16_ESCAPE): that lane prices and deposits its
        // symbols one by one from the full tables (src/bitbuffer.cpp:45-73 appends descriptors of any length)
        P.esc = ESC && emax >= 0xD000u;
        if (ESC && __any(P.esc)) {               // wave-uniform
            if (P.esc) P.L = region_escape_bits(p.len8, p.code64, tab, in.x, pb, FULL ? uint32_t(E_VEC) : in.nvalid);
        }
        P.inc = wave_inclusive_sum(P.L);
        return P;
    };
    auto fetch_full = [&](uint64_t r) -> LaneIn {                // round r is whole: no bounds checks
        LaneIn in;
        const uint64_t v = v0 + r * RS + tid;                     // (an idle lane reads a vector of the next round: inside the stream, unused)
        in.x = reinterpret_cast<const uint4 *>(p.data)[v];
        in.nvalid = act ? uint32_t(E_VEC) : 0u;                   // (the last rounds of a region run with bounds checks on vectors loaded here)
        in.head = p.prev0;
        if (lane == 0 && v) in.head = uint32_t(p.data[v * E_VEC - 1]);
        return in;
    };
    using ESC_T = std::integral_constant<bool, ESCK>;
    LaneIn D0 = fetch(0), D1 = fetch(1), D2 = fetch(2), D3 = fetch(3);
    uint32_t pb0 = head_byte(D0), pb1 = head_byte(D1);
    uint32_t E1[16];
    Packed P;                                    // loop-carried: round r packed
    {
        uint32_t E0[16];
        lookup16(D0, pb0, E0);
        lookup16(D1, pb1, E1);
        P = pack(E0, D0, pb0, std::false_type{}, ESC_T{});
    }
    if (lane == 63) sb[wave] = P.inc;            // round 0's piece count
    uint4 x0 = D0.x;                             // round r's input (the escape path re-reads it) and valid bytes
    uint32_t nvalid0 = D0.nvalid;
    static_assert(E_WAVES == 16, "the scan below is one DPP row");
    // Ea holds the entries of round r + 1 (packed here), Eb receives those of round r + 2: the steady-state loop runs two
    // rounds per trip with the two arrays swapping roles, so the sixteen entries are never copied
    auto round = [&](uint64_t r, auto full_c, uint32_t (&Ea)[16], uint32_t (&Eb)[16]) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value, ESC = ESCK;
        uint32_t *sbr = sb + (r & 1u) * 16u, *sbn = sb + ((r + 1u) & 1u) * 16u;
        ENC_STAMP(0);                            // flush of the previous round (+ loop overhead)
        // [r4] ONE barrier per round where the rounds fit half the image.  The first barrier orders two things: the waves'
        // bit counts of this round (written before the previous round's second barrier: visible without it) and "every wave
        // has flushed round r - 1" before anything of round r is deposited.  With round r in the OTHER half of the image than
        // round r - 1 the second needs no barrier: half (r & 1) was last flushed for round r - 2, and every wave finished that
        // flush before it reached round r - 1's second barrier, which lies behind us.  A round that does not fit a half (more
        // than ~7.9 bits per symbol: uniform bytes), the round behind one, the first round and the escape kernel keep both.
        uint32_t cs = sbr[lane & 15u];
        if (r == 0 || !prev_half) { __syncthreads(); cs = sbr[lane & 15u]; }
        ENC_STAMP(1);                            // barrier 1 (if any)
        // bits of the round in front of this wave / in the whole round: every row of 16 lanes scans the 16 counts
        cs += uint32_t(__builtin_amdgcn_update_dpp(0, int(cs), 0x111, 0xF, 0xF, true));      // row_shr:1
        cs += uint32_t(__builtin_amdgcn_update_dpp(0, int(cs), 0x112, 0xF, 0xF, true));      // row_shr:2
        cs += uint32_t(__builtin_amdgcn_update_dpp(0, int(cs), 0x114, 0xF, 0xF, true));      // row_shr:4
        cs += uint32_t(__builtin_amdgcn_update_dpp(0, int(cs), 0x118, 0xF, 0xF, true));      // row_shr:8
        const uint32_t tot = uint32_t(__builtin_amdgcn_readlane(int(cs), 15));
        const uint32_t pre = wave ? uint32_t(__builtin_amdgcn_readlane(int(cs), int(wave) - 1)) : 0u;
        // (workgroup-uniform: cur and tot are) this round in a half of its own?  If the previous one was not, its flush of the
        // whole image may still be running: the barrier above was taken (prev_half false) and the halves are free again.
        const bool half = !ESC && cur + tot <= R_HALF_CAP_BITS;
        if (!half && r != 0 && prev_half) __syncthreads();       // a whole-image round behind a half round: wait for that flush
        imgr = half ? img + (r & 1u) * uint32_t(R_HALF_WORDS) : img;
        prev_half = half;
        if (carry) { atomicOr(&imgr[0], carry); carry = 0; }     // the previous round's partial word (the words it lands in are free: see above)
        const uint32_t exc = pre + P.inc - P.L;  // bits of the round in front of this lane
        const uint64_t off = (v0 + r * RS + tid) * E_VEC;
        const uint32_t nvalid = FULL ? (act ? uint32_t(E_VEC) : 0u) : nvalid0;
        if (p.index && nvalid && ((uint32_t(off) & (S - 1u)) == 0u))
            p.index[off >> p.chunk_shift] = (uint64_t(pb0) << 56) | (abs_round + exc);
        if (p.fine && nvalid && (lane & (uint32_t(1u << T_SUB_SHIFT) / 16u - 1u)) == 0u)               // fine index (mh_kernels.h, TileParams): every fourth lane
            p.fine[off >> T_SUB_SHIFT] = (pb0 << 24) | (uint32_t(abs_round + exc) & FINE_POS_MASK);
        const bool fits = !ESC || cur + tot <= R_IMG_CAP_BITS;   // workgroup-uniform: the round fits the image (always, without escapes)
        if (fits) {
            uint32_t o = cur + exc;
            if (!ESC || !P.esc) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (P.gl[q]) deposit<false>(imgr, P.g[q] << (64u - P.gl[q]), o, 0, 0);
                    o += P.gl[q];
                }
            } else {
                region_escape_deposit(p.len8, p.code64, tab, imgr, x0, pb0, nvalid, o);
            }
        } else {
            // more bits than the image holds (only a model with many codes far over 12 bits can do that): one
            // wave's piece at a time — at most 1024 x 64 bits — each deposited symbol by symbol and flushed
            for (uint32_t m = 0; m < uint32_t(E_WAVES); ++m) {
                const uint32_t upto = uint32_t(__builtin_amdgcn_readlane(int(cs), int(m)));
                const uint32_t before = m ? uint32_t(__builtin_amdgcn_readlane(int(cs), int(m) - 1)) : 0u;
                if (wave == m) region_escape_deposit(p.len8, p.code64, tab, img, x0, pb0, nvalid, cur + (exc - pre));
                __syncthreads();
                const uint32_t nfull = (cur + (upto - before)) >> 5;
                const uint32_t t = flush(nfull);
                seam_first = seam_first && nfull == 0;
                gbase += nfull;
                cur = (cur + (upto - before)) & 31u;
                __syncthreads();                 // the flush's clears are done before anything is OR-ed in again
                if (t) atomicOr(&img[0], t);
            }
        }
        ENC_STAMP(2);                            // exchange + deposits issued
        // ---- while the LDS works the deposits off: the next rounds
        // ([r5] the previous round's flush was moved here as well — a half round's flush deferred until the next round's deposits
        // are on their way, its reads queued behind them — and measured SLOWER, 9.44 against 8.83 ms per 16 GiB, 2.475 against 2.405
        // per 4 GiB: profiles/r05/encoder/README.md, commit 7159b51; a wave's flush reads wait for its own ~10 atomics per lane)
        const LaneIn D4 = FULL ? fetch_full(r + 4) : fetch(r + 4);
        const uint32_t pb2 = head_byte(D2);
        lookup16(D2, pb2, Eb);
        const Packed Pn = pack(Ea, D1, pb1, full_c, ESC_T{});
        if (lane == 63) sbn[wave] = Pn.inc;      // round r + 1's piece count (the other half of sb: round r's is still being read)
        ENC_STAMP(3);                            // lookups issued + pack + scan of the next round
        if (fits) {
            __syncthreads();
            ENC_STAMP(4);                        // barrier 2
            const uint32_t nfull = (cur + tot) >> 5;
            carry = flush(nfull);
            // no barrier here: the next round touches the image only behind ITS first barrier, which every wave
            // reaches after its share of this flush
            seam_first = seam_first && nfull == 0;
            gbase += nfull;
            cur = (cur + tot) & 31u;
        }
        abs_round += tot;
        P = Pn;
        x0 = D1.x; nvalid0 = D1.nvalid; pb0 = pb1;
        D1 = D2; pb1 = pb2;
        D2 = D3;
        D3 = D4;
    };
    // leading rounds whose 16 KiB, and those of the four rounds behind them, are whole: the steady state
    const uint64_t whole = (p.n >> 4) < v1 ? (p.n >> 4) : v1;       // vectors with all 16 bytes inside the stream
    const uint64_t rounds_full = whole > v0 + (E_THREADS - RS) ? (whole - v0 - (E_THREADS - RS)) / RS : 0;   // (idle lanes read up to 64 vectors further)
    const uint64_t r_fast = rounds_full > 4 ? rounds_full - 4 : 0;
    uint64_t r = 0;
    uint32_t E2[16];
#pragma unroll 1
    for (; r + 1 < r_fast; r += 2) {
        round(r, std::true_type{}, E1, E2);
        round(r + 1, std::true_type{}, E2, E1);
    }
#pragma unroll 1
    for (; r < rounds; ++r) {
        round(r, std::false_type{}, E1, E2);
#pragma unroll
        for (int j = 0; j < 16; ++j) E1[j] = E2[j];
    }
    // the region's last partial dword: shared with the next region (or the stream's end), zeroed by the scan
    if (cur != 0 && gbase < rp.cap_words) {
        if (carry) atomicOr(&out32[gbase], __builtin_bswap32(carry));
        else if (tid == 0 && img[0]) atomicOr(&out32[gbase], __builtin_bswap32(img[0]));   // (left by the piece-by-piece path)
    }
    // The region was priced from the histogram workspace; if the buffer was refilled between the histogram and
    // this call the counts are another input's and the regions overlap or leave gaps: say so.
    if (tid == 0 && abs_round != s0 + rp.region_bits[blockIdx.x]) atomicExch(rp.status, MHK_STATUS_CORRUPT);
#ifdef MH_ENC_STAMP
    if (lane == 0)                               // cycle sums per phase, over all waves: bytes 8..47 of the status block
        for (int i = 0; i < 5; ++i) atomicAdd(reinterpret_cast<unsigned long long *>(rp.status) + 1 + i, stamp_acc[i]);
#endif
}


// ---- encode, pass 1: p.len_slot = len8[ctx * 256 + sym] (HBM), p.prev0 = the 16-bit start context
__global__ __launch_bounds__(E_THREADS) void enc2_len_kernel(LenParams p) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave0 = uint64_t(blockIdx.x) * E_WAVES + (threadIdx.x >> 6);
    const uint64_t nwaves = uint64_t(gridDim.x) * E_WAVES;
    for (uint64_t wt = wave0; wt < p.nwt; wt += nwaves) {
        uint32_t sum = 0;
#pragma unroll 1
        for (int k = 0; k < E_SUBSTEPS; ++k) {
            const LaneIn in = load_raw2(p.data, p.n, wt * E_WT + uint64_t(k) * E_SUB + lane * E_VEC, p.prev0);
            uint32_t ctx = head_ctx(in);
            const uint32_t x[4] = {in.x.x, in.x.y, in.x.z, in.x.w};
            uint32_t l[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {                         // 16 independent gathers in flight
                const uint32_t key = (ctx << 8) | ((x[j >> 2] >> (8 * (j & 3))) & 255u);
                l[j] = uint32_t(j) < in.nvalid ? uint32_t(p.len_slot[key]) : 0u;
                ctx = key & 0xFFFFu;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) sum += l[j] > 64u ? 0u : l[j];
        }
        sum = wave_sum(sum);
        if (lane == 0) p.wt_bits[wt] = sum;
    }
}

// ---- encode, pass 2
__global__ __launch_bounds__(E_THREADS) void enc2_emit_kernel(EmitParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem) + wave * E_STAGE_WORDS;
    for (int i = lane; i < E_STAGE_WORDS; i += 64) stage[i] = 0;
    __syncthreads();
    if (*p.status != MHK_STATUS_OK) return;     // capacity overrun found by the scan: write nothing
    uint32_t *out32 = reinterpret_cast<uint32_t *>(p.out);
    const uint64_t wave0 = uint64_t(blockIdx.x) * E_WAVES + wave;
    const uint64_t nwaves = uint64_t(gridDim.x) * E_WAVES;
    for (uint64_t wt = wave0; wt < p.nwt; wt += nwaves) {
        const uint64_t s = p.wt_start[wt];
        uint64_t gbase = s >> 5;
        uint32_t cur = uint32_t(s & 31u);
        uint64_t abs_bits = s;
        uint32_t seam0 = cur != 0 ? SEAM_OR : SEAM_NONE;
#pragma unroll 1
        for (int k = 0; k < E_SUBSTEPS; ++k) {
            const uint64_t off = wt * E_WT + uint64_t(k) * E_SUB + lane * E_VEC;
            const LaneIn in = load_raw2(p.data, p.n, off, p.prev0);
            const uint32_t ctx0 = head_ctx(in);
            // all 16 (length, codeword) pairs of the lane are gathered at once: 32 loads in flight instead of
            // one dependent round trip per symbol (first version: 150 GB/s, bound by exactly that latency)
            uint32_t l[16];
            uint64_t c[16];
            const uint32_t x[4] = {in.x.x, in.x.y, in.x.z, in.x.w};
            bool escape = p.enc64 == nullptr;
            if (p.enc64) {                       // one 8-byte gather per symbol: length in the top byte
                uint32_t ctx = ctx0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t key = (ctx << 8) | ((x[j >> 2] >> (8 * (j & 3))) & 255u);
                    const uint64_t e = uint32_t(j) < in.nvalid ? p.enc64[key] : 0ull;
                    l[j] = uint32_t(e >> 56);
                    c[j] = e & 0x00FFFFFFFFFFFFFFull;
                    escape = escape || l[j] == 255u;
                    ctx = key & 0xFFFFu;
                }
            }
            if (__any(escape)) {                 // a code of more than 56 bits somewhere in the wave (or no packed table)
                uint32_t ctx = ctx0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t key = (ctx << 8) | ((x[j >> 2] >> (8 * (j & 3))) & 255u);
                    const bool valid = uint32_t(j) < in.nvalid;
                    l[j] = valid ? uint32_t(p.len8[key]) : 0u;
                    c[j] = valid ? p.code64[key] : 0ull;
                    ctx = key & 0xFFFFu;
                }
            }
            uint32_t L = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) { if (l[j] > 64u) l[j] = 0; L += l[j]; }
            const uint32_t inc = wave_inclusive_sum(L);
            uint32_t sub_bits = __builtin_amdgcn_readlane(inc, 63);
            if (cur + sub_bits <= uint32_t(E_STAGE_WORDS - 3) * 32u) {       // the usual case: the sub-step fits the image
                const uint32_t exc = inc - L;
                const uint32_t S = 1u << p.chunk_shift;
                if (p.index && in.nvalid && ((uint32_t(off) & (S - 1u)) == 0u))
                    p.index[off >> p.chunk_shift] = (uint64_t(ctx0) << 48) | (abs_bits + exc);
                fine2_entry(p, S, lane, off, in.nvalid, ctx0, exc);
                uint32_t o = cur + exc;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (l[j]) deposit<false>(stage, c[j] << (64u - l[j]), o, 0, 0);
                    o += l[j];
                }
                const uint32_t nfull = (cur + sub_bits) >> 5;
                flush_words(stage, out32, gbase, nfull, seam0, lane);
                if (nfull) seam0 = SEAM_NONE;
                gbase += nfull;
                cur = (cur + sub_bits) & 31u;
            } else {                                                          // very long codes: fill and flush in rounds
                emit_substep_slow<2>(p, nullptr, stage, out32, in.x, ctx0, in.nvalid, lane, off, abs_bits, gbase, cur, seam0, sub_bits);
            }
            abs_bits += sub_bits;
        }
        if (cur != 0 && lane == 0) {
            atomicOr(&out32[gbase], __builtin_bswap32(stage[0]));
            stage[0] = 0;
        }
    }
}

uint64_t encode_wave_tiles(uint64_t n) { return (n + E_WT - 1) / E_WT; }

// workspace: [0,64) status | wt_bits u32[nwt] | wt_start u64[nwt] | blk_sum u64[nblk + 1]
struct EncWs { size_t off_bits, off_start, off_blk, total; uint64_t nwt, nblk; };
static EncWs enc_ws_layout(uint64_t n) {
    EncWs w;
    w.nwt = encode_wave_tiles(n);
    w.nblk = (w.nwt + SCAN_BLOCK - 1) / SCAN_BLOCK;
    auto up = [](size_t v) { return (v + 63) & ~size_t(63); };
    w.off_bits = 64;
    w.off_start = up(w.off_bits + size_t(w.nwt) * 4);
    w.off_blk = up(w.off_start + size_t(w.nwt) * 8);
    w.total = up(w.off_blk + size_t(w.nblk + 1) * 8);
    if (w.total < 64 + 2 * 1024 * 8 + 1024 * 4) w.total = 64 + 2 * 1024 * 8 + 1024 * 4;   // the region path keeps <= 1024 lengths, starts and escape flags here
    return w;
}
size_t encode_workspace_bytes(uint64_t n) { return enc_ws_layout(n).total; }

__global__ void empty_payload_kernel(const unsigned long long *start_bit, unsigned long long *nbits, uint8_t *out, uint64_t cap) {
    const unsigned long long b0 = start_bit ? (*start_bit & 7ull) : 0;
    *nbits = b0;
    if (b0 && cap) out[0] = 0;
}

// sum over the histogram of count x code length = the payload bits this model produces for data with
// that histogram (a shard's LOCAL histogram: its start offset is known before it is encoded)
__global__ __launch_bounds__(256) void payload_bits_kernel(const unsigned long long *counts, const uint8_t *len8, uint32_t entries,
                                                           unsigned long long *out) {
    __shared__ unsigned long long part[256];
    unsigned long long acc = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < entries; i += gridDim.x * 256u) acc += counts[i] * len8[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (int(threadIdx.x) < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0 && part[0]) atomicAdd(out, part[0]);
}

hipError_t launch_payload_bits(const unsigned long long *d_counts, const uint8_t *d_len8, uint32_t entries, unsigned long long *d_out,
                               hipStream_t st) {
    hipError_t e = hipMemsetAsync(d_out, 0, 8, st);
    if (e != hipSuccess) return e;
    const unsigned grid = entries > 65536u ? 1024u : 1u;
    hipLaunchKernelGGL(payload_bits_kernel, dim3(grid), dim3(256), 0, st, d_counts, d_len8, entries, d_out);
    return hipGetLastError();
}

// MH_ENCODE2_PATH=two_pass: order 2 through the length pass + emit pair also when the hot image is there (A/B runs, tests)
static bool encode2_two_pass() {
    const char *v = getenv("MH_ENCODE2_PATH");
    return v && !strcmp(v, "two_pass");
}

hipError_t launch_encode(const EncodeArgs &a, void *d_ws, hipStream_t st) {
    unsigned char *ws = static_cast<unsigned char *>(d_ws);
    const EncWs L = enc_ws_layout(a.n);
    int *status = reinterpret_cast<int *>(ws);
    hipError_t e = hipMemsetAsync(ws, 0, 64, st);
    if (e != hipSuccess) return e;
    if (a.n == 0) {                                          // nothing to emit: the payload "ends" at its start offset
        hipLaunchKernelGGL(empty_payload_kernel, dim3(1), dim3(1), 0, st, a.start_bit, a.nbits, a.out, a.cap);
        return hipGetLastError();
    }
    e = once_per_device(&DeviceState::encode_ready, [] {
        hipError_t r = allow_lds(reinterpret_cast<const void *>(enc_len_kernel<1>), LEN_LDS_BYTES);
        if (r == hipSuccess) r = allow_lds(reinterpret_cast<const void *>(enc_len_kernel<2>), LEN_LDS_BYTES);
        if (r == hipSuccess) r = allow_lds(reinterpret_cast<const void *>(enc_emit_kernel<2>), EMIT_LDS_BYTES);
        return r != hipSuccess ? r : allow_lds(reinterpret_cast<const void *>(enc_emit_kernel<1>), EMIT_LDS_BYTES);
    });
    if (e != hipSuccess) return e;
    // which encoder ran (status block bytes 8..11, mh_dev_encode_path): ENC_PATH_LENGTH_PASS
    (void)launch_set_word(reinterpret_cast<uint32_t *>(ws + 8), uint32_t(ENC_PATH_LENGTH_PASS), st);
    uint32_t *wt_bits = reinterpret_cast<uint32_t *>(ws + L.off_bits);
    unsigned long long *wt_start = reinterpret_cast<unsigned long long *>(ws + L.off_start);
    unsigned long long *blk_sum = reinterpret_cast<unsigned long long *>(ws + L.off_blk);

    uint64_t want = (L.nwt + E_WAVES - 1) / E_WAVES;
    int grid = int(want > uint64_t(2 * cu_count()) ? uint64_t(2 * cu_count()) : want);
    const bool hot2 = a.order == 2 && a.o2hot && a.o2hot_bytes && a.o2hot_bytes <= uint32_t(LEN_LDS_BYTES);
    if (hot2 && !a.no_chain && !encode2_two_pass()) {                       // one pass: enc_chain_kernel
        e = once_per_device(&DeviceState::chain_ready, [] { return allow_lds(reinterpret_cast<const void *>(enc_chain_kernel), EMIT_LDS_BYTES); });
        if (e != hipSuccess) return e;
        const uint64_t groups = (L.nwt + E_WAVES - 1) / E_WAVES;
        e = hipMemsetAsync(wt_start, 0, size_t(groups) * 8, st);            // the groups' state words
        if (e != hipSuccess) return e;
        (void)launch_set_word(reinterpret_cast<uint32_t *>(ws + 8), uint32_t(ENC_PATH_CHAIN), st);
        ChainParams cp;
        cp.e = EmitParams{a.data, a.n, a.prev0, a.chunk_shift, a.out, a.enc16, a.len8, a.code64, a.enc64, nullptr, L.nwt, a.index, status,
                          a.fine, a.o2hot, a.o2hot_bytes};
        cp.state = wt_start;
        cp.start_bit = a.start_bit;
        cp.cap = a.cap & ~uint64_t(3);                                       // whole dwords are stored
        cp.nbits = a.nbits;
        cp.status = status;
        // test hooks (they never change a byte of a stream: the encoder reports MHK_STATUS_TIMEOUT and the caller retries with the
        // two-pass pair): "timeout" = as if the leader's look-back had run out, "timeout_follower" = as if a follower's wait had
        const char *probe = getenv("MH_CHAIN_PROBE");
        cp.probe = !probe ? 0u : !strcmp(probe, "timeout") ? 2u : !strcmp(probe, "timeout_follower") ? 3u : 0u;
#ifdef MH_EXP_PROBES
        if (probe && !strcmp(probe, "nolookback")) cp.probe = 1u;             // diagnostic builds only: output wrong
#endif
        cp.sync = reinterpret_cast<uint32_t *>(ws + 32);                     // (zeroed with the status block above)
        const int cgrid = int(groups > uint64_t(cu_count()) ? uint64_t(cu_count()) : groups);
        hipLaunchKernelGGL(enc_chain_kernel, dim3(cgrid), dim3(E_THREADS), ((a.o2hot_bytes + 15u) & ~15u) + E_WAVES * E_STAGE_WORDS * 4, st, cp);
        return hipGetLastError();
    }
    if (hot2) {                                              // the live contexts' tables in LDS (o2hot_lookup16)
        LenParams lp{a.data, a.n, a.prev0, a.len8, wt_bits, L.nwt, a.o2hot, a.o2hot_bytes};
        hipLaunchKernelGGL(enc_len_kernel<2>, dim3(grid), dim3(E_THREADS), (a.o2hot_bytes + 15u) & ~15u, st, lp);
    } else if (a.order == 2) {                               // lengths gathered from the full table (a.len8)
        LenParams lp{a.data, a.n, a.prev0, a.len8, wt_bits, L.nwt, nullptr, 0};
        hipLaunchKernelGGL(enc2_len_kernel, dim3(grid), dim3(E_THREADS), 0, st, lp);
    } else {
        LenParams lp{a.data, a.n, a.prev0, a.len_slot, wt_bits, L.nwt, nullptr, 0};
        hipLaunchKernelGGL(enc_len_kernel<1>, dim3(grid), dim3(E_THREADS), LEN_LDS_BYTES, st, lp);
    }

    hipLaunchKernelGGL(scan_local_kernel, dim3(unsigned(L.nblk)), dim3(SCAN_THREADS), 0, st, wt_bits, L.nwt, wt_start, blk_sum);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, blk_sum, L.nblk, a.start_bit);
    // the emit pass stores whole dwords: only the 4-byte-aligned part of the buffer counts as capacity
    ScanParams sp{wt_start, blk_sum, L.nwt, L.nblk, a.out, a.cap & ~uint64_t(3), a.nbits, status};
    hipLaunchKernelGGL(scan_apply_kernel, dim3(unsigned(L.nblk)), dim3(SCAN_THREADS), 0, st, sp);

    EmitParams ep{a.data, a.n, a.prev0, a.chunk_shift, a.out, a.enc16, a.len8, a.code64, a.enc64, wt_start, L.nwt, a.index, status,
                  a.fine, a.o2hot, a.o2hot_bytes};
    if (hot2) {
        grid = int(want > uint64_t(cu_count()) ? uint64_t(cu_count()) : want);
        hipLaunchKernelGGL(enc_emit_kernel<2>, dim3(grid), dim3(E_THREADS), ((a.o2hot_bytes + 15u) & ~15u) + E_WAVES * E_STAGE_WORDS * 4, st, ep);
        return hipGetLastError();
    }
    if (a.order == 2) {                                      // no table in LDS: two workgroups per CU
        hipLaunchKernelGGL(enc2_emit_kernel, dim3(grid), dim3(E_THREADS), E_WAVES * E_STAGE_WORDS * 4, st, ep);
        return hipGetLastError();
    }
    grid = int(want > uint64_t(cu_count()) ? uint64_t(cu_count()) : want);
    hipLaunchKernelGGL(enc_emit_kernel<1>, dim3(grid), dim3(E_THREADS), EMIT_LDS_BYTES, st, ep);
    return hipGetLastError();
}

// Encode after a region-mode histogram of the same input (d_hist_ws as launch_hist_o1 left it): no length pass.
// Order 1/0 models without escape codes only (the caller checks); hipErrorInvalidValue when the workspace
// cannot be a region histogram of n bytes.  A workspace that holds another input's histogram is caught on
// the device (status MHK_STATUS_CORRUPT, nothing written).
hipError_t launch_encode_regions(const EncodeArgs &a, const void *d_hist_ws, size_t hist_ws_bytes, void *d_ws, hipStream_t st) {
    unsigned char *ws = static_cast<unsigned char *>(d_ws);
    int *status = reinterpret_cast<int *>(ws);
    hipError_t e = hipMemsetAsync(ws, 0, 64, st);
    if (e != hipSuccess) return e;
    if (a.n == 0) {
        hipLaunchKernelGGL(empty_payload_kernel, dim3(1), dim3(1), 0, st, a.start_bit, a.nbits, a.out, a.cap);
        return hipGetLastError();
    }
    const RegionGeom g = region_geom(a.n);
    if (!d_hist_ws || hist_ws_bytes < g.total || g.grid > 1024) return hipErrorInvalidValue;
    e = once_per_device(&DeviceState::region_ready, [] {
        hipError_t r = allow_lds(reinterpret_cast<const void *>(enc_region_kernel<false>), REGION_LDS_BYTES);
        if (r == hipSuccess) r = allow_lds(reinterpret_cast<const void *>(enc_region_kernel<false, true>), REGION_LDS_BYTES);
        if (r == hipSuccess) r = allow_lds(reinterpret_cast<const void *>(region_bits_kernel), 65536);
        return r != hipSuccess ? r : allow_lds(reinterpret_cast<const void *>(enc_region_kernel<true>), REGION_LDS_BYTES);
    });
    if (e != hipSuccess) return e;
    const unsigned char *hws = static_cast<const unsigned char *>(d_hist_ws);
    unsigned long long *region_bits = reinterpret_cast<unsigned long long *>(ws + 64);
    unsigned long long *region_start = region_bits + 1024;
    uint32_t *region_esc = reinterpret_cast<uint32_t *>(region_start + 1024);
    const HistHeader expect{HIST_WS_MAGIC, a.n, reinterpret_cast<unsigned long long>(a.data), g.region_vecs, uint32_t(g.grid), a.prev0, g.cross_cap, 0};
    hipLaunchKernelGGL(region_bits_kernel, dim3(g.grid), dim3(1024), 65536, st, reinterpret_cast<const HistHeader *>(hws + 64), expect,
                       reinterpret_cast<const uint32_t *>(hws + g.off_slab), reinterpret_cast<const uint32_t *>(hws + g.off_cross), a.len8,
                       region_bits, region_esc, status);
    hipLaunchKernelGGL(region_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, region_bits, uint32_t(g.grid), region_start, a.start_bit,
                       a.out, a.cap & ~uint64_t(3), a.nbits, status);
    EmitParams ep{a.data, a.n, a.prev0, a.chunk_shift, a.out, a.enc16, a.len8, a.code64, nullptr, nullptr, 0, a.index, status, a.fine, nullptr, 0};
    RegionParams rp{region_start, region_bits, region_esc, g.region_vecs, g.nvec_up, (a.cap & ~uint64_t(3)) >> 2, status};
    (void)launch_set_word(reinterpret_cast<uint32_t *>(ws + 8), uint32_t(a.max_len > mh::ENC16_MAX_LEN ? ENC_PATH_REGIONS_ESCAPES : ENC_PATH_REGIONS), st);
    hipLaunchKernelGGL(enc_region_kernel<false>, dim3(g.grid), dim3(E_THREADS), REGION_LDS_BYTES, st, ep, rp);
    if (a.max_len >= 8)                          // a region of ~8 bits per symbol needs codes that long: its rounds run with fewer lanes [r5]
        hipLaunchKernelGGL((enc_region_kernel<false, true>), dim3(g.grid), dim3(E_THREADS), REGION_LDS_BYTES, st, ep, rp);
    if (a.max_len > mh::ENC16_MAX_LEN)           // the model has codes over 12 bits: the regions that contain any
        hipLaunchKernelGGL(enc_region_kernel<true>, dim3(g.grid), dim3(E_THREADS), REGION_LDS_BYTES, st, ep, rp);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void enc64_pack_kernel(const uint8_t *len8, const unsigned long long *code64, unsigned long long *enc64, uint64_t n) {
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
        const uint32_t l = len8[i];
        enc64[i] = l <= 56u ? ((unsigned long long)(l) << 56) | code64[i] : 0xFF00000000000000ull;
    }
}
hipError_t launch_enc64_pack(const uint8_t *len8, const uint64_t *code64, uint64_t *enc64, uint64_t n, hipStream_t st) {
    hipLaunchKernelGGL(enc64_pack_kernel, dim3(unsigned(cu_count()) * 8u), dim3(256), 0, st, len8,
                       reinterpret_cast<const unsigned long long *>(code64), reinterpret_cast<unsigned long long *>(enc64), n);
    return hipGetLastError();
}

// The redo pass: one lane per chunk listed in p.redo (count in [0]), runtime table widths, with the tree walk for
// codes longer than both table levels.  Normally the list is empty and the launch returns at once.

hipError_t launch_scan_local(const uint32_t *d_vals, uint64_t n, unsigned long long *d_start, unsigned long long *d_blk_sum, hipStream_t st) {
    const uint64_t nblk = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(scan_local_kernel, dim3(unsigned(nblk)), dim3(SCAN_THREADS), 0, st, d_vals, n, d_start, d_blk_sum);
    return hipGetLastError();
}
hipError_t launch_scan_top(unsigned long long *d_blk_sum, uint64_t nblk, const unsigned long long *d_carry0, hipStream_t st) {
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, d_blk_sum, nblk, d_carry0);
    return hipGetLastError();
}

}  // namespace mhk
