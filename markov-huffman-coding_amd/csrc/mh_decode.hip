// mh_decode.hip — the chunk decoder of the Markov-Huffman hot path for gfx950 (the tile decoder is
// mh_tile.hip).
//   decode_kernel      two-level decode tables (level 1 in LDS = the reference's 8-bit LUT), K chunks per lane, granule FIFO
//                      input, redo pass with the tree walk
//   decode2_kernel     order 2: one lane per chunk, every table level gathered (extension, parity unpinned)
#include "mh_decode_dev.hpp"

namespace mhk {

// Walk for codes longer than P + h (rare).  `skip` = P + h bits of the window have NOT been consumed.
// Returns false on a corrupt stream.
template <typename LS>
__device__ __forceinline__ bool walk_long(const DecTables &t, LS &ls, uint32_t prev, uint32_t e2, uint32_t skip,
                                          uint32_t &sym) {
    ls.buf <<= skip; ls.cnt -= skip;
    uint32_t node = e2 & 0x1FFu;
    const uint32_t *tr = t.tree + prev * TREE_STRIDE;
    for (int guard = 0; guard < 256; ++guard) {
        ls.template refill<true>();
        uint32_t bit = uint32_t(ls.buf >> 63);
        ls.buf <<= 1; ls.cnt -= 1u;
        uint32_t pair = tr[node];
        uint32_t c = bit ? (pair >> 16) : (pair & 0xFFFFu);
        if (c & TREE_LEAF) { sym = c & 255u; ls.template refill<true>(); return true; }
        node = c;
    }
    sym = 0;
    return false;
}

// One symbol from each of the lane's K independent streams.  prim and sec_base live in LDS; sec lives
// in LDS too whenever the model's tables fit (t.sec then points into LDS).  The second-level step is
// skipped by the whole wave when no lane needs it.
// REFILL 1: top the bit windows up first: >= 33 bits unless the window was empty (then 32).  A
// table-resolved code is at most P + h <= 16 bits, so one refill covers two symbols (two 16-bit codes leave
// one bit), four when the model has no code longer than 8 bits.  (Three symbols per refill for codes of at
// most 11 bits — with a second refill for the window that was empty — measured no faster: the kernel is
// not bound by its instruction count.)
// A null table entry consumes nothing; the caller detects it because the chunk then ends at the wrong
// bit offset.
// pe[k] is the entry that resolved the stream's previous symbol: only its low byte (the symbol) is
// defined.  PC / HC: P and H when they are known at compile time (8), 0 = read them from `t`.  With
// both widths at 8 bits the table indices are byte shuffles (one v_perm each).
#ifndef MH_DEC_LAG
#define MH_DEC_LAG 1              // streams between a first-level lookup and its second-level gather (A/B builds)
#endif
template <int PC>
__device__ __forceinline__ uint32_t prim_index(uint32_t pe, uint32_t hi, uint32_t P) {
    if (PC == 8) return __builtin_amdgcn_perm(pe, hi, 0x0C0C0403u);             // sym << 8 | hi >> 24
    return ((pe & 255u) << P) + __builtin_amdgcn_ubfe(hi, 32u - P, P);
}
// inserts the low byte of `e` as byte j of `d` (j is a constant after unrolling)
__device__ __forceinline__ uint32_t put_byte(uint32_t d, uint32_t e, int j) {
    const uint32_t sel = j == 0 ? 0x03020104u : j == 1 ? 0x03020400u : j == 2 ? 0x03040100u : 0x04020100u;
    return __builtin_amdgcn_perm(e, d, sel);
}

// WALK: codes longer than both table levels are walked in place.  The K-stream hot loop runs without
// it (K inlined copies of the walk cost 13 % of the decode time in registers and code): such a stream
// sets its bit in `redo`, and the kernel hands the chunk to the redo pass.
// Hot loop of the L2 (direct) layout, in two halves so that the caller can put its own global loads and
// stores BETWEEN them: a wave's vector-memory results come back in order, so a granule load or a store
// burst issued before the step's gathers would stand between the wave and their results for a whole trip
// to HBM; issued behind them it has until the next step's gathers.
//   issue:   refill, first-level lookup, second-level reads on their way
//   consume: the entry that resolves each symbol, window shift
// EVERY lane gathers, with no exec masking and no test whether anyone needs to.  A leaf entry carries bit
// 15, so its index is >= 0x8000 << H, past the end of the table (at most 32767 << H entries): the buffer
// bounds check answers such a lane with 0 and sends nothing to the cache.  Unresolved codes are only
// accumulated (leafacc, shared by the lane's K streams, loses bit 15); the caller looks at it once per chunk
// group.  Until then such a stream decodes garbage: every table index stays in range or bounds-checked,
// the input FIFO clamps its granule index.
template <int K, int REFILL, int PC, int HC, typename LS>
__device__ __forceinline__ void direct_issue(const uint16_t *prim, const DecTables &t, LS (&ls)[K], const uint32_t (&pe)[K],
                                             uint32_t (&e)[K], uint32_t (&e2)[K]) {
    const uint32_t P = PC ? uint32_t(PC) : t.P;
    const uint32_t H = HC ? uint32_t(HC) : t.H;
    uint32_t hi[K];
    // a stream's first-level lookup leaves right behind its own refill, and its second-level gather MH_DEC_LAG
    // streams later, as soon as that lookup is back: both latencies then run under the refills of the streams
    // after it (a wave issues in order, and the refills are branches the compiler does not move loads across)
    auto gather = [&](int k) __attribute__((always_inline)) {
        uint32_t idx;
        if (PC == 8 && HC == 8) idx = __builtin_amdgcn_perm(e[k], hi[k], 0x0C050402u);   // e << 8 | byte 2 of hi
        else idx = (e[k] << H) | __builtin_amdgcn_ubfe(hi[k], 32u - P - H, H);
        e2[k] = uint32_t(uint16_t(__builtin_amdgcn_raw_buffer_load_b16(t.sec_rsrc, int(idx << 1), 0, 0)));
    };
    constexpr int LAG = MH_DEC_LAG < K ? MH_DEC_LAG : K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (REFILL) ls[k].template refill<false>();
        hi[k] = uint32_t(ls[k].buf >> 32);
        e[k] = prim[prim_index<PC>(pe[k], hi[k], P)];
        if (k >= LAG) gather(k - LAG);
    }
#pragma unroll
    for (int k = K - LAG; k < K; ++k) gather(k);
}
template <int K, typename LS>
__device__ __forceinline__ void direct_consume(LS (&ls)[K], uint32_t (&pe)[K], const uint32_t (&e)[K], const uint32_t (&e2)[K],
                                               uint32_t &leafacc) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t ef = e[k] > e2[k] ? e[k] : e2[k];
        leafacc &= ef;
        const uint32_t len = __builtin_amdgcn_ubfe(ef, 8, 5);
        ls[k].buf <<= len;
        ls[k].cnt -= len;
        pe[k] = ef;
    }
}

template <int K, bool CHECKED, int REFILL, bool DIRECT, int PC, int HC, bool WALK, typename LS>
__device__ __forceinline__ void decode_step(const uint16_t *prim, const uint32_t *sec_base, const DecTables &t,
                                            LS (&ls)[K], uint32_t (&pe)[K], bool &bad, uint32_t &redo, uint32_t &leafacc) {
    const uint32_t P = PC ? uint32_t(PC) : t.P;
    const uint32_t H = HC ? uint32_t(HC) : t.H;
    uint32_t hi[K], e[K], sb[K], ef[K];
    // a stream's first-level lookup leaves right behind its own refill: its LDS latency then runs under the
    // refills of the streams after it (a wave issues in order, and the refills are branches the compiler
    // does not move loads across)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (REFILL) ls[k].template refill<CHECKED>();
        hi[k] = uint32_t(ls[k].buf >> 32);
        e[k] = prim[prim_index<PC>(pe[k], hi[k], P)];
        sb[k] = DIRECT ? 0u : sec_base[pe[k] & 255u];           // independent of e[k]: same latency
    }
    uint32_t all = DEC16_LEAF;
#pragma unroll
    for (int k = 0; k < K; ++k) { ef[k] = e[k]; all &= e[k]; }
    if (__any(all == 0)) {                                      // wave-uniform: some stream hit an inner entry
        uint32_t e2[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool in = (e[k] & DEC16_LEAF) == 0;
            const uint32_t h = DIRECT ? H : ((e[k] >> 12) & 7u) + 1u;
            // only the lanes that need it take part in the gather: every extra quad of lanes costs the
            // vector L1 a tag lookup even when it reads a dummy address.  The direct (L2) layout gathers
            // through a buffer resource: 32-bit offsets, no 64-bit address per lane
            e2[k] = 0;
            uint32_t idx;
            if (DIRECT && PC == 8 && HC == 8) idx = __builtin_amdgcn_perm(e[k], hi[k], 0x0C050402u);   // e << 8 | byte 2 of hi
            else if (DIRECT) idx = (e[k] << H) | __builtin_amdgcn_ubfe(hi[k], 32u - P - H, H);
            else idx = sb[k] + (e[k] & 0xFFFu) + __builtin_amdgcn_ubfe(hi[k], 32u - P - h, h);
            if (in) e2[k] = DIRECT ? uint32_t(uint16_t(__builtin_amdgcn_raw_buffer_load_b16(t.sec_rsrc, int(idx << 1), 0, 0))) : uint32_t(t.sec[idx]);
        }
        // leaves carry bit 15 and lanes without a second level hold 0: the larger one is the entry that
        // resolves the symbol; if both are inner the result has no leaf flag and the code is walked
        uint32_t all2 = DEC16_LEAF;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            ef[k] = e[k] > e2[k] ? e[k] : e2[k];
            all2 &= ef[k];
        }
        if (__any(all2 == 0)) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (!(ef[k] & DEC16_LEAF)) {
                    if (WALK) {
                        const uint32_t h = DIRECT ? H : ((e[k] >> 12) & 7u) + 1u;
                        uint32_t s = 0;
                        if (!walk_long(t, ls[k], pe[k] & 255u, e2[k], P + h, s)) bad = true;
                        ef[k] = DEC16_LEAF | s;                  // length 0: already consumed
                    } else {
                        redo |= 1u << k;                         // the rest of this chunk decodes to nothing
                        ef[k] = DEC16_LEAF;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t len = __builtin_amdgcn_ubfe(ef[k], 8, 5);   // a null entry consumes nothing: the chunk then ends at the wrong bit
        ls[k].buf <<= len;
        ls[k].cnt -= len;
        pe[k] = ef[k];
    }
}

constexpr int DEC_THREADS = 512;
constexpr int DEC_LDS_MAX = 163840;

// Decodes `nsym` symbols of ONE chunk that must end at bit `end_bits` (tail groups and the ragged
// last chunk).
template <bool DIRECT>
__device__ __forceinline__ void decode_chunk_single(const uint16_t *lut, const uint32_t *sub_base, const DecTables &t,
                                                    const uint8_t *payload, uint64_t total_bytes, uint64_t nbits,
                                                    uint64_t entry, uint64_t end_bits, uint8_t *o, uint32_t nsym, int *status) {
    const uint64_t bitpos = entry & 0x00FFFFFFFFFFFFFFull;
    if (bitpos >= nbits) { atomicExch(status, MHK_STATUS_CORRUPT); return; }
    LaneStream<8> ls[1];
    uint32_t prev[1] = {uint32_t(entry >> 56)};
    ls[0].init(payload, total_bytes, bitpos);
    bool bad = false;
    uint32_t q = 0, redo = 0;
    uint32_t leafacc = DEC16_LEAF;                               // only the hot loop defers the check
    for (uint32_t i = 0; i < nsym; ++i) {
        decode_step<1, true, 1, DIRECT, 0, 0, true>(lut, sub_base, t, ls, prev, bad, redo, leafacc);
        q |= (prev[0] & 255u) << (8u * (i & 3u));
        if ((i & 3u) == 3u) { *reinterpret_cast<uint32_t *>(o + i - 3u) = q; q = 0; }
    }
    for (uint32_t i = nsym & ~3u; i < nsym; ++i) o[i] = uint8_t(q >> (8u * (i & 3u)));
    if (bad || ls[0].position() != uint32_t(end_bits)) atomicExch(status, MHK_STATUS_CORRUPT);
}

// one burst of OUTB 16-byte pieces per stream: stream k of the lane decodes chunk c0 + k * NT
template <int K, int OUTB, int NT>
__device__ __forceinline__ void store_burst(const DecParams &p, uint64_t c0, const uint32_t (&Q)[K][OUTB][4], uint32_t b) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        uint4 *o16 = reinterpret_cast<uint4 *>(p.out + ((c0 + uint64_t(k) * NT) << p.chunk_shift)) + b * OUTB;
#pragma unroll
        for (int u = 0; u < OUTB; ++u) o16[u] = make_uint4(Q[k][u][0], Q[k][u][1], Q[k][u][2], Q[k][u][3]);
    }
}

// ---- the chunk decoder's variants [r5]: every instantiation the launcher can select, by name (mh_kernels.h: DecVariant; the
// one that ran is on record in the workspace, mh_dev_decode_variant; tests/test_gpu_decode_variants.py drives each of them
// against the oracle).  Rounds 1-3 grew this kernel eleven template parameters and A/B macros; what survived the measurements:
//   sec_lds  both table levels in LDS (else: uniform second-level tables of 2^H entries gathered from L2, "direct" layout)
//   spr      symbols per window refill (2, or 4 when no code exceeds 8 bits)
//   k        independent streams (chunks) per lane
//   gw       dwords per input granule (8 = 32 B, 16 = 64 B)
//   outb     16-byte stores per output burst
//   pc / hc  first-level width / second-level height as compile-time constants (0: read from the parameters)
//   redo     the redo pass: one lane per listed chunk, tree walk for codes longer than both levels
//   depth    slots of the input granule FIFO
// Models whose tables live in LDS and whose codes are all <= 8 bits are bound by how the streams touch HBM (32-byte granules
// re-fetch every 128-byte line four times, 16-byte stores double the write traffic): they run two streams with 64-byte
// granules and bursts when the payload is a large part of the traffic (uniform bytes: 1.6x), four light streams otherwise
// (41 %-ratio text: the wide form is 8 % slower).  With a second level the dependent lookups dominate and four streams win
// (Zipf code lengths, all tables in LDS: 5.5 ms light, 7.7 ms wide per 4 GiB).  The general L2 layout of rounds 1-2 (indexed
// second-level tables gathered from L2) is gone: it was only chosen for more than 32 767 depth-8 inner nodes, and 256
// contexts of at most 256 leaves have at most 256 x 127 of them (a context with 128 would hold 256 leaves below depth 8: a
// Kraft sum of at most 1/2).
struct DecCfg { bool sec_lds; int spr; int k, gw, outb, pc, hc; bool redo; int depth; };
__host__ __device__ constexpr DecCfg dec_cfg(int v) {
    return v == DV_LDS_WIDE       ? DecCfg{true, 4, 2, 16, 4, 8, 0, false, 2}
         : v == DV_LDS_SHORT      ? DecCfg{true, 4, 4, 8, 2, 8, 0, false, 2}
         : v == DV_LDS_TWO_LEVEL  ? DecCfg{true, 2, 4, 8, 2, 0, 0, false, 2}
         : v == DV_LDS_TWO_LEVEL_P8 ? DecCfg{true, 2, 4, 8, 2, 8, 0, false, 2}
         : v == DV_L2_DIRECT      ? DecCfg{false, 2, 4, 8, 4, 8, 0, false, 1}
         : v == DV_L2_DIRECT_H2   ? DecCfg{false, 2, 4, 8, 4, 8, 2, false, 1}
         : v == DV_L2_DIRECT_H3   ? DecCfg{false, 2, 4, 8, 4, 8, 3, false, 1}
         : v == DV_L2_DIRECT_H4   ? DecCfg{false, 2, 4, 8, 4, 8, 4, false, 1}
         : v == DV_L2_DIRECT_H8   ? DecCfg{false, 2, 4, 8, 4, 8, 8, false, 1}
         : v == DV_REDO_LDS       ? DecCfg{true, 2, 1, 8, 1, 0, 0, true, 2}
                                  : DecCfg{false, 2, 1, 8, 1, 0, 0, true, 2};      // DV_REDO_L2_DIRECT
}
constexpr int DEC_NT = 512;                                       // lanes per workgroup, every variant
template <int V>
__global__ __launch_bounds__(DEC_NT) void decode_kernel(DecParams p) {
    constexpr DecCfg C = dec_cfg(V);
    constexpr bool SEC_LDS = C.sec_lds, DIRECT = !C.sec_lds, REDO = C.redo;
    constexpr int SPR = C.spr, K = C.k, GW = C.gw, OUTB = C.outb, PC = C.pc, HC = C.hc, NT = DEC_NT, DEPTH = C.depth;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (REDO && p.redo[0] == 0) return;                         // the usual case: nothing was handed over
    if (p.d_nbits) {                                            // payload length still on the device (mh_dev_decode_dn)
        p.nbits = *p.d_nbits;
        p.payload_bytes = (p.nbits + 7) >> 3;
    }
    // LDS: sec_base u32[256] | prim u16[256 << P] | sec u16[nsec] (only when the model's tables fit)
    uint32_t *sub_base = reinterpret_cast<uint32_t *>(smem);
    uint16_t *lut = reinterpret_cast<uint16_t *>(smem + 1024);
    const uint32_t nprim16 = (256u << p.P) / 8u;                // uint4 units
    for (uint32_t i = threadIdx.x; i < nprim16; i += NT)
        reinterpret_cast<uint4 *>(lut)[i] = reinterpret_cast<const uint4 *>(p.prim)[i];
    uint16_t *lsec = lut + (256u << p.P);
    if (SEC_LDS) {
        const uint32_t nsec16 = (p.nsec + 7u) / 8u;             // the buffer is padded to 16 bytes
        for (uint32_t i = threadIdx.x; i < nsec16; i += NT)
            reinterpret_cast<uint4 *>(lsec)[i] = reinterpret_cast<const uint4 *>(p.sec)[i];
    }
    if (threadIdx.x < 256) sub_base[threadIdx.x] = p.sec_base[threadIdx.x];
    __syncthreads();

    // (a hybrid — tables of the frequent contexts in LDS, the rest in L2 — was measured in both rounds and did
    //  not pay, with the masked and with the bounds-checked gather: 25.42 vs 25.42 ms)
    const DecTables tabs{SEC_LDS ? lsec : p.sec, p.tree, p.P, p.direct, p.H,
                         __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.sec), 0, int((p.nsec + 8u) * 2u), 0x00020000)};
    const uint32_t S = 1u << p.chunk_shift;
    const uint64_t full_chunks = p.n >> p.chunk_shift;          // chunks with exactly S symbols
    if (REDO) {
        // ---- redo pass: the chunks the hot loop gave up on (a code longer than both table levels),
        // one lane per chunk, with the walk
        const uint32_t count = p.redo[0];
        for (uint32_t i = blockIdx.x * NT + threadIdx.x; i < count; i += gridDim.x * NT) {
            const uint64_t c = p.redo[1 + i];
            const uint64_t first = c << p.chunk_shift;
            const uint32_t nsym = (p.n - first) >= S ? S : uint32_t(p.n - first);
            const uint64_t endb = (c + 1 < p.nchunks) ? (p.index[c + 1] & 0x00FFFFFFFFFFFFFFull) : p.nbits;
            decode_chunk_single<DIRECT>(lut, sub_base, tabs, p.payload, p.payload_bytes, p.nbits, p.index[c], endb, p.out + first, nsym, p.status);
        }
        return;
    }
    const uint64_t group = uint64_t(NT) * K;           // chunks per workgroup iteration
    constexpr int BLK16 = (2 * GW) / 16;                        // 16-symbol groups per refill block
    for (uint64_t g0 = uint64_t(blockIdx.x) * group; g0 < p.nchunks; g0 += uint64_t(gridDim.x) * group) {
        const uint64_t c0 = g0 + threadIdx.x;                   // stream k -> chunk c0 + k * NT
        if (c0 + uint64_t(K - 1) * NT < full_chunks) {
            // ---- K full chunks: interleaved decode
            LaneStream<GW, DEPTH> ls[K];
            uint32_t prev[K];
            bool ok = true;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint64_t c = c0 + uint64_t(k) * NT;
                const uint64_t entry = p.index[c];
                const uint64_t bitpos = entry & 0x00FFFFFFFFFFFFFFull;
                const uint64_t endpos = (c + 1 < p.nchunks) ? (p.index[c + 1] & 0x00FFFFFFFFFFFFFFull) : p.nbits;
                prev[k] = uint32_t(entry >> 56);
                const bool fine = bitpos < p.nbits && endpos >= bitpos && endpos - bitpos <= (uint64_t(S) << 6);
                ok = ok && fine;
                if (fine) ls[k].init(p.payload, p.payload_bytes, bitpos);
            }
            if (!ok) { atomicExch(p.status, MHK_STATUS_CORRUPT); continue; }
            bool bad = false;
            uint32_t redo = 0;                                   // bit k: stream k met a code the tables do not resolve
            uint32_t leafacc = DEC16_LEAF;
            // Q[k] = the stream's burst of OUTB 16-byte pieces; the pieces rotate through it so that the
            // 16-symbol body below writes a fixed set of registers (the u loop stays rolled: code size)
            uint32_t Q[K][OUTB][4] = {};
            constexpr bool SPLIT = DIRECT && !SEC_LDS;           // table gathers from L2 in every step
            const uint32_t nburst = (S >> 4) / OUTB;
            for (uint32_t burst = 0; burst < nburst; ++burst) {
#pragma unroll 1
                for (int u = 0; u < OUTB; ++u) {                 // 16 symbols -> one uint4 per stream
                    uint32_t d[K];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if ((j & 3) == 0) {
#pragma unroll
                            for (int k = 0; k < K; ++k) d[k] = 0;
                        }
                        uint32_t e[K], e2[K];
                        if (SPLIT) {
                            if (j % SPR == 0) direct_issue<K, 1, PC, HC>(lut, tabs, ls, prev, e, e2);
                            else direct_issue<K, 0, PC, HC>(lut, tabs, ls, prev, e, e2);
                        }
                        if (j == 0) {
                            // the first step of the 16 carries the traffic: the previous burst's stores and the
                            // streams' next input granules leave behind its table gathers (see direct_issue)
                            if (u == 0 && burst != 0) store_burst<K, OUTB, NT>(p, c0, Q, burst - 1);
                            if (u % BLK16 == 0) {
#pragma unroll
                                for (int k = 0; k < K; ++k) ls[k].block_sync();
                            }
#pragma unroll
                            for (int k = 0; k < K; ++k)
#pragma unroll
                                for (int t = 0; t + 1 < OUTB; ++t) { Q[k][t][0] = Q[k][t + 1][0]; Q[k][t][1] = Q[k][t + 1][1]; Q[k][t][2] = Q[k][t + 1][2]; Q[k][t][3] = Q[k][t + 1][3]; }
                        }
                        if (SPLIT) direct_consume<K>(ls, prev, e, e2, leafacc);
                        else if (j % SPR == 0) decode_step<K, false, 1, DIRECT, PC, HC, false>(lut, sub_base, tabs, ls, prev, bad, redo, leafacc);
                        else decode_step<K, false, 0, DIRECT, PC, HC, false>(lut, sub_base, tabs, ls, prev, bad, redo, leafacc);
#pragma unroll
                        for (int k = 0; k < K; ++k) d[k] = put_byte(d[k], prev[k], j & 3);
                        if ((j & 3) == 3) {
#pragma unroll
                            for (int k = 0; k < K; ++k) Q[k][OUTB - 1][j >> 2] = d[k];
                        }
                    }
                }
            }
            store_burst<K, OUTB, NT>(p, c0, Q, nburst - 1);
            if (!(leafacc & DEC16_LEAF)) redo = (1u << K) - 1u;  // some stream of this lane: all K chunks go to the redo pass
            // every chunk must end exactly where the next one starts (null entries, a wrong table or a
            // damaged stream all miss it)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint64_t c = c0 + uint64_t(k) * NT;
                const uint64_t endpos = (c + 1 < p.nchunks) ? (p.index[c + 1] & 0x00FFFFFFFFFFFFFFull) : p.nbits;
                if (redo & (1u << k)) p.redo[1u + atomicAdd(p.redo, 1u)] = uint32_t(c);    // its output is rewritten by the redo pass
                else bad |= ls[k].position() != uint32_t(endpos);
            }
            if (bad) atomicExch(p.status, MHK_STATUS_CORRUPT);
        } else {
            // ---- end of the stream: whatever chunks exist, one at a time
            for (int k = 0; k < K; ++k) {
                const uint64_t c = c0 + uint64_t(k) * NT;
                if (c >= p.nchunks) break;
                const uint64_t first = c << p.chunk_shift;
                const uint32_t nsym = (p.n - first) >= S ? S : uint32_t(p.n - first);
                const uint64_t endb = (c + 1 < p.nchunks) ? (p.index[c + 1] & 0x00FFFFFFFFFFFFFFull) : p.nbits;
                decode_chunk_single<DIRECT>(lut, sub_base, tabs, p.payload, p.payload_bytes, p.nbits, p.index[c], endb, p.out + first, nsym, p.status);
            }
        }
    }
}


// ---- decode: one lane per chunk; p.prim / p.sec / p.sec_base / p.tree are the order-2 tables (general form, P = 8)
__global__ __launch_bounds__(256) void decode2_kernel(DecParams p) {
    if (p.d_nbits) { p.nbits = *p.d_nbits; p.payload_bytes = (p.nbits + 7) >> 3; }
    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    const DecTables tabs{p.sec, p.tree, p.P, 0u, 0u};
    const uint32_t S = 1u << p.chunk_shift;
    // redo_list: only the chunks the tile decoder handed over (p.redo[0] of them, numbers behind it)
    const uint64_t nwork = p.redo_list ? uint64_t(p.redo[0]) : p.nchunks;
    for (uint64_t w = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; w < nwork; w += uint64_t(gridDim.x) * blockDim.x) {
        const uint64_t c = p.redo_list ? uint64_t(p.redo[1u + w]) : w;
        const uint64_t entry = p.index[c];
        uint64_t pos = entry & IDX2_POS;
        uint32_t ctx = uint32_t(entry >> 48);
        const uint64_t endpos = (c + 1 < p.nchunks) ? (p.index[c + 1] & IDX2_POS) : p.nbits;
        const uint64_t first = c << p.chunk_shift;
        const uint32_t nsym = (p.n - first) >= S ? S : uint32_t(p.n - first);
        if (pos >= p.nbits || endpos < pos || endpos > p.nbits) { atomicExch(p.status, MHK_STATUS_CORRUPT); continue; }
        GranuleCursor bc;
        bc.init(src, pos);
        bool bad = false;
        uint8_t *o = p.out + first;
        // 64 symbols -> one 64-byte burst (four 16-byte stores back to back: whole HBM bursts even when the line
        // is evicted between two bursts; dword stores reach HBM as partial writes)
        uint32_t i = 0;
        for (; i + 64u <= nsym && !bad; i += 64u) {
            uint4 Q[4] = {};                                      // the pieces rotate through Q: the 16-symbol body stays rolled
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {
                uint32_t w[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t q = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        uint32_t used = 0;
                        const uint32_t sym = decode_one(p.prim, p.sec_base, tabs, src, bc, ctx, used, bad);
                        pos += used;
                        ctx = ((ctx << 8) | sym) & 0xFFFFu;
                        q |= sym << (8 * b);
                    }
                    w[d] = q;
                }
                Q[0] = Q[1]; Q[1] = Q[2]; Q[2] = Q[3]; Q[3] = make_uint4(w[0], w[1], w[2], w[3]);
            }
            uint4 *o16 = reinterpret_cast<uint4 *>(o + i);
#pragma unroll
            for (int u = 0; u < 4; ++u) o16[u] = Q[u];
        }
        for (; i < nsym && !bad; ++i) {                          // the ragged last chunk
            uint32_t used = 0;
            const uint32_t sym = decode_one(p.prim, p.sec_base, tabs, src, bc, ctx, used, bad);
            pos += used;
            ctx = ((ctx << 8) | sym) & 0xFFFFu;
            o[i] = uint8_t(sym);
        }
        if (bad || pos != endpos) atomicExch(p.status, MHK_STATUS_CORRUPT);
    }
}


hipError_t launch_decode_redo(DecParams p, hipStream_t st) {
    if (p.order == 2) {                                      // order 2: the one-lane-per-chunk decoder over the list
        p.redo_list = 1;
        const uint64_t want2 = (p.nchunks + 255) / 256;
        const uint64_t cap2 = uint64_t(cu_count()) * 8;
        hipLaunchKernelGGL(decode2_kernel, dim3(unsigned(want2 > cap2 ? cap2 : (want2 < 1 ? 1 : want2))), dim3(256), 0, st, p);
        return hipGetLastError();
    }
    auto r_lds = decode_kernel<DV_REDO_LDS>;
    auto r_l2d = decode_kernel<DV_REDO_L2_DIRECT>;
    hipError_t e = once_per_device(&DeviceState::redo_ready, [&] {
        const void *all[] = {(const void *)r_lds, (const void *)r_l2d};
        for (const void *f : all) {
            hipError_t r = allow_lds(f, DEC_LDS_MAX);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    });
    if (e != hipSuccess) return e;
    if (!p.sec_lds && !p.direct) return hipErrorInvalidValue;      // (the general L2 layout no longer exists: see dec_cfg)
    if (!p.sec_lds && p.P != 8) return hipErrorInvalidValue;
    const size_t lds = 1024 + (size_t(256) << p.P) * 2 + (p.sec_lds ? ((size_t(p.nsec) * 2 + 15) & ~size_t(15)) : 0);
    if (lds > size_t(DEC_LDS_MAX)) return hipErrorInvalidValue;
    const uint64_t rwant = (p.nchunks + DEC_THREADS - 1) / DEC_THREADS;
    const int rgrid = int(rwant > uint64_t(cu_count()) ? uint64_t(cu_count()) : rwant);
    hipLaunchKernelGGL(p.sec_lds ? r_lds : r_l2d, dim3(rgrid < 1 ? 1 : rgrid), dim3(DEC_THREADS), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_decode(DecParams p, void *d_ws, hipStream_t st) {
    p.status = reinterpret_cast<int *>(d_ws);
    p.redo = reinterpret_cast<uint32_t *>(static_cast<char *>(d_ws) + 64);      // [0] = count, then chunk numbers
    hipError_t e = hipMemsetAsync(d_ws, 0, 64 + 16, st);
    if (e != hipSuccess || p.nchunks == 0) return e;
    if (p.order == 2) {
        const uint64_t want2 = (p.nchunks + 255) / 256;
        const uint64_t cap2 = uint64_t(cu_count()) * 8;
        hipLaunchKernelGGL(decode2_kernel, dim3(unsigned(want2 > cap2 ? cap2 : want2)), dim3(256), 0, st, p);
        return hipGetLastError();
    }
    // which variant (dec_cfg above says what each is and why)
    if (!p.sec_lds && (p.P != 8 || !p.direct)) return hipErrorInvalidValue;       // the L2 layout is built with P = 8, uniform tables
    size_t lds = 1024 + (size_t(256) << p.P) * 2 + (p.sec_lds ? ((size_t(p.nsec) * 2 + 15) & ~size_t(15)) : 0);
    if (lds > size_t(DEC_LDS_MAX)) return hipErrorInvalidValue;
    const bool short_codes = p.nsec == 0 && p.P == 8;              // no second level at all: every code <= 8 bits
    const bool wide = p.sec_lds && short_codes && p.n > 0 && p.nbits * 10 > p.n * 8 * 6;      // ratio > 0.6
    int v;
    if (wide) v = DV_LDS_WIDE;
    else if (p.sec_lds) v = short_codes ? DV_LDS_SHORT : p.P == 8 ? DV_LDS_TWO_LEVEL_P8 : DV_LDS_TWO_LEVEL;
    else v = p.H == 2 ? DV_L2_DIRECT_H2 : p.H == 3 ? DV_L2_DIRECT_H3 : p.H == 4 ? DV_L2_DIRECT_H4 : p.H == 8 ? DV_L2_DIRECT_H8 : DV_L2_DIRECT;
    void (*const kern[DV_REDO_LDS])(DecParams) = {decode_kernel<DV_LDS_WIDE>, decode_kernel<DV_LDS_SHORT>, decode_kernel<DV_LDS_TWO_LEVEL>,
                                                  decode_kernel<DV_LDS_TWO_LEVEL_P8>, decode_kernel<DV_L2_DIRECT>, decode_kernel<DV_L2_DIRECT_H2>,
                                                  decode_kernel<DV_L2_DIRECT_H3>, decode_kernel<DV_L2_DIRECT_H4>, decode_kernel<DV_L2_DIRECT_H8>};
    static_assert(DV_LDS_WIDE == 0 && DV_L2_DIRECT_H8 == 8 && DV_REDO_LDS == 9, "the table above is indexed by DecVariant");
    e = once_per_device(&DeviceState::decode_ready, [&] {
        for (int i = 0; i < DV_REDO_LDS; ++i) {
            hipError_t r = allow_lds(reinterpret_cast<const void *>(kern[i]), DEC_LDS_MAX);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    });
    if (e != hipSuccess) return e;
    const uint64_t per_block = uint64_t(DEC_NT) * uint64_t(dec_cfg(v).k);
    const uint64_t want = (p.nchunks + per_block - 1) / per_block;
    const int grid = int(want > uint64_t(cu_count()) ? uint64_t(cu_count()) : want);
    (void)launch_set_word(reinterpret_cast<uint32_t *>(static_cast<char *>(d_ws) + 44), uint32_t(v) + 1u, st);   // mh_dev_decode_variant
    hipLaunchKernelGGL(kern[v], dim3(grid), dim3(DEC_NT), lds, st, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // chunks with a code longer than both table levels: normally none, and the pass returns at once
    return launch_decode_redo(p, st);
}


}  // namespace mhk
