// mh_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the Markov-Huffman hot path.
//
//   hist_o1_kernel     256x256 conditional histogram, LDS-resident packed counters        (a1)
//   hist_o0_kernel     256-bin histogram                                                  (a2)
//   encode_kernel      single pass: LDS codeword table, wave prefix-sum of bit lengths,
//                      LDS bit assembly, decoupled look-back across tiles, coalesced store (a9-a12)
//   decode_kernel      LDS 8-bit LUTs per context + tree-walk fallback, one lane per chunk (a13-a15)
//   build_index_kernel sequential index builder for streams that come without an index    (N1)
//
// (aN) = row of SURVEY.md §8(a).  All integer/bit work: no MFMA.  No CUDA idioms: waves are 64 wide,
// cross-lane traffic uses __shfl_up/__ballot on 64 lanes, inter-workgroup hand-off uses single 8-byte
// agent-scope relaxed atomics (the data is the flag).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mh_kernels.h"
#include "mh_model.hpp"

namespace mhk {

using mh::DEC16_INNER;
using mh::TREE_LEAF;
using mh::TREE_STRIDE;

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// (L, tail7) monoid: L = bit length of a run of codewords, tail = its last min(7, L) bits, right
// aligned.  combine(a, b) describes the concatenation a||b.  Identity = 0.
// 32-bit packing (inside a tile): L << 7 | tail.       L < 2^25
// 64-bit packing (tile descriptors): status << 62 | tail << 55 | L.   L < 2^55
__device__ __forceinline__ uint32_t comb32(uint32_t a, uint32_t b) {
    uint32_t lb = b >> 7;
    uint32_t tail = lb >= 7 ? (b & 127u) : (((a & 127u) << lb) | (b & 127u)) & 127u;
    return (((a >> 7) + lb) << 7) | tail;
}

constexpr uint64_t D_LMASK = (1ull << 55) - 1;
constexpr uint64_t D_PAYLOAD = (1ull << 62) - 1;
constexpr uint64_t D_AGG = 1ull << 62;
constexpr uint64_t D_PREFIX = 2ull << 62;

__device__ __forceinline__ uint64_t comb64(uint64_t a, uint64_t b) {
    uint64_t lb = b & D_LMASK;
    uint64_t ta = (a >> 55) & 127u, tb = (b >> 55) & 127u;
    uint64_t tail = lb >= 7 ? tb : (((ta << lb) | tb) & 127u);
    return (((a & D_LMASK) + lb) & D_LMASK) | (tail << 55);
}
__device__ __forceinline__ uint64_t widen(uint32_t p) { return (uint64_t(p & 127u) << 55) | uint64_t(p >> 7); }

__device__ __forceinline__ uint64_t ld_desc(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_desc(unsigned long long *p, uint64_t v) {
    __hip_atomic_store(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// histogram, order 1
// ------------------------------------------------------------------------------------------------
// 65536 counters do not fit LDS as u32 (256 KiB > 160 KiB), so two 15-bit counters plus a guard bit
// each share one LDS word: bits [14:0]+[15] and [30:16]+[31].  A returning ds_add tells the lane that
// took a counter from 0x7FFF to 0x8000; that lane subtracts the guard bit again and credits 32768 to
// the 64-bit counter in HBM.  A guard bit never carries into the neighbour because fewer than 32768
// adds can be in flight between the add that sets it and the subtract that clears it (the workgroup
// has 1024 lanes x 16 adds).  Slot order = enc_slot(window): the symbol is folded into the low bits so
// that skewed contexts spread over LDS banks.
constexpr int HIST_THREADS = 1024;
constexpr int HIST_LDS_BYTES = 32768 * 4;

__device__ __forceinline__ void hist_add(uint32_t *h, unsigned long long *counts, uint32_t window) {
    uint32_t slot = mh::enc_slot(window);
    uint32_t hiHalf = slot & 1u;
    uint32_t old = atomicAdd(&h[slot >> 1], hiHalf ? 0x10000u : 1u);
    uint32_t field = hiHalf ? (old >> 16) : (old & 0xFFFFu);
    if (field == 0x7FFFu) {
        atomicSub(&h[slot >> 1], hiHalf ? 0x80000000u : 0x8000u);
        uint32_t sym = slot >> 8, prev = (slot ^ sym) & 255u;
        atomicAdd(&counts[prev * 256u + sym], 32768ull);
    }
}

__global__ __launch_bounds__(HIST_THREADS) void hist_o1_kernel(const uint8_t *__restrict__ data, uint64_t n,
                                                              uint32_t prev0, unsigned long long *counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    for (int i = threadIdx.x; i < 32768 / 4; i += HIST_THREADS) reinterpret_cast<uint4 *>(h)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    const uint64_t nvec = n >> 4;  // whole 16-byte vectors
    const uint4 *vdata = reinterpret_cast<const uint4 *>(data);
    for (uint64_t v = uint64_t(blockIdx.x) * HIST_THREADS + threadIdx.x; v < nvec; v += uint64_t(gridDim.x) * HIST_THREADS) {
        uint4 x = vdata[v];
        uint32_t pb = v ? uint32_t(data[v * 16 - 1]) : prev0;
        hist_add(h, counts, ((x.x << 8) | pb) & 0xFFFFu);
        hist_add(h, counts, x.x & 0xFFFFu);
        hist_add(h, counts, (x.x >> 8) & 0xFFFFu);
        hist_add(h, counts, x.x >> 16);
        hist_add(h, counts, __builtin_amdgcn_alignbyte(x.y, x.x, 3) & 0xFFFFu);
        hist_add(h, counts, x.y & 0xFFFFu);
        hist_add(h, counts, (x.y >> 8) & 0xFFFFu);
        hist_add(h, counts, x.y >> 16);
        hist_add(h, counts, __builtin_amdgcn_alignbyte(x.z, x.y, 3) & 0xFFFFu);
        hist_add(h, counts, x.z & 0xFFFFu);
        hist_add(h, counts, (x.z >> 8) & 0xFFFFu);
        hist_add(h, counts, x.z >> 16);
        hist_add(h, counts, __builtin_amdgcn_alignbyte(x.w, x.z, 3) & 0xFFFFu);
        hist_add(h, counts, x.w & 0xFFFFu);
        hist_add(h, counts, (x.w >> 8) & 0xFFFFu);
        hist_add(h, counts, x.w >> 16);
    }
    // ragged tail (< 16 bytes): one lane of block 0
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t i = nvec << 4;
        uint32_t prev = i ? uint32_t(data[i - 1]) : prev0;
        for (; i < n; ++i) {
            uint32_t c = data[i];
            hist_add(h, counts, (c << 8) | prev);
            prev = c;
        }
    }
    __syncthreads();
    // flush: one 64-bit atomic per non-zero counter
    for (uint32_t w = threadIdx.x; w < 32768u; w += HIST_THREADS) {
        uint32_t v = h[w];
        uint32_t lo = v & 0xFFFFu, hi = v >> 16;
        uint32_t slot = w << 1;
        if (lo) {
            uint32_t sym = slot >> 8, prev = (slot ^ sym) & 255u;
            atomicAdd(&counts[prev * 256u + sym], (unsigned long long)lo);
        }
        if (hi) {
            uint32_t s1 = slot | 1u;
            uint32_t sym = s1 >> 8, prev = (s1 ^ sym) & 255u;
            atomicAdd(&counts[prev * 256u + sym], (unsigned long long)hi);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// histogram, order 0: 256 bins, one private copy per wave (16 x 1 KiB), u32 per workgroup
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HIST_THREADS) void hist_o0_kernel(const uint8_t *__restrict__ data, uint64_t n,
                                                              unsigned long long *counts) {
    __shared__ uint32_t h[16][256];
    for (int i = threadIdx.x; i < 16 * 256; i += HIST_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[threadIdx.x >> 6];
    const uint64_t nvec = n >> 4;
    const uint4 *vdata = reinterpret_cast<const uint4 *>(data);
    for (uint64_t v = uint64_t(blockIdx.x) * HIST_THREADS + threadIdx.x; v < nvec; v += uint64_t(gridDim.x) * HIST_THREADS) {
        uint4 x = vdata[v];
        uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(&mine[w[k] & 255u], 1u);
            atomicAdd(&mine[(w[k] >> 8) & 255u], 1u);
            atomicAdd(&mine[(w[k] >> 16) & 255u], 1u);
            atomicAdd(&mine[w[k] >> 24], 1u);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (uint64_t i = nvec << 4; i < n; ++i) atomicAdd(&mine[data[i]], 1u);
    __syncthreads();
    if (threadIdx.x < 256) {
        unsigned long long s = 0;
        for (int w = 0; w < 16; ++w) s += h[w][threadIdx.x];
        if (s) atomicAdd(&counts[threadIdx.x], s);
    }
}

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------
// One workgroup (1024 lanes, one per CU because the codeword table takes 128 KiB of LDS) pulls tiles
// of ENC_TILE input bytes from a ticket counter.  Per tile:
//   A  each lane loads 8 consecutive bytes (+ the byte before them), looks up 8 codewords in LDS,
//      concatenates them in registers (two <=48-bit groups), and the wave scans (L, tail7);
//   B  waves exchange their aggregates through LDS; wave 0 publishes the tile aggregate, looks back
//      over earlier tiles' descriptors for the tile's absolute bit offset and publishes the inclusive
//      prefix; every lane ORs its bits into the LDS staging image at tile-local alignment;
//   C  the staging image is funnel-shifted to the absolute alignment, byte-swapped to the stream's
//      MSB-first order and stored as coalesced dwords (byte stores at the two seams; the byte that
//      straddles two tiles is written by the later tile, which got the earlier bits as `tail7`).
// Codes longer than 12 bits are escapes into the full table in HBM/L2; a wave that sees one deposits
// symbol by symbol, and a tile whose bits exceed the staging image is emitted in several rounds.
constexpr int ENC_THREADS = 1024;
constexpr int ENC_WAVES = ENC_THREADS / 64;
constexpr int ENC_SPL = 8;                          // symbols per lane per tile
constexpr int ENC_TILE = ENC_THREADS * ENC_SPL;     // 8192 input bytes
constexpr int ENC_STAGE_WORDS = ENC_TILE * mh::ENC16_MAX_LEN / 32;   // 3072 words = 12 KiB
constexpr int ENC_STAGE_BITS = ENC_STAGE_WORDS * 32;
constexpr int ENC_MAX_CPT = ENC_TILE / 256;         // chunks per tile at the smallest chunk size
constexpr int ENC_LDS_BYTES = 131072 + (ENC_STAGE_WORDS + 4) * 4 + ENC_WAVES * 4 + ENC_MAX_CPT * 8 + 64;
constexpr uint32_t SPIN_LIMIT = 1u << 22;

// OR a left-aligned string (first bit at bit 63 of `vl`) into the staging image at tile-local bit
// offset `o`.  CLIP: only words inside [wbase, wbase + ENC_STAGE_WORDS) are touched.
template <bool CLIP>
__device__ __forceinline__ void deposit(uint32_t *stage, uint64_t vl, uint32_t o, uint32_t wbase) {
    uint32_t hi = uint32_t(vl >> 32), lo = uint32_t(vl);
    uint32_t sh = o & 31u;
    uint32_t w0 = hi >> sh;
    uint32_t w1 = __builtin_amdgcn_alignbit(hi, lo, sh);
    uint32_t w2 = __builtin_amdgcn_alignbit(lo, 0u, sh);
    uint32_t wi = (o >> 5) - wbase;  // wraps when below the window; the unsigned compare rejects it
    if (CLIP) {
        if (w0 && wi < uint32_t(ENC_STAGE_WORDS)) atomicOr(&stage[wi], w0);
        if (w1 && wi + 1u < uint32_t(ENC_STAGE_WORDS)) atomicOr(&stage[wi + 1u], w1);
        if (w2 && wi + 2u < uint32_t(ENC_STAGE_WORDS)) atomicOr(&stage[wi + 2u], w2);
    } else {
        if (w0) atomicOr(&stage[wi], w0);
        if (w1) atomicOr(&stage[wi + 1u], w1);
        if (w2) atomicOr(&stage[wi + 2u], w2);
    }
}

// Phase C for one round: bits [0, lr) of the staging image sit at absolute bit offset s; `carry` holds
// the (s & 7) stream bits just before s.
__device__ __forceinline__ void write_out(const uint32_t *stage, uint8_t *out, uint64_t cap, uint64_t s,
                                          uint32_t carry, uint32_t lr, bool last) {
    uint64_t e = s + lr;
    uint64_t b0 = s >> 3, b1 = last ? (e + 7) >> 3 : e >> 3;
    if (b1 > cap) b1 = cap;      // capacity overrun is reported by the caller; never write past it
    if (b0 >= b1) return;
    uint64_t g0 = b0 >> 2, g1 = (b1 + 3) >> 2;
    uint32_t k = uint32_t(s & 7u);
    uint32_t cbits = carry & ((1u << k) - 1u);
    for (uint64_t g = g0 + threadIdx.x; g < g1; g += ENC_THREADS) {
        int32_t o = int32_t(int64_t(g << 5) - int64_t(s));   // |o| < 2^20
        uint32_t val;
        if (o < 0) {
            uint32_t m = uint32_t(-o);                       // 1..31
            val = (cbits << (32u - m)) | (stage[0] >> m);
        } else {
            uint32_t q = uint32_t(o) >> 5, r = uint32_t(o) & 31u;
            uint64_t two = (uint64_t(stage[q]) << 32) | stage[q + 1u];
            val = uint32_t(two >> (32u - r));
        }
        uint64_t lo = g << 2, hi = lo + 4;
        if (lo >= b0 && hi <= b1) {
            reinterpret_cast<uint32_t *>(out)[g] = __builtin_bswap32(val);
        } else {
            for (uint64_t b = (lo > b0 ? lo : b0); b < (hi < b1 ? hi : b1); ++b)
                out[b] = uint8_t(val >> (24u - 8u * uint32_t(b - lo)));
        }
    }
}

__global__ __launch_bounds__(ENC_THREADS) void encode_kernel(EncParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem + 131072);
    uint32_t *wagg = stage + ENC_STAGE_WORDS + 4;
    uint32_t *chunk_off = wagg + ENC_WAVES;
    uint32_t *chunk_prev = chunk_off + ENC_MAX_CPT;
    uint32_t *sh = chunk_prev + ENC_MAX_CPT;   // [0]=tile [1]=abort [2..3]=s [4]=carry [5]=round carry
    volatile uint32_t *vsh = sh;

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    for (int i = tid; i < 8192; i += ENC_THREADS)
        reinterpret_cast<uint4 *>(tab)[i] = reinterpret_cast<const uint4 *>(p.enc16)[i];
    for (int i = tid; i < ENC_STAGE_WORDS + 4; i += ENC_THREADS) stage[i] = 0;
    if (tid == 0) { sh[0] = atomicAdd(p.ticket, 1u); sh[1] = 0; }
    __syncthreads();

    const uint32_t S = 1u << p.chunk_shift;
    const uint32_t cpt = ENC_TILE >> p.chunk_shift;

    for (;;) {
        const uint32_t tile = vsh[0];
        if (tile >= p.ntiles || vsh[1]) break;

        // ---------------------------------------------------------------- phase A
        const uint64_t off = uint64_t(tile) * ENC_TILE + uint64_t(tid) * ENC_SPL;
        uint32_t lo = 0, hi = 0, nvalid = 0, pb = p.prev0;
        if (off < p.n) {
            uint64_t rem = p.n - off;
            nvalid = rem >= ENC_SPL ? ENC_SPL : uint32_t(rem);
            if (nvalid == ENC_SPL) {
                uint2 v = *reinterpret_cast<const uint2 *>(p.data + off);
                lo = v.x; hi = v.y;
            } else {
                for (uint32_t j = 0; j < nvalid; ++j) {
                    uint32_t b = p.data[off + j];
                    if (j < 4) lo |= b << (8 * j); else hi |= b << (8 * (j - 4));
                }
            }
            if (off) pb = p.data[off - 1];
        }
        uint32_t win[ENC_SPL];
        win[0] = ((lo << 8) | pb) & 0xFFFFu;
        win[1] = lo & 0xFFFFu;
        win[2] = (lo >> 8) & 0xFFFFu;
        win[3] = lo >> 16;
        win[4] = __builtin_amdgcn_alignbyte(hi, lo, 3) & 0xFFFFu;
        win[5] = hi & 0xFFFFu;
        win[6] = (hi >> 8) & 0xFFFFu;
        win[7] = hi >> 16;
        uint32_t ent[ENC_SPL];
        bool esc = false;
#pragma unroll
        for (int j = 0; j < ENC_SPL; ++j) {
            ent[j] = (uint32_t(j) < nvalid) ? uint32_t(tab[mh::enc_slot(win[j])]) : 0u;
            esc |= ent[j] >= 0xD000u;
        }
        const bool slow = __any(esc) != 0;     // wave-uniform

        uint64_t g0 = 0, g1 = 0;               // fast path: two groups of four codes, right aligned
        uint32_t gl0 = 0, gl1 = 0;
        uint32_t slen[ENC_SPL];                // slow path: per-symbol codes
        uint64_t scode[ENC_SPL];
        uint32_t P;                            // (L << 7) | tail7 of this lane
        if (!slow) {
            uint32_t l[ENC_SPL], c[ENC_SPL];
#pragma unroll
            for (int j = 0; j < ENC_SPL; ++j) { l[j] = ent[j] >> 12; c[j] = ent[j] & 0xFFFu; }
            uint32_t p01 = (c[0] << l[1]) | c[1], l01 = l[0] + l[1];
            uint32_t p23 = (c[2] << l[3]) | c[3], l23 = l[2] + l[3];
            uint32_t p45 = (c[4] << l[5]) | c[5], l45 = l[4] + l[5];
            uint32_t p67 = (c[6] << l[7]) | c[7], l67 = l[6] + l[7];
            g0 = (uint64_t(p01) << l23) | p23; gl0 = l01 + l23;
            g1 = (uint64_t(p45) << l67) | p67; gl1 = l45 + l67;
            uint32_t t0 = uint32_t(g0) & 127u, t1 = uint32_t(g1) & 127u;
            uint32_t tail = gl1 >= 7 ? t1 : (((t0 << gl1) | t1) & 127u);
            P = ((gl0 + gl1) << 7) | tail;
        } else {
            P = 0;
#pragma unroll
            for (int j = 0; j < ENC_SPL; ++j) {
                uint32_t e = ent[j];
                uint32_t l = e >> 12;
                uint64_t c = e & 0xFFFu;
                if (e >= 0xD000u) {
                    uint32_t nat = ((win[j] & 255u) << 8) | (win[j] >> 8);   // prev * 256 + sym
                    l = p.len8[nat];
                    c = p.code64[nat];
                    if (l > 64u) { l = 0; c = 0; }   // rejected on the host; keep the device safe
                }
                slen[j] = l; scode[j] = c;
                uint32_t t = uint32_t(c) & 127u;     // c < 2^l, so for l < 7 this is the whole code
                P = comb32(P, (l << 7) | t);
            }
        }
        // inclusive wave scan of (L, tail7)
        uint32_t inc = P;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(inc, d);
            if (lane >= uint32_t(d)) inc = comb32(t, inc);
        }
        uint32_t exc = __shfl_up(inc, 1);
        if (lane == 0) exc = 0;
        if (lane == 63) wagg[wave] = inc;
        __syncthreads();                                                   // ---- barrier 1

        // ---------------------------------------------------------------- phase B
        uint32_t base = 0, tile_agg = 0;
#pragma unroll
        for (int w = 0; w < ENC_WAVES; ++w) {
            uint32_t a = wagg[w];
            if (uint32_t(w) < wave) base = comb32(base, a);
            tile_agg = comb32(tile_agg, a);
        }
        const uint32_t my_off = (base >> 7) + (exc >> 7);    // tile-local bit offset of this lane
        const uint32_t tile_bits = tile_agg >> 7;

        if (wave == 0) {
            // publish the aggregate, look back for the exclusive prefix, publish the inclusive prefix
            uint64_t agg64 = widen(tile_agg);
            uint64_t excl = p.seed & D_PAYLOAD;
            bool timeout = false;
            if (tile != 0) {
                if (lane == 0) st_desc(&p.desc[tile], D_AGG | agg64);
                excl = 0;
                int64_t top = int64_t(tile) - 1;
                for (;;) {
                    int64_t mine = top - 63 + int64_t(lane);
                    uint64_t v = D_PREFIX | (p.seed & D_PAYLOAD);      // virtual tile -1
                    uint32_t spins = 0;
                    for (;;) {
                        if (mine >= 0) v = ld_desc(&p.desc[mine]);
                        if (__all((v >> 62) != 0)) break;
                        if (++spins > SPIN_LIMIT) { timeout = true; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (timeout) break;
                    uint64_t pm = __ballot((v >> 62) == 2);
                    int ptop = pm ? 63 - __builtin_clzll(pm) : -1;
                    uint64_t x = (int(lane) >= ptop) ? (v & D_PAYLOAD) : 0ull;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        uint64_t t = __shfl_up(x, d);
                        if (lane >= uint32_t(d)) x = comb64(t, x);
                    }
                    uint64_t window = __shfl(x, 63);
                    excl = comb64(window, excl);
                    if (pm) break;
                    top -= 64;
                }
            }
            if (lane == 0) {
                if (timeout) {
                    sh[1] = 1;
                    atomicExch(p.status, MHK_STATUS_TIMEOUT);
                } else {
                    st_desc(&p.desc[tile], D_PREFIX | comb64(excl, agg64));
                    sh[2] = uint32_t(excl & D_LMASK);
                    sh[3] = uint32_t((excl & D_LMASK) >> 32);
                    sh[4] = uint32_t(excl >> 55) & 127u;
                }
            }
        }
        // chunk index bookkeeping: the lane that starts a chunk records its offset and context
        if ((tid * ENC_SPL & (S - 1u)) == 0u) {
            uint32_t c = (tid * ENC_SPL) >> p.chunk_shift;
            chunk_off[c] = my_off;
            chunk_prev[c] = pb;
        }

        const uint32_t nrounds = tile_bits <= uint32_t(ENC_STAGE_BITS) ? 1u : (tile_bits + ENC_STAGE_BITS - 1) / ENC_STAGE_BITS;
        for (uint32_t r = 0; r < nrounds; ++r) {
            const uint32_t wbase = r * ENC_STAGE_WORDS;
            if (!slow) {
                // fast-path waves never exceed 12 bits/symbol, but the tile may still be in multi-round
                // mode because of another wave: clip whenever nrounds > 1
                if (nrounds == 1) {
                    if (gl0) deposit<false>(stage, g0 << (64u - gl0), my_off, 0);
                    if (gl1) deposit<false>(stage, g1 << (64u - gl1), my_off + gl0, 0);
                } else {
                    if (gl0) deposit<true>(stage, g0 << (64u - gl0), my_off, wbase);
                    if (gl1) deposit<true>(stage, g1 << (64u - gl1), my_off + gl0, wbase);
                }
            } else {
                uint32_t o = my_off;
#pragma unroll
                for (int j = 0; j < ENC_SPL; ++j) {
                    if (slen[j]) deposit<true>(stage, scode[j] << (64u - slen[j]), o, wbase);
                    o += slen[j];
                }
            }
            __syncthreads();                                               // ---- barrier 2
            if (vsh[1]) break;                                             // look-back timed out
            const uint64_t s_tile = (uint64_t(vsh[3]) << 32) | vsh[2];
            const uint32_t lr = (r + 1 == nrounds) ? tile_bits - r * ENC_STAGE_BITS : uint32_t(ENC_STAGE_BITS);
            const uint64_t s_round = s_tile + uint64_t(r) * ENC_STAGE_BITS;
            // round r > 0 continues from the previous round: its carry sits in slot 5 + ((r - 1) & 1)
            const uint32_t carry = r == 0 ? vsh[4] : vsh[5u + ((r - 1u) & 1u)];
            const bool last = (tile + 1 == p.ntiles) && (r + 1 == nrounds);
            write_out(stage, p.out, p.cap, s_round, carry, lr, last);
            if (r == 0 && p.index && tid < cpt) {
                uint64_t ci = uint64_t(tile) * cpt + tid;
                if ((ci << p.chunk_shift) < p.n)
                    p.index[ci] = (uint64_t(chunk_prev[tid]) << 56) | (s_tile + chunk_off[tid]);
            }
            if (tid == 0) {
                sh[5u + (r & 1u)] = stage[ENC_STAGE_WORDS - 1] & 127u;    // slot not read this round
                if (last) {
                    *p.nbits = s_tile + tile_bits;
                    if (((s_tile + tile_bits + 7) >> 3) > p.cap) atomicExch(p.status, MHK_STATUS_CAPACITY);
                }
                if (r + 1 == nrounds) sh[0] = atomicAdd(p.ticket, 1u);
            }
            __syncthreads();                                               // ---- barrier 3
            for (int i = tid; i < ENC_STAGE_WORDS + 4; i += ENC_THREADS) stage[i] = 0;
            if (r + 1 < nrounds) __syncthreads();                          // multi-round only
        }
    }
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------
// Stream words are big-endian in stream order: word w holds stream bits [32w, 32w+32), first bit in
// bit 31.  Reads past the last payload byte return zero bits (src/bitbuffer.cpp:116-127).
struct BitSrc {
    const uint8_t *p;
    uint64_t full_words;   // payload_bytes / 4
    uint64_t bytes;
    __device__ __forceinline__ uint32_t word(uint64_t w) const {
        if (w < full_words) return __builtin_bswap32(reinterpret_cast<const uint32_t *>(p)[w]);
        uint32_t v = 0;
        for (uint32_t i = 0; i < 4; ++i) {
            uint64_t b = (w << 2) + i;
            if (b < bytes) v |= uint32_t(p[b]) << (24u - 8u * i);
        }
        return v;
    }
};

struct BitCursor {
    uint64_t buf;     // next bits, first at bit 63
    uint32_t cnt;     // valid bits in buf
    uint64_t next;    // next word index to fetch
    __device__ __forceinline__ void init(const BitSrc &src, uint64_t bitpos) {
        uint64_t w = bitpos >> 5;
        uint32_t sh = uint32_t(bitpos & 31u);
        buf = ((uint64_t(src.word(w)) << 32) | src.word(w + 1)) << sh;
        cnt = 64u - sh;
        next = w + 2;
    }
    __device__ __forceinline__ void refill(const BitSrc &src) {
        if (cnt <= 32u) {
            buf |= uint64_t(src.word(next++)) << (32u - cnt);
            cnt += 32u;
        }
    }
};

// Decodes one symbol; returns the symbol (0..255) or -1 on a corrupt stream.  *used = bits consumed.
template <typename LutPtr>
__device__ __forceinline__ int decode_one(LutPtr lut, const uint32_t *__restrict__ tree, const BitSrc &src,
                                          BitCursor &bc, uint32_t prev, uint32_t *used) {
    bc.refill(src);
    uint32_t e = lut[(prev << 8) | uint32_t(bc.buf >> 56)];
    if (e == 0) return -1;
    if (!(e & DEC16_INNER)) {
        uint32_t len = e >> 8;
        bc.buf <<= len; bc.cnt -= len; *used = len;
        return int(e & 255u);
    }
    // code longer than 8 bits: consume the window, then walk the context's tree bit by bit
    uint32_t node = e & 0x1FFu, n = 8;
    bc.buf <<= 8; bc.cnt -= 8;
    const uint32_t *t = tree + prev * TREE_STRIDE;
    for (int guard = 0; guard < 256; ++guard) {
        bc.refill(src);
        uint32_t bit = uint32_t(bc.buf >> 63);
        bc.buf <<= 1; bc.cnt -= 1; ++n;
        uint32_t pair = t[node];
        uint32_t c = bit ? (pair >> 16) : (pair & 0xFFFFu);
        if (c & TREE_LEAF) { *used = n; return int(c & 255u); }
        node = c;
    }
    return -1;
}

constexpr int DEC_THREADS = 1024;
constexpr int DEC_LDS_BYTES = 131072;

__global__ __launch_bounds__(DEC_THREADS) void decode_kernel(DecParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *lut = reinterpret_cast<uint16_t *>(smem);
    for (int i = threadIdx.x; i < 8192; i += DEC_THREADS)
        reinterpret_cast<uint4 *>(lut)[i] = reinterpret_cast<const uint4 *>(p.dec16)[i];
    __syncthreads();

    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    const uint32_t S = 1u << p.chunk_shift;
    for (uint64_t chunk = uint64_t(blockIdx.x) * DEC_THREADS + threadIdx.x; chunk < p.nchunks;
         chunk += uint64_t(gridDim.x) * DEC_THREADS) {
        const uint64_t entry = p.index[chunk];
        const uint64_t bitpos = entry & 0x00FFFFFFFFFFFFFFull;
        uint32_t prev = uint32_t(entry >> 56);
        const uint64_t first = chunk << p.chunk_shift;
        const uint32_t nsym = (p.n - first) >= S ? S : uint32_t(p.n - first);
        if (bitpos > p.nbits) { atomicExch(p.status, MHK_STATUS_CORRUPT); continue; }
        BitCursor bc;
        bc.init(src, bitpos);
        uint8_t *o = p.out + first;
        uint32_t packed = 0;
        bool bad = false;
        for (uint32_t i = 0; i < nsym; ++i) {
            uint32_t used;
            int sym = decode_one(lut, p.tree, src, bc, prev, &used);
            if (sym < 0) { bad = true; break; }
            prev = uint32_t(sym);
            packed |= uint32_t(sym) << (8u * (i & 3u));
            if ((i & 3u) == 3u) {
                *reinterpret_cast<uint32_t *>(o + i - 3u) = packed;
                packed = 0;
            }
        }
        if (bad) { atomicExch(p.status, MHK_STATUS_CORRUPT); continue; }
        for (uint32_t i = nsym & ~3u; i < nsym; ++i) o[i] = uint8_t(packed >> (8u * (i & 3u)));
    }
}

// Sequential pass over a stream that has no index (one produced by the reference, src/coding.cpp
// has none): walks the whole payload once on one lane, recording (bit offset, context) every
// chunk_symbols symbols and the total symbol count.  The loop condition is the reference's
// `while(bi < length)` (src/coding.cpp:124).
__global__ void build_index_kernel(IdxParams p) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const BitSrc src{p.payload, p.payload_bytes >> 2, p.payload_bytes};
    BitCursor bc;
    bc.init(src, 0);
    uint64_t bi = 0, nsym = 0;
    uint32_t prev = p.prev0;
    const uint64_t S = 1ull << p.chunk_shift;
    while (bi < p.nbits) {
        if ((nsym & (S - 1)) == 0) {
            uint64_t ci = nsym >> p.chunk_shift;
            if (ci >= p.index_cap) { atomicExch(p.status, MHK_STATUS_CAPACITY); break; }
            p.index[ci] = (uint64_t(prev) << 56) | bi;
        }
        uint32_t used;
        int sym = decode_one(p.dec16, p.tree, src, bc, prev, &used);
        if (sym < 0) { atomicExch(p.status, MHK_STATUS_CORRUPT); break; }
        prev = uint32_t(sym);
        bi += used;
        ++nsym;
    }
    if (bi > p.nbits) atomicExch(p.status, MHK_STATUS_CORRUPT);
    *p.n_symbols = nsym;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int g_cu_count = 0;

static int cu_count() {
    if (g_cu_count == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        g_cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_cu_count;
}

static hipError_t allow_lds(const void *fn, int bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

hipError_t launch_hist_o1(const uint8_t *d_data, uint64_t n, uint32_t prev0, unsigned long long *d_counts, hipStream_t st) {
    hipError_t e = hipMemsetAsync(d_counts, 0, 65536 * sizeof(unsigned long long), st);
    if (e != hipSuccess || n == 0) return e;
    static bool once = false;
    if (!once) { e = allow_lds(reinterpret_cast<const void *>(hist_o1_kernel), HIST_LDS_BYTES); if (e != hipSuccess) return e; once = true; }
    uint64_t nvec = n >> 4;
    uint64_t want = (nvec + HIST_THREADS - 1) / HIST_THREADS;
    int grid = int(want < 1 ? 1 : (want > uint64_t(cu_count()) ? uint64_t(cu_count()) : want));
    hipLaunchKernelGGL(hist_o1_kernel, dim3(grid), dim3(HIST_THREADS), HIST_LDS_BYTES, st, d_data, n, prev0, d_counts);
    return hipGetLastError();
}

hipError_t launch_hist_o0(const uint8_t *d_data, uint64_t n, unsigned long long *d_counts, hipStream_t st) {
    hipError_t e = hipMemsetAsync(d_counts, 0, 256 * sizeof(unsigned long long), st);
    if (e != hipSuccess || n == 0) return e;
    uint64_t nvec = n >> 4;
    uint64_t want = (nvec + HIST_THREADS - 1) / HIST_THREADS;
    int grid = int(want < 1 ? 1 : (want > uint64_t(2 * cu_count()) ? uint64_t(2 * cu_count()) : want));
    hipLaunchKernelGGL(hist_o0_kernel, dim3(grid), dim3(HIST_THREADS), 0, st, d_data, n, d_counts);
    return hipGetLastError();
}

uint64_t encode_tiles(uint64_t n) { return (n + ENC_TILE - 1) / ENC_TILE; }

size_t encode_workspace_bytes(uint64_t n) {
    // [0,64): status(int) + ticket(u32) + pad ; then one descriptor per tile
    return 64 + size_t(encode_tiles(n)) * 8 + 64;
}

hipError_t launch_encode(EncParams p, void *d_ws, hipStream_t st) {
    // workspace layout: see encode_workspace_bytes
    unsigned char *ws = static_cast<unsigned char *>(d_ws);
    p.status = reinterpret_cast<int *>(ws);
    p.ticket = reinterpret_cast<unsigned int *>(ws + 4);
    p.desc = reinterpret_cast<unsigned long long *>(ws + 64);
    p.ntiles = uint32_t(encode_tiles(p.n));
    hipError_t e = hipMemsetAsync(ws, 0, 64 + size_t(p.ntiles) * 8, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(p.nbits, 0, 8, st);
    if (e != hipSuccess || p.n == 0) return e;
    static bool once = false;
    if (!once) { e = allow_lds(reinterpret_cast<const void *>(encode_kernel), ENC_LDS_BYTES); if (e != hipSuccess) return e; once = true; }
    int grid = int(p.ntiles < uint32_t(cu_count()) ? p.ntiles : uint32_t(cu_count()));
    hipLaunchKernelGGL(encode_kernel, dim3(grid), dim3(ENC_THREADS), ENC_LDS_BYTES, st, p);
    return hipGetLastError();
}

hipError_t launch_decode(DecParams p, void *d_ws, hipStream_t st) {
    p.status = reinterpret_cast<int *>(d_ws);
    hipError_t e = hipMemsetAsync(d_ws, 0, 64, st);
    if (e != hipSuccess || p.nchunks == 0) return e;
    static bool once = false;
    if (!once) { e = allow_lds(reinterpret_cast<const void *>(decode_kernel), DEC_LDS_BYTES); if (e != hipSuccess) return e; once = true; }
    uint64_t want = (p.nchunks + DEC_THREADS - 1) / DEC_THREADS;
    int grid = int(want > uint64_t(cu_count()) ? uint64_t(cu_count()) : want);
    hipLaunchKernelGGL(decode_kernel, dim3(grid), dim3(DEC_THREADS), DEC_LDS_BYTES, st, p);
    return hipGetLastError();
}

hipError_t launch_build_index(IdxParams p, void *d_ws, hipStream_t st) {
    p.status = reinterpret_cast<int *>(d_ws);
    hipError_t e = hipMemsetAsync(d_ws, 0, 64, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(p.n_symbols, 0, 8, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(build_index_kernel, dim3(1), dim3(64), 0, st, p);
    return hipGetLastError();
}

}  // namespace mhk
