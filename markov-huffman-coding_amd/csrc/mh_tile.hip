// mh_tile.hip — the wave-contiguous ("tile") decoder of the order-1 hot path.
//
// decode_kernel (mh_decode.hip) gives every lane its own 1024-symbol chunk: a lane's compressed bytes are
// ~0.7 KiB from its neighbour's, so every 32-byte input granule and every 64-byte output burst is a cache line
// of its own — 64 lines per wave instruction, 2.3x the algorithmic HBM traffic, the vector-memory pipe 63 % busy.
// Here a WAVE decodes 64 ADJACENT sub-chunks of 64 symbols (one per lane; K such tiles side by side): with the
// device-only fine index (mh_kernels.h, TileParams: context byte + bit offset per 64 symbols, written by the
// encoders and by the index builder) the wave's input is ONE contiguous piece of the payload — about 3 KiB per
// tile for Zipf(1.1) — and its output one contiguous 4 KiB per tile:
//   * the piece is loaded with coalesced 16-byte loads, bit-reversed inside its bytes (so that the first stream
//     bit of a dword is bit 0) and parked in a wave-private LDS region: every payload byte crosses the memory
//     pipe once, in whole lines;
//   * a lane's bit window is two LDS dwords and one v_alignbit — no register FIFO, no refill branches, ~12
//     registers of state per stream instead of ~60, so sixteen waves per CU run instead of eight;
//   * tables are indexed LSB-first (tree_pack_kernel, lsb = 1): first level (P bits, P <= 8) in LDS, uniform
//     second-level tables (2^H entries) gathered from L2 by every lane through a bounds-checked buffer load
//     (a leaf entry used as a table id lands past the end of the table: answered with 0, no cache access);
//   * 64 symbols per stream end in four 16-byte stores per lane; adjacent lanes complete whole lines.
// The LDS not taken by the first-level table is divided into the waves' regions; every workgroup first finds the
// largest piece any of its waves will stage and sizes its regions (and the number of waves it keeps) from that:
// nothing is assumed about the compression ratio, locally or globally.
// Chunks the tiles do not cover (the stream's ragged end, pieces larger than the whole LDS, codes longer than
// P + H bits) go to the redo pass of the chunk decoder (one lane per chunk, legacy tables, tree walk).
// Reference semantics: i_coding_provider::decompress, src/coding.cpp:118-157; bit order src/bitbuffer.cpp:12.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "mh_kernels.h"
#include "mh_model.hpp"

namespace mhk {

using mh::DEC16_LEAF;

constexpr int T_SUB = 1 << T_SUB_SHIFT;          // symbols per sub-chunk = per fine-index entry
constexpr int T_TILE = 64 * T_SUB;               // symbols per tile: one sub-chunk per lane
constexpr int T_NU = T_SUB / 16;                 // 16-byte pieces of a stream's output
constexpr int T_LPK = 1024 / T_SUB;              // lanes whose pieces make one KiB of output
constexpr int T_NG = 64 / T_LPK;
static_assert(T_SUB == 32 || T_SUB == 64, "the output transposition below is written for 32- or 64-symbol sub-chunks");
constexpr int T_THREADS = 1024;
constexpr int T_WAVES = T_THREADS / 64;
constexpr int T_LDS_BYTES = 163840;
constexpr uint64_t T_POS_MASK = 0x00FFFFFFFFFFFFFFull;

// bit offset of sub-chunk j's first symbol: the chunk index entry in front of it supplies the high bits
// (order 2: the fine entry's low half is the distance from the chunk's index entry; 0xFFFF = does not fit: a position
// past the end of the payload results, which fails the tile's sanity check and sends it to the redo pass)
constexpr uint64_t T_POS_MASK2 = 0x0000FFFFFFFFFFFFull;
template <int O2>
__device__ __forceinline__ uint64_t sub_pos(const TileParams &p, uint64_t j, uint32_t f) {
    if (O2) {
        const uint64_t base = p.index[(j << T_SUB_SHIFT) >> p.chunk_shift] & T_POS_MASK2;
        return (f & 0xFFFFu) == 0xFFFFu ? ~0ull >> 1 : base + (f & 0xFFFFu);
    }
    const uint64_t base = p.index[(j << T_SUB_SHIFT) >> p.chunk_shift] & T_POS_MASK;
    return base + ((f - uint32_t(base)) & FINE_POS_MASK);
}

// first payload byte a wave stages for a piece that starts at bit `start` (16-byte aligned), and the byte count
__device__ __forceinline__ uint64_t stage_first(uint64_t start) { return (start >> 3) & ~uint64_t(15); }

// LDS through absolute byte addresses: the kernel's address arithmetic happens on plain integers (a pointer
// derived from `smem` costs an add of the segment's base, which the compiler does not fold, on every access)
typedef __attribute__((address_space(3))) const uint32_t lds_u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ __attribute__((address_space(3))) const T *lds_ptr(uint32_t byte_addr) {
    return reinterpret_cast<__attribute__((address_space(3))) const T *>(byte_addr);
}
template <typename T>
__device__ __forceinline__ __attribute__((address_space(3))) T *lds_wptr(uint32_t byte_addr) {
    return reinterpret_cast<__attribute__((address_space(3))) T *>(byte_addr);
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *generic) {
    return uint32_t(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const void *)(generic)));
}

// diagnostic build only (MH_TILE_STAMP): shader-clock stamp, all of the wave's LDS / scalar loads drained first
__device__ __forceinline__ unsigned long long tile_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// inserts the low byte of `e` as byte j of `d`
__device__ __forceinline__ uint32_t tile_put_byte(uint32_t d, uint32_t e, int j) {
    const uint32_t sel = j == 0 ? 0x03020104u : j == 1 ? 0x03020400u : j == 2 ? 0x03040100u : 0x04020100u;
    return __builtin_amdgcn_perm(e, d, sel);
}

#ifdef MH_EXP_PROBES
#include "mh_tile_probes.hpp"
#else
__host__ __device__ constexpr uint32_t tile_probe_reserve(int) { return 0u; }
#endif

// K tiles per wave, PC = first-level width (compile time), HC = second-level height (0: read p.H)
// OUT: how a stream's 64 bytes leave (A/B, MH_TILE_OUT): 0 = a 16-byte store per 16 symbols (adjacent lanes 64 bytes apart),
// 1 = four such stores back to back at the end of the tile, 2 = through the wave's LDS region, transposed, so that every
// store instruction writes one contiguous KiB
// O2: order-2 tables of the live contexts (TileParams): 32-bit entries whose high half is the next context's slot
// G: how the second level is reached.  0 = the shipped form (every lane of every stream gathers); any other value exists
// only in the diagnostic library (mh_tile_probes.hpp: cost probes and the round-5 candidates, see profiles/r05/)
template <int K, int PC, int HC, int OUT, int WIN, int STAMP = 0, int O2 = 0, int G = 0>
__global__ __launch_bounds__(T_THREADS) void decode_tile_kernel(TileParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t P = PC;
    const uint32_t PRIM_BYTES = O2 ? ((p.nslots << P) * 4u + 15u) & ~15u : (256u << P) * 2u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (p.d_nbits) {
        p.nbits = *p.d_nbits;
        p.payload_bytes = (p.nbits + 7) >> 3;
    }
    uint16_t *lut = reinterpret_cast<uint16_t *>(smem);
    for (uint32_t i = tid; i < PRIM_BYTES / 16u; i += T_THREADS)
        reinterpret_cast<uint4 *>(lut)[i] = reinterpret_cast<const uint4 *>(p.prim)[i];
    __syncthreads();
    const uint64_t nsub = (p.n + T_SUB - 1) >> T_SUB_SHIFT;
    constexpr uint32_t RESERVE = tile_probe_reserve(G);           // (0 in the shipped kernel)
    const uint32_t free_bytes = uint32_t(T_LDS_BYTES) - PRIM_BYTES - RESERVE;
    // ---- [r4] geometry, per workgroup: the largest piece any of ITS waves will stage.  The workgroup owns blocks of T_WAVES
    // consecutive wave pieces (block B = blockIdx.x + m * gridDim.x), whatever the number of waves it keeps, so the set is
    // known before the regions are sized (a kernel of its own did this for the whole stream: 0.11 ms per 16 GiB, 1 % of a
    // 2 GiB shard's step).  A piece too large for all of the LDS is not counted: its chunks go to the redo pass below.
    // If all T_WAVES waves fit but for a few outsized pieces (the tail of 8 192 pieces' lengths), the regions are sized for
    // sixteen waves anyway and those pieces take the redo pass with the other leftovers: a sixteenth more streams in flight
    // for the whole workgroup is worth more than the chunk decoder costs on under 1 % of it.
    uint32_t maxb;
    {
        uint32_t *s_max = reinterpret_cast<uint32_t *>(smem + PRIM_BYTES + RESERVE);       // (the first region: not in use yet)
        if (tid < 2) s_max[tid] = 0;
        __syncthreads();
        const uint32_t cap_bytes = free_bytes - 32u;
        const uint32_t fit_all = ((free_bytes / uint32_t(T_WAVES)) & ~15u) - 32u;   // a piece of at most this many bytes: T_WAVES regions fit
        uint32_t mx = 0, over = 0;
        const uint64_t nblk = (p.ntiles + T_WAVES - 1) / T_WAVES;          // blocks of T_WAVES pieces in the stream
        const uint64_t mine = nblk > blockIdx.x ? (nblk - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;   // ... of this workgroup
        for (uint64_t i = tid; i < mine * T_WAVES; i += T_THREADS) {          // all of its pieces, spread over its threads
            const uint64_t t = (blockIdx.x + (i / T_WAVES) * gridDim.x) * T_WAVES + (i % T_WAVES);
            if (t >= p.ntiles) continue;
            const uint64_t j0 = t * (64u * K), j1 = j0 + 64u * K;
            const uint64_t start = sub_pos<O2>(p, j0, p.fine[j0]);
            const uint64_t end = j1 < nsub ? sub_pos<O2>(p, j1, p.fine[j1]) : p.nbits;
            if (end < start || end > p.nbits) continue;              // the decoding wave reports it
            const uint64_t bytes = ((end + 7) >> 3) - stage_first(start);
            if (bytes <= cap_bytes && uint32_t(bytes) > mx) mx = uint32_t(bytes);
            over += bytes > fit_all ? 1u : 0u;
        }
        if (mx) atomicMax(&s_max[0], mx);
        if (over) atomicAdd(&s_max[1], over);
        __syncthreads();
        maxb = s_max[0];
        const uint32_t nover = s_max[1];
        __syncthreads();                                              // (read by everybody before a wave stages into it)
        if (maxb > fit_all && uint64_t(nover) * 128u <= mine * T_WAVES) maxb = fit_all;
    }
    // ---- the waves' LDS regions: as many waves as regions of the largest piece fit (the rest leave)
    uint32_t region = (maxb + 15u + 16u) & ~15u;                  // + the dword behind the last one a window read may touch
    if (OUT == 2 && region < 1024u) region = 1024u;               // the output transposition needs one KiB
    uint32_t nw = free_bytes / region;
    if (nw == 0) { nw = 1; region = free_bytes & ~15u; }
    if (nw > uint32_t(T_WAVES)) nw = T_WAVES;
    const uint32_t H = HC ? uint32_t(HC) : p.H;
    constexpr uint32_t TSYM = uint32_t(K) * T_TILE;               // symbols per wave piece
    // ---- what the tiles do not cover: the chunks behind the last full piece (block 0, wave 0)
    if (blockIdx.x == 0 && wave == 0) {
        const uint64_t c_first = (p.ntiles * TSYM) >> p.chunk_shift;
        for (uint64_t c = c_first + lane; c < p.nchunks; c += 64) p.redo[1u + atomicAdd(p.redo, 1u)] = uint32_t(c);
    }
    if (wave >= nw) return;                                       // no barrier below this line
    unsigned char *reg = smem + PRIM_BYTES + RESERVE + wave * region;
    if (lds_addr_of(smem) != 0u) {                                // the first-level table is addressed from LDS address 0
        if (tid == 0) atomicExch(p.status, MHK_STATUS_CORRUPT);
        return;
    }
    const uint32_t reg_bit0 = lds_addr_of(reg) * 8u;              // LDS bit address of the region's first bit
    const __amdgpu_buffer_rsrc_t sec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.sec), 0, p.nsec ? int((p.nsec + 8u) * (O2 ? 4u : 2u)) : 0, 0x00020000);
    const uint32_t chunks_per_tile = T_TILE >> p.chunk_shift;     // >= 1: chunk_shift <= 12 (launch_decode_tile)

    unsigned long long seg[4] = {0, 0, 0, 0};
    for (uint64_t tb = blockIdx.x; tb * T_WAVES < p.ntiles; tb += gridDim.x)
    for (uint32_t tj = wave; tj < uint32_t(T_WAVES); tj += nw) {
        const uint64_t t = tb * T_WAVES + tj;
        if (t >= p.ntiles) break;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (STAMP) t0 = tile_stamp();
        // ---- positions: lane l, stream k decodes sub-chunk j = (t * K + k) * 64 + l
        uint64_t pos[K];
        uint32_t cf[K];                                           // low byte: the context (previous symbol)
        const uint64_t j0 = t * (64u * K);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t j = j0 + uint64_t(k) * 64u + lane;
            const uint32_t f = p.fine[j];
            pos[k] = sub_pos<O2>(p, j, f);
            cf[k] = O2 ? uint32_t(p.ctx2slot[f >> 16]) << 16 : f >> 24;     // order 2: the context's slot (0xFFFF: none)
        }
        const uint64_t jn = j0 + 64u * K;
        uint64_t end = p.nbits;
        if (jn < nsub) end = sub_pos<O2>(p, jn, p.fine[jn]);      // same address in every lane
        const uint64_t start = __shfl(pos[0], 0);
        const uint64_t b0 = stage_first(start);
        const uint64_t nbytes = ((end + 7) >> 3) - b0;
        // every sub-chunk starts inside the piece, in order (a damaged index fails here, not in the loop)
        bool sane = end >= start && end <= p.nbits;
#pragma unroll
        for (int k = 0; k < K; ++k) sane = sane && pos[k] >= start && pos[k] <= end;
        if (O2) {
            // a sub-chunk whose offset did not fit its fine entry, or whose context has no slot: the chunk decoder takes the piece
            bool fits = true;
#pragma unroll
            for (int k = 0; k < K; ++k) fits = fits && (cf[k] >> 16) < p.nslots;
            if (!__all(sane && fits)) {
                const uint64_t c0 = (t * (uint32_t(K) * T_TILE)) >> p.chunk_shift;
                for (uint32_t c = lane; c < (T_TILE >> p.chunk_shift) * K; c += 64) p.redo[1u + atomicAdd(p.redo, 1u)] = uint32_t(c0 + c);
                continue;
            }
        }
        if (!__all(sane)) {
            if (lane == 0) atomicExch(p.status, MHK_STATUS_CORRUPT);
            continue;
        }
        if (nbytes + 16u > region) {                              // larger than anything the LDS can hold: chunk decoder
            const uint64_t c0 = (t * TSYM) >> p.chunk_shift;
            for (uint32_t c = lane; c < chunks_per_tile * K; c += 64) p.redo[1u + atomicAdd(p.redo, 1u)] = uint32_t(c0 + c);
            continue;
        }
        if (STAMP) t1 = tile_stamp();
        // ---- stage the piece: coalesced 16-byte loads, bits reversed inside every byte, so that stream bit i of
        // the piece sits at bit i & 31 of LDS dword i >> 5
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(p.payload + b0);
            const uint32_t nvec = uint32_t((nbytes + 15) >> 4);
            for (uint32_t i = lane; i < nvec; i += 64u) {
                uint4 v = src[i];
                v.x = __builtin_bswap32(__builtin_bitreverse32(v.x));
                v.y = __builtin_bswap32(__builtin_bitreverse32(v.y));
                v.z = __builtin_bswap32(__builtin_bitreverse32(v.z));
                v.w = __builtin_bswap32(__builtin_bitreverse32(v.w));
                *reinterpret_cast<uint4 *>(reg + i * 16u) = v;
            }
            // the dword behind the piece (a window may read it) holds zero bits (src/bitbuffer.cpp:116-127)
            if (lane == 0) *reinterpret_cast<uint4 *>(reg + nvec * 16u) = make_uint4(0, 0, 0, 0);
        }
        if (STAMP) t2 = tile_stamp();
        // LDS operations of one wave execute in order: the reads below see the writes above
        uint32_t q[K], q0[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { q[k] = reg_bit0 + uint32_t(pos[k] - b0 * 8u); q0[k] = q[k]; }
        uint32_t leafacc = DEC16_LEAF;
        uint4 *o16[K];
#pragma unroll
        for (int k = 0; k < K; ++k) o16[k] = reinterpret_cast<uint4 *>(p.out + ((j0 + uint64_t(k) * 64u + lane) << T_SUB_SHIFT));
        uint4 Q[OUT ? K : 1][T_NU];                               // OUT 1, 2: the stream's output bytes; the pieces rotate through
        // WIN 1: the bit window lives in registers — lo/hi hold the next cnt stream bits (first in bit 0 of lo), `ahead`
        // the LDS dword behind them, loaded one refill early; every second symbol a lane with fewer than 32 bits
        // left takes `ahead` in (two table-resolved codes are at most 2 (P + H) <= 32 bits).  One masked ds_read_b32
        // per ~5.5 symbols instead of a ds_read2_b32 per symbol: the LDS array was busy 60 % of the kernel's time,
        // 60 % of that bank conflicts (profiles/r03), most of it window traffic.
        uint32_t lo[K], hi[K], cnt[K], ahead[K], wnext[K];
        if (WIN) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t wa = (q[k] >> 3) & ~3u, sh = q[k] & 31u;
                const lds_u32 *wp = lds_ptr<uint32_t>(wa);
                const uint32_t d0 = wp[0], d1 = wp[1];
                ahead[k] = wp[2];
                wnext[k] = wa + 12u;
                lo[k] = __builtin_amdgcn_alignbit(d1, d0, sh);
                hi[k] = d1 >> sh;
                cnt[k] = 64u - sh;
            }
        }
        // One symbol of every stream per step, in phases that the scheduler may not mix (it otherwise finishes
        // one stream's window before it asks for the next one's: two LDS round trips in a row instead of one):
        //   A  (WIN 0) window dwords on their way (one ds_read2_b32 per stream)
        //   B  window, first-level lookups on their way
        //   C  second-level gathers on their way (every lane; a leaf indexes past the end: 0, no cache access)
        //   D  the resolving entry, position, context, output byte
#pragma unroll 1
        for (int u = 0; u < T_NU; ++u) {                          // 16 symbols -> one 16-byte piece per stream
            uint32_t w4[K][4];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                uint32_t w0[K], w1[K], win[K], e[K], e2[K];
                if (WIN) {
                    if ((j & 1) == 0) {
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            if (cnt[k] < 32u) {
                                const uint64_t t = uint64_t(ahead[k]) << cnt[k];
                                lo[k] |= uint32_t(t);
                                hi[k] = uint32_t(t >> 32);            // (fewer than 32 bits left: hi was empty)
                                cnt[k] += 32u;
                                ahead[k] = *lds_ptr<uint32_t>(wnext[k]);
                                wnext[k] += 4u;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const lds_u32 *wp = lds_ptr<uint32_t>((q[k] >> 3) & ~3u);
                        w0[k] = wp[0];
                        w1[k] = wp[1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    win[k] = WIN ? lo[k] : __builtin_amdgcn_alignbit(w1[k], w0[k], q[k]);   // 32 stream bits from bit q on, first in bit 0
                    // byte address of the entry (the table starts at LDS address 0, checked at entry): context << (P + 1) | bits << 1
                    if (O2) {                                      // slot << (P + 2) | bits << 2: 32-bit entries
                        // (column XOR-ed with the slot: text's frequent codes otherwise put every lane on the same few banks)
                        e[k] = *lds_ptr<uint32_t>((((win[k] ^ (cf[k] >> 16)) << 2) & ((4u << P) - 4u)) | ((cf[k] >> 16) << (P + 2)));
                    } else {
                        const uint32_t csh = P == 7 ? __builtin_amdgcn_perm(0u, cf[k], 0x0C0C000Cu)       // byte 0 -> byte 1
                                                    : (cf[k] & 255u) << (P + 1);
                        e[k] = *lds_ptr<uint16_t>(((win[k] << 1) & ((2u << P) - 2u)) | csh);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef MH_EXP_PROBES
                if (G != 0 && !O2) tile_second_probe<K, G, PC>(e, win, cf, H, sec_rsrc, lane, e2);
                else
#endif
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (O2) {      // (a leaf carries bit 15: shifted by H + 2 it lies past the (nslots << P) << H entries whatever its high half holds)
                        const uint32_t idx2 = (e[k] << (H + 2)) | ((win[k] >> (P - 2)) & ((4u << H) - 4u));
                        e2[k] = uint32_t(__builtin_amdgcn_raw_buffer_load_b32(sec_rsrc, int(idx2), 0, 0));
                    } else {
                        const uint32_t idx2 = (e[k] << (H + 1)) | ((win[k] >> (P - 1)) & ((2u << H) - 2u));   // byte offset of the entry
                        e2[k] = uint32_t(uint16_t(__builtin_amdgcn_raw_buffer_load_b16(sec_rsrc, int(idx2), 0, 0)));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    uint32_t ef;
                    asm("v_max_u32 %0, %1, %2" : "=v"(ef) : "v"(e[k]), "v"(e2[k]));    // (opaque: keeps the value a plain 32-bit one)
                    leafacc &= ef;
                    const uint32_t len = __builtin_amdgcn_ubfe(ef, 8, 5);
                    if (WIN) {
                        lo[k] = __builtin_amdgcn_alignbit(hi[k], lo[k], len);
                        hi[k] >>= len;
                        cnt[k] -= len;
                    } else {
                        q[k] += len;
                    }
                    cf[k] = ef;
                    w4[k][j >> 2] = (j & 3) == 0 ? (ef & 255u) : tile_put_byte(w4[k][j >> 2], ef, j & 3);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint4 v = make_uint4(w4[k][0], w4[k][1], w4[k][2], w4[k][3]);
                if (OUT == 0) o16[k][u] = v;
                else {
#pragma unroll
                    for (int i = 0; i + 1 < T_NU; ++i) Q[k][i] = Q[k][i + 1];
                    Q[k][T_NU - 1] = v;
                }
            }
        }
        if (STAMP) t3 = tile_stamp();
        if (OUT == 1) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int u = 0; u < T_NU; ++u) o16[k][u] = Q[k][u];
        }
        if (OUT == 2) {
            // The input piece is used up: its LDS region now turns the tile's output around, one KiB (T_LPK lanes' pieces) at a
            // time, so that every store instruction writes 64 x 16 contiguous bytes.  A lane's piece u goes to slot
            // u ^ swz(lane) of its T_SUB bytes: the eight lanes one ds_write_b128 group serves then hit different bank quads.
            const uint32_t rb = reg_bit0 >> 3;
            auto swz = [](uint32_t l) -> uint32_t { return T_SUB == 64 ? (l >> 1) & 3u : (l >> 2) & 1u; };
#pragma unroll
            for (int k = 0; k < K; ++k) {
#pragma unroll
                for (int g = 0; g < T_NG; ++g) {
                    if (lane / uint32_t(T_LPK) == uint32_t(g)) {
#pragma unroll
                        for (int u = 0; u < T_NU; ++u)
                            *lds_wptr<u32x4>(rb + (lane % uint32_t(T_LPK)) * uint32_t(T_SUB) + ((uint32_t(u) ^ swz(lane)) << 4)) =
                                u32x4{Q[k][u].x, Q[k][u].y, Q[k][u].z, Q[k][u].w};
                    }
                    __builtin_amdgcn_wave_barrier();                    // (compiler fence: other lanes' writes, same wave, in order)
                    asm volatile("" ::: "memory");
                    const uint32_t l = lane / uint32_t(T_NU), u = lane % uint32_t(T_NU);       // reader: piece u of lane l (within the group)
                    const u32x4 v = *lds_ptr<u32x4>(rb + l * uint32_t(T_SUB) + ((u ^ swz(l)) << 4));
                    reinterpret_cast<uint4 *>(p.out + ((j0 + uint64_t(k) * 64u + uint32_t(g) * uint32_t(T_LPK)) << T_SUB_SHIFT))[lane] = make_uint4(v.x, v.y, v.z, v.w);
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
                }
            }
        }
        // ---- every sub-chunk must end exactly where the next one starts (null entries, a wrong table or a damaged
        // stream all miss it); a code that neither table level resolves sends the tile's chunks to the redo pass
        if (WIN) {
#pragma unroll
            for (int k = 0; k < K; ++k) q[k] = (wnext[k] - 4u) * 8u - cnt[k];       // `ahead` is the dword at wnext - 4: not in the window yet
        }
        const uint32_t qend = reg_bit0 + uint32_t(end - b0 * 8u);
        bool bad = false;
        uint32_t unresolved = (leafacc & DEC16_LEAF) ? 0u : 1u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            uint32_t nxt = __shfl_down(q0[k], 1);
            const uint32_t first_of_next = k + 1 < K ? __shfl(q0[k + 1 < K ? k + 1 : k], 0) : qend;
            if (lane == 63) nxt = first_of_next;
            bad = bad || q[k] != nxt;
        }
        if (STAMP) {
            const unsigned long long t4 = tile_stamp();
            seg[0] += t1 - t0; seg[1] += t2 - t1; seg[2] += t3 - t2; seg[3] += t4 - t3;
        }
#ifdef MH_EXP_PROBES
        if (p.probe) continue;
#endif
        if (__any(unresolved)) {                                  // rare: all chunks of this piece again, with the walk
            const uint64_t c0 = (t * TSYM) >> p.chunk_shift;
            for (uint32_t c = lane; c < chunks_per_tile * K; c += 64) p.redo[1u + atomicAdd(p.redo, 1u)] = uint32_t(c0 + c);
        } else if (__any(bad)) {
            if (lane == 0) atomicExch(p.status, MHK_STATUS_CORRUPT);
        }
    }
    if (STAMP && lane == 0) {                                     // cycles per segment, summed over the waves (diagnostic build)
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(p.status) + 1;   // bytes 8..39 of the status block
        for (int i = 0; i < 4; ++i) atomicAdd(&dst[i], seg[i]);
    }
}

// ---- index builder, fast path (the reference's streams carry no index, src/coding.cpp:35-59) ---------
// The segment iteration of mh_index.hip gives a lane 4096 bits of payload 512 bytes from its neighbour's and gathers both
// table levels from L2: three such passes cost ten times the decode they prepare.  Here a WAVE takes 128 ADJACENT segments of
// IX_SEG_BITS bits (two per lane) — one contiguous 4 KiB of payload, staged through LDS exactly as the tile decoder stages
// its pieces — with the tile decoder's first level in LDS:
//   mode 0  a lane that does not know its start state begins warm_bits in front of its segment in context ' ' (Huffman
//           streams re-synchronise within a few symbols), notes the state it ENTERS its segment with (s16), decodes to the
//           segment's end and leaves the end state (e16) and the number of symbols that start in the segment (c16).
//           Segment 0 starts exact.  If s16[i] == e16[i - 1] for every i, every state is the true one (induction from
//           segment 0); the segments whose warm-up did not synchronise are listed (index_tile_dirty_kernel) and decoded
//           again from e16[i - 1] (index_tile_repair_kernel: one thread each, general tables) until a pass lists none:
//           the same fixed point as the segment iteration's.
//   mode 1  true start states and symbol numbers (a prefix sum of c16) known: the lanes decode once more and write the
//           chunk index entry / fine index entry of every symbol whose number is a multiple of chunk_symbols / 64.
// Both modes cost a decode without output; no payload pass through L2-gathered first levels.
constexpr uint32_t IX_TILE_BYTES = IX_TILE_BITS / 8;
constexpr uint32_t IX_STAGE_LEAD = IX_WARM_BITS_MAX / 8;                    // bytes staged in front of the tile: the longest warm-up
constexpr uint32_t IX_STAGE_BYTES = IX_TILE_BYTES + IX_STAGE_LEAD + 32u;   // + what a code past the end and a window read may touch
constexpr uint32_t IX_REGION = IX_STAGE_BYTES + 16u;
static_assert(IX_STAGE_LEAD % 16 == 0 && IX_STAGE_BYTES % 16 == 0 && IX_TILE_BYTES % 16 == 0, "16-byte staging loads");
static_assert(IX_TILE_SEGS == 128, "two segments per lane");

template <int MODE, int PC>
__global__ __launch_bounds__(T_THREADS) void index_tile_kernel(IdxParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t P = PC, PRIM_BYTES = (256u << P) * 2u;
    constexpr int K = 2;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < PRIM_BYTES / 16u; i += T_THREADS)
        reinterpret_cast<uint4 *>(smem)[i] = reinterpret_cast<const uint4 *>(p.tprim)[i];
    __syncthreads();
    constexpr uint32_t NW = (uint32_t(T_LDS_BYTES) - PRIM_BYTES) / IX_REGION < uint32_t(T_WAVES) ? (uint32_t(T_LDS_BYTES) - PRIM_BYTES) / IX_REGION : uint32_t(T_WAVES);
    if (wave >= NW) return;                                       // no barrier below this line
    if (lds_addr_of(smem) != 0u) {                                // the first-level table is addressed from LDS address 0
        if (tid == 0) atomicExch(p.status, MHK_STATUS_CORRUPT);
        return;
    }
    unsigned char *reg = smem + PRIM_BYTES + wave * IX_REGION;
    const uint32_t reg_bit0 = lds_addr_of(reg) * 8u;
    const uint32_t H = p.tH;
    const __amdgpu_buffer_rsrc_t sec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.tsec), 0, p.tnsec ? int((p.tnsec + 8u) * 2u) : 0, 0x00020000);
    const uint64_t vec_total = (p.payload_bytes + 15u) >> 4;     // (the payload is readable up to the next 64-byte boundary: mh.h)
    const uint64_t cmask = (1ull << p.chunk_shift) - 1ull;

    for (uint64_t t = uint64_t(blockIdx.x) * NW + wave; t < p.ntile5; t += uint64_t(gridDim.x) * NW) {
        const uint64_t sb = t ? t * IX_TILE_BYTES - IX_STAGE_LEAD : 0ull;      // first staged payload byte (16-byte aligned)
        // ---- stage [tile start - warm-up, tile end + slack): bits reversed inside every byte, as the tile decoder does
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(p.payload + sb);
            const uint64_t left = vec_total - (sb >> 4);
            const uint32_t nvec = left < IX_STAGE_BYTES / 16u ? uint32_t(left) : IX_STAGE_BYTES / 16u;
            for (uint32_t i = lane; i < IX_REGION / 16u; i += 64u) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (i < nvec) {
                    v = src[i];
                    v.x = __builtin_bswap32(__builtin_bitreverse32(v.x));
                    v.y = __builtin_bswap32(__builtin_bitreverse32(v.y));
                    v.z = __builtin_bswap32(__builtin_bitreverse32(v.z));
                    v.w = __builtin_bswap32(__builtin_bitreverse32(v.w));
                }
                *reinterpret_cast<uint4 *>(reg + i * 16u) = v;
            }
        }
        // LDS operations of one wave execute in order: the reads below see the writes above
        // lane l takes segments l and l + 64 of the tile: two independent streams per lane hide each other's lookups.
        // The loop is branch-free: "entered its segment" and "done" are comparisons of the position (a finished stream stands
        // still), the entry state is caught by a select, everything a lane carries lives in vector registers.
        uint64_t seg[K], base[K];
        uint32_t qb0[K], qe0[K], q[K], ctx[K], k[K], S[K], badv[K], want_e[K], want_c[K], next_entry[K];
        bool active[K], last[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            seg[j] = t * IX_TILE_SEGS + uint32_t(j) * 64u + lane;
            const uint64_t b0 = seg[j] * IX_SEG_BITS;
            active[j] = b0 < p.nbits;
            const uint64_t e0 = b0 + IX_SEG_BITS < p.nbits ? b0 + IX_SEG_BITS : p.nbits;
            last[j] = active[j] && e0 == p.nbits;
            qb0[j] = reg_bit0 + uint32_t(b0 - sb * 8u);
            qe0[j] = active[j] ? reg_bit0 + uint32_t(e0 - sb * 8u) : qb0[j];       // (not active: done at once)
            k[j] = 0; S[j] = IX_INVALID; badv[j] = DEC16_LEAF; base[j] = 0; want_e[j] = want_c[j] = next_entry[j] = 0;   // (badv: bit 15 stays set while every symbol resolved)
            if (MODE == 0) {
                const bool exact = seg[j] == 0;
                ctx[j] = exact ? p.prev0 : 0x20u;
                const uint32_t room = qb0[j] - reg_bit0;          // (tile 0 has nothing staged in front of it)
                q[j] = exact || !active[j] ? qb0[j] : qb0[j] - (p.warm_bits < room ? p.warm_bits : room);
            } else {
                // true start state: the end state of the segment in front (segment 0: the stream's start)
                uint32_t pe = p.prev0 << 8;
                if (active[j] && seg[j]) pe = p.e16[seg[j] - 1];
                ctx[j] = pe >> 8;
                q[j] = active[j] ? qb0[j] + (pe & 255u) : qb0[j];
                want_e[j] = active[j] ? p.e16[seg[j]] : 0u;
                want_c[j] = active[j] ? (p.c16[seg[j]] & IX_C16_COUNT) : 0u;
            }
        }
        if (MODE == 1) {                                          // symbol numbers: the tile's base + the counts of the segments in front
            uint64_t run = p.tile_base[t];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                uint32_t inc = want_c[j];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= uint32_t(d)) inc += o; }
                base[j] = run + (inc - want_c[j]);
                next_entry[j] = (0u - uint32_t(base[j])) & ((1u << T_SUB_SHIFT) - 1u);
                run += __shfl(inc, 63);
            }
        }
        uint32_t overflow = 0;
        // One symbol step of both streams.  ENTERED: every stream of the wave is inside its segment (always so with true start
        // states; in the states pass from the moment the last lane has finished its warm-up) — no "entered?" test, no entry
        // state to catch: 22 instead of 30 vector instructions per stream step [r5].
        auto step = [&](auto entered_c) __attribute__((always_inline)) {
            constexpr bool ENTERED = decltype(entered_c)::value;
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    if (q[j] < qe0[j] && k[j] == next_entry[j]) { // symbol number base + k is a multiple of 64 (chunks are multiples of 64 symbols)
                        next_entry[j] += 1u << T_SUB_SHIFT;
                        const uint64_t g = base[j] + k[j];
                        const uint64_t pos = sb * 8u + (q[j] - reg_bit0);
                        if ((g & cmask) == 0) {
                            const uint64_t ci = g >> p.chunk_shift;
                            if (ci < p.index_cap) p.index[ci] = (uint64_t(ctx[j]) << 56) | pos; else overflow = 1;
                        }
                        if (p.fine && (g >> T_SUB_SHIFT) < p.fine_cap) p.fine[g >> T_SUB_SHIFT] = (ctx[j] << 24) | (uint32_t(pos) & FINE_POS_MASK);
                    }
                }
            }
            uint32_t w0[K], w1[K], win[K], e[K], e2[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const lds_u32 *wp = lds_ptr<uint32_t>((q[j] >> 3) & ~3u);
                w0[j] = wp[0];
                w1[j] = wp[1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                win[j] = __builtin_amdgcn_alignbit(w1[j], w0[j], q[j]);
                e[j] = *lds_ptr<uint16_t>(((win[j] << 1) & ((2u << P) - 2u)) | (ctx[j] << (P + 1)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint32_t idx2 = (e[j] << (H + 1)) | ((win[j] >> (P - 1)) & ((2u << H) - 2u));
                e2[j] = uint32_t(uint16_t(__builtin_amdgcn_raw_buffer_load_b16(sec_rsrc, int(idx2), 0, 0)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const bool go = q[j] < qe0[j];                            // still decoding (a finished stream stands still)
                const uint32_t ef = e[j] > e2[j] ? e[j] : e2[j];
                if (ENTERED) {
                    // a finished stream "decodes" a leaf of no bits that yields its own context: no selects on position and context.
                    // An entry that is no leaf (an empty context's null entry — a guess may run into one —, a code neither table
                    // level resolves) clears bit 15 of the accumulator; whatever its bits then do to the stream, the segment is
                    // marked and decoded again (states pass) or reported (entries pass), and the loop is bounded.
                    const uint32_t efm = go ? ef : (DEC16_LEAF | ctx[j]);
                    badv[j] &= efm;
                    q[j] += __builtin_amdgcn_ubfe(efm, 8, 5);
                    ctx[j] = efm & 255u;
                    k[j] += go ? 1u : 0u;
                } else {
                    const bool in = q[j] >= qb0[j];                       // inside its segment (or past it)
                    if (MODE == 0) S[j] = (in && S[j] == IX_INVALID) ? ((ctx[j] << 8) | (q[j] - qb0[j])) : S[j];
                    const uint32_t len = __builtin_amdgcn_ubfe(ef, 8, 5);
                    const bool ok = (ef & DEC16_LEAF) != 0u;              // (a leaf's length is at least one bit)
                    badv[j] &= (in && go && !ok) ? 0u : ~0u;
                    ctx[j] = (ok && go) ? (ef & 255u) : ctx[j];
                    q[j] += go ? (ok ? len : 1u) : 0u;
                    k[j] += (in && go) ? 1u : 0u;
                }
            }
        };
        uint32_t it = 0;
        if (MODE == 0) {
            // the warm-up phase: until every stream of the wave has entered its segment (a stream in its warm-up moves on by at least
            // one bit per step, so this ends within warm_bits steps; inactive streams stand at their segment's start)
#pragma unroll 1
            for (; it < IX_WARM_BITS_MAX + 64u; it += 4u) {
                if (__all(q[0] >= qb0[0] && q[1] >= qb0[1])) break;
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) step(std::false_type{});
            }
#pragma unroll
            for (int j = 0; j < K; ++j)                           // a stream that entered with the phase's last step: its entry state
                S[j] = (q[j] >= qb0[j] && S[j] == IX_INVALID) ? ((ctx[j] << 8) | (q[j] - qb0[j])) : S[j];
        }
#pragma unroll 1
        for (; it < IX_SEG_BITS + IX_WARM_BITS_MAX + 64u; it += 4u) {  // (a symbol takes at least one bit)
            if (!__any(q[0] < qe0[0] || q[1] < qe0[1])) break;       // [r5] asked once per four steps: a finished stream stands still
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) step(std::true_type{});
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (!active[j]) continue;
            const bool done = q[j] >= qe0[j];
            const uint32_t E = (ctx[j] << 8) | (q[j] - qe0[j]);
            if (MODE == 0) {
                if (S[j] == IX_INVALID && done) S[j] = E;         // (a code that spans the whole segment: entered and left at once)
                p.s16[seg[j]] = uint16_t(!(badv[j] & DEC16_LEAF) || !done ? IX_INVALID : S[j]);
                p.e16[seg[j]] = uint16_t(done ? E : IX_INVALID);
                p.c16[seg[j]] = uint16_t(k[j]);
            } else {
                // with true start states a null entry, an end state or a count that differs from the converged ones, or a stream
                // that does not end exactly at nbits (src/coding.cpp:124,158) means the stream does not belong to this table
                if (!(badv[j] & DEC16_LEAF) || !done || E != want_e[j] || k[j] != want_c[j] || (last[j] && (E & 255u) != 0u)) atomicExch(p.status, MHK_STATUS_CORRUPT);
            }
        }
        if (MODE == 1 && overflow) atomicExch(p.status, MHK_STATUS_CAPACITY);
    }
}

// ---- the segment decoder [r5] -----------------------------------------------------
// A stream that came without any index used to be decoded THREE times: index_tile_kernel<0> (states), <1> (index entries),
// then the tile decoder.  After the states pass and its repairs every segment's (IX_SEG_BITS bits) entry state and symbol count are
// known, so a prefix sum gives the output offset of its first symbol and the second pass can emit the bytes itself: same
// staging, same two streams per lane, same tables; a lane keeps the symbols of a round of 64 steps in registers (byte j of
// the round = step j in every lane: static register indices) and writes its run to out[first symbol of the segment + 64
// round ...) — no fine index is written or read and the payload is decoded twice, not three times.  End state and count of
// every segment must come out as converged, the stream must end exactly at nbits (src/coding.cpp:124,158), and nothing is
// written beyond out_cap (MHK_STATUS_CAPACITY).
constexpr uint32_t SD_REGION = IX_TILE_BYTES + 48u;            // the tile + what a code past its end and a window read may touch + a zero vector
static_assert(SD_REGION % 16 == 0, "16-byte staging stores");
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x2_unaligned __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint16_t u16_unaligned __attribute__((aligned(1)));

// A round = SD_GROUPS groups of 16 steps: the symbols of a round stay in registers (byte j of the round = step j in every lane)
constexpr int SD_GROUPS = 5;
constexpr uint32_t SD_ROUND = 16u * SD_GROUPS;
// m <= SD_ROUND bytes of a lane's round (Qk[g] = bytes 16 g ..) to d, any alignment: whole 16-byte groups, then 8 / 4 / 2 / 1
__device__ __forceinline__ void seg_store_run(uint8_t *d, const uint4 (&Qk)[SD_GROUPS], uint32_t m) {
#pragma unroll
    for (int g = 0; g < SD_GROUPS; ++g)
        if (m >= 16u * uint32_t(g + 1)) *reinterpret_cast<u32x4_unaligned *>(d + 16 * g) = u32x4_unaligned{Qk[g].x, Qk[g].y, Qk[g].z, Qk[g].w};
    const uint32_t gt = m >> 4, t = m & 15u;
    if (gt >= uint32_t(SD_GROUPS) || t == 0u) return;
    // the group the run ends in, by selects on plain values (a select between array ELEMENTS is turned into an indexed load,
    // and the whole array then lives in scratch memory)
    uint32_t c[SD_GROUPS][4];
#pragma unroll
    for (int g = 0; g < SD_GROUPS; ++g) {
        c[g][0] = Qk[g].x; c[g][1] = Qk[g].y; c[g][2] = Qk[g].z; c[g][3] = Qk[g].w;
#pragma unroll
        for (int i = 0; i < 4; ++i) asm("" : "+v"(c[g][i]));
    }
    uint32_t T[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        T[i] = c[SD_GROUPS - 1][i];
#pragma unroll
        for (int g = SD_GROUPS - 2; g >= 0; --g) T[i] = gt == uint32_t(g) ? c[g][i] : T[i];
    }
    uint8_t *dt = d + 16u * gt;
    if (t & 8u) { *reinterpret_cast<u32x2_unaligned *>(dt) = u32x2_unaligned{T[0], T[1]}; T[0] = T[2]; T[1] = T[3]; dt += 8; }
    if (t & 4u) { *reinterpret_cast<u32_unaligned *>(dt) = T[0]; T[0] = T[1]; dt += 4; }
    if (t & 2u) { *reinterpret_cast<u16_unaligned *>(dt) = uint16_t(T[0]); T[0] >>= 16; dt += 2; }
    if (t & 1u) *dt = uint8_t(T[0]);
}

template <int PC>
__global__ __launch_bounds__(T_THREADS) void segment_decode_kernel(IdxParams p, uint8_t *out, uint64_t out_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t P = PC, PRIM_BYTES = (256u << P) * 2u;
    constexpr int K = 2;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < PRIM_BYTES / 16u; i += T_THREADS)
        reinterpret_cast<uint4 *>(smem)[i] = reinterpret_cast<const uint4 *>(p.tprim)[i];
    __syncthreads();
    constexpr uint32_t NW = (uint32_t(T_LDS_BYTES) - PRIM_BYTES) / SD_REGION < uint32_t(T_WAVES) ? (uint32_t(T_LDS_BYTES) - PRIM_BYTES) / SD_REGION : uint32_t(T_WAVES);
    if (wave >= NW) return;                                       // no barrier below this line
    if (lds_addr_of(smem) != 0u) {                                // the first-level table is addressed from LDS address 0
        if (tid == 0) atomicExch(p.status, MHK_STATUS_CORRUPT);
        return;
    }
    unsigned char *reg = smem + PRIM_BYTES + wave * SD_REGION;
    const uint32_t reg_bit0 = lds_addr_of(reg) * 8u;
    const uint32_t H = p.tH;
    const __amdgpu_buffer_rsrc_t sec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.tsec), 0, p.tnsec ? int((p.tnsec + 8u) * 2u) : 0, 0x00020000);
    const uint64_t vec_total = (p.payload_bytes + 15u) >> 4;     // (the payload is readable up to the next 64-byte boundary: mh.h)

    for (uint64_t t = uint64_t(blockIdx.x) * NW + wave; t < p.ntile5; t += uint64_t(gridDim.x) * NW) {
        const uint64_t sb = t * IX_TILE_BYTES;                    // first staged payload byte (16-byte aligned)
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(p.payload + sb);
            const uint64_t left = vec_total - (sb >> 4);
            const uint32_t nvec = left < (IX_TILE_BYTES + 32u) / 16u ? uint32_t(left) : (IX_TILE_BYTES + 32u) / 16u;
            for (uint32_t i = lane; i < SD_REGION / 16u; i += 64u) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (i < nvec) {
                    v = src[i];
                    v.x = __builtin_bswap32(__builtin_bitreverse32(v.x));
                    v.y = __builtin_bswap32(__builtin_bitreverse32(v.y));
                    v.z = __builtin_bswap32(__builtin_bitreverse32(v.z));
                    v.w = __builtin_bswap32(__builtin_bitreverse32(v.w));
                }
                *reinterpret_cast<uint4 *>(reg + i * 16u) = v;
            }
        }
        // LDS operations of one wave execute in order: the reads below see the writes above
        uint64_t seg[K], base[K];
        uint32_t qe0[K], q[K], ctx[K], k[K], want_e[K], want_c[K], unres[K];
        bool active[K], last[K], skip[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            seg[j] = t * IX_TILE_SEGS + uint32_t(j) * 64u + lane;
            const uint64_t b0 = seg[j] * IX_SEG_BITS;
            active[j] = b0 < p.nbits;
            skip[j] = false;
            const uint64_t e0 = b0 + IX_SEG_BITS < p.nbits ? b0 + IX_SEG_BITS : p.nbits;
            last[j] = active[j] && e0 == p.nbits;
            const uint32_t qb0 = reg_bit0 + uint32_t(b0 - sb * 8u);
            qe0[j] = active[j] ? reg_bit0 + uint32_t(e0 - sb * 8u) : qb0;          // (not active: done at once)
            uint32_t pe = p.prev0 << 8;                          // true start state: the end state of the segment in front
            if (active[j] && seg[j]) pe = p.e16[seg[j] - 1];
            ctx[j] = pe >> 8;
            q[j] = active[j] ? qb0 + (pe & 255u) : qb0;
            want_e[j] = active[j] ? p.e16[seg[j]] : 0u;
            const uint32_t c = active[j] ? p.c16[seg[j]] : 0u;
            want_c[j] = c & IX_C16_COUNT;
            k[j] = 0; unres[j] = DEC16_LEAF;
            if (c & IX_C16_WALK) {                                // a code these tables do not resolve: the segment goes to the walk
                const uint32_t slot = atomicAdd(&p.changed[IDX_MAX_PASSES - 1], 1u);      // (segment_walk_emit_kernel; its bytes keep their place in the prefix)
                if (slot < p.dirty_cap) p.dirty_list[slot] = uint32_t(seg[j]);
                active[j] = false;
                q[j] = qe0[j];                                    // done at once; nothing of it is stored or checked here
                skip[j] = true;
            }
        }
        // output offsets: the tile's base + the counts of the segments in front
        bool fits = true;
        {
            uint64_t run = p.tile_base[t];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                uint32_t inc = want_c[j];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= uint32_t(d)) inc += o; }
                base[j] = run + (inc - want_c[j]);
                run += __shfl(inc, 63);
                fits = fits && base[j] + want_c[j] <= out_cap;
            }
        }
        if (!__all(fits)) {                                       // the caller's buffer is too small: nothing of this tile is written
            if (lane == 0) atomicExch(p.status, MHK_STATUS_CAPACITY);
            continue;
        }
#pragma unroll 1
        for (uint32_t r = 0; r < (IX_SEG_BITS + SD_ROUND - 1u) / SD_ROUND + 1u; ++r) {   // rounds of SD_ROUND steps (a symbol takes at least one bit)
            uint4 Q[K][SD_GROUPS];
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int g = 0; g < SD_GROUPS; ++g) Q[j][g] = make_uint4(0, 0, 0, 0);
            uint32_t ug = 0;
#pragma unroll 1
            for (; ug < uint32_t(SD_GROUPS); ++ug) {
                if (!__any(q[0] < qe0[0] || q[1] < qe0[1])) break;
                uint32_t w4[K][4];
                auto steps8 = [&](auto half_c) __attribute__((always_inline)) {
                    constexpr int J0 = decltype(half_c)::value * 8;
#pragma unroll
                    for (int jj = J0; jj < J0 + 8; ++jj) {
                        uint32_t w0[K], w1[K], win[K], e[K], e2[K];
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            const lds_u32 *wp = lds_ptr<uint32_t>((q[j] >> 3) & ~3u);
                            w0[j] = wp[0];
                            w1[j] = wp[1];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            win[j] = __builtin_amdgcn_alignbit(w1[j], w0[j], q[j]);
                            e[j] = *lds_ptr<uint16_t>(((win[j] << 1) & ((2u << P) - 2u)) | (ctx[j] << (P + 1)));
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            const uint32_t idx2 = (e[j] << (H + 1)) | ((win[j] >> (P - 1)) & ((2u << H) - 2u));
                            e2[j] = uint32_t(uint16_t(__builtin_amdgcn_raw_buffer_load_b16(sec_rsrc, int(idx2), 0, 0)));
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            const bool go = q[j] < qe0[j];                    // still decoding
                            const uint32_t ef = e[j] > e2[j] ? e[j] : e2[j];
                            // a finished stream "decodes" a leaf of no bits that yields its own context (index_tile_kernel's step)
                            const uint32_t efm = go ? ef : (DEC16_LEAF | ctx[j]);
                            unres[j] &= efm;                                  // (bit 15 cleared: some symbol neither level resolved)
                            q[j] += __builtin_amdgcn_ubfe(efm, 8, 5);
                            ctx[j] = efm & 255u;
                            k[j] += go ? 1u : 0u;
                            w4[j][jj >> 2] = (jj & 3) == 0 ? ctx[j] : tile_put_byte(w4[j][jj >> 2], efm, jj & 3);
                        }
                    }
                };
                steps8(std::integral_constant<int, 0>{});
#pragma unroll
                for (int j = 0; j < K; ++j) { w4[j][2] = 0; w4[j][3] = 0; }
                if (__any(q[0] < qe0[0] || q[1] < qe0[1])) steps8(std::integral_constant<int, 1>{});   // (asked per eight steps: the slowest lane sets the pace)
#pragma unroll
                for (int j = 0; j < K; ++j) {
#pragma unroll
                    for (int g = 0; g + 1 < SD_GROUPS; ++g) Q[j][g] = Q[j][g + 1];
                    Q[j][SD_GROUPS - 1] = make_uint4(w4[j][0], w4[j][1], w4[j][2], w4[j][3]);
                }
            }
            if (ug == 0u) break;                                  // (wave-uniform: every stream of the tile has finished)
            for (uint32_t i = ug; i < uint32_t(SD_GROUPS); ++i) { // the round's first group to Q[0]
#pragma unroll
                for (int j = 0; j < K; ++j)
#pragma unroll
                    for (int g = 0; g + 1 < SD_GROUPS; ++g) Q[j][g] = Q[j][g + 1];
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                // bytes of this round that are the segment's: never past its converged count (a stream that does not belong
                // to the table is reported below; it must not write into its neighbour's bytes)
                const uint32_t have = skip[j] ? 0u : (k[j] < want_c[j] ? k[j] : want_c[j]);
                const uint32_t m = have > SD_ROUND * r ? (have - SD_ROUND * r < SD_ROUND ? have - SD_ROUND * r : SD_ROUND) : 0u;
#ifdef MH_EXP_PROBES
                if (p.iter & 1u) continue;                         // diagnostic library, MH_SEG_PROBE=1: no stores (output wrong): what the loop alone costs
#endif
                if (m) seg_store_run(out + base[j] + SD_ROUND * r, Q[j], m);
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (!active[j]) continue;
            const bool done = q[j] >= qe0[j];
            const uint32_t E = (ctx[j] << 8) | (q[j] - qe0[j]);
            if (!(unres[j] & DEC16_LEAF) || !done || E != want_e[j] || k[j] != want_c[j] || (last[j] && (E & 255u) != 0u)) atomicExch(p.status, MHK_STATUS_CORRUPT);
        }
    }
}

hipError_t launch_segment_decode(const IdxParams &p, uint8_t *d_out, uint64_t out_cap, hipStream_t st) {
    if (p.tP != 7 || !p.tprim || p.order == 2 || !p.e16 || !p.c16 || !p.tile_base) return hipErrorInvalidValue;
    void (*kern)(IdxParams, uint8_t *, uint64_t) = segment_decode_kernel<7>;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        static std::mutex mu;
        static std::vector<std::pair<const void *, int>> done;
        std::lock_guard<std::mutex> lock(mu);
        const std::pair<const void *, int> key(reinterpret_cast<const void *>(kern), dev);
        if (std::find(done.begin(), done.end(), key) == done.end()) {
            e = hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
            if (e != hipSuccess) return e;
            done.push_back(key);
        }
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const uint64_t want = (p.ntile5 + T_WAVES - 1) / T_WAVES;
    const unsigned grid = unsigned(want < 1 ? 1 : (want > uint64_t(cus) ? uint64_t(cus) : want));
#ifdef MH_EXP_PROBES
    if (const char *pr = getenv("MH_SEG_PROBE")) { IdxParams pp = p; pp.iter = uint32_t(atoi(pr)); hipLaunchKernelGGL(kern, dim3(grid), dim3(T_THREADS), T_LDS_BYTES, st, pp, d_out, out_cap); return hipGetLastError(); }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T_THREADS), T_LDS_BYTES, st, p, d_out, out_cap);
    return hipGetLastError();
}

hipError_t launch_index_tile(const IdxParams &p, int mode, hipStream_t st) {
    if (p.tP != 7 || !p.tprim || p.order == 2) return hipErrorInvalidValue;
    if (mode == 0 && (p.warm_bits == 0 || p.warm_bits > IX_WARM_BITS_MAX)) return hipErrorInvalidValue;
    void (*kern)(IdxParams) = mode == 0 ? index_tile_kernel<0, 7> : index_tile_kernel<1, 7>;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        static std::mutex mu;
        static std::vector<std::pair<const void *, int>> done;
        std::lock_guard<std::mutex> lock(mu);
        const std::pair<const void *, int> key(reinterpret_cast<const void *>(kern), dev);
        if (std::find(done.begin(), done.end(), key) == done.end()) {
            e = hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
            if (e != hipSuccess) return e;
            done.push_back(key);
        }
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const uint64_t want = (p.ntile5 + T_WAVES - 1) / T_WAVES;
    const unsigned grid = unsigned(want < 1 ? 1 : (want > uint64_t(cus) ? uint64_t(cus) : want));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T_THREADS), T_LDS_BYTES, st, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------

size_t decode_tile_workspace_extra() { return 64; }


template <int K, int O2 = 0>
static hipError_t launch_tile_with(void (*kern)(TileParams), TileParams p, const DecParams &legacy, void *d_ws, hipStream_t st) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    // the dynamic-LDS attribute is a per-device, per-function setting: made once per (function, device), under a lock
    // (models are shared by threads and mh_set_device() may switch devices inside one process)
    {
        static std::mutex mu;
        static std::vector<std::pair<const void *, int>> done;
        std::lock_guard<std::mutex> lock(mu);
        const std::pair<const void *, int> key(reinterpret_cast<const void *>(kern), dev);
        if (std::find(done.begin(), done.end(), key) == done.end()) {
            e = hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
            if (e != hipSuccess) return e;
            done.push_back(key);
        }
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    p.ntiles = p.n / (uint64_t(K) * T_TILE);
    // workspace: [0,64) status | [64, 64 + 16) redo count ... as launch_decode lays it out; the geometry word lives in
    // the status block (bytes 4..7; bytes 8..39 take the diagnostic build's cycle sums)
    p.status = reinterpret_cast<int *>(d_ws);
    p.redo = reinterpret_cast<uint32_t *>(static_cast<char *>(d_ws) + 64);
    e = hipMemsetAsync(d_ws, 0, 64 + 16, st);
    if (e != hipSuccess) return e;
    if (p.ntiles) {
        const uint32_t prim_bytes = O2 ? (((p.nslots << p.P) * 4u + 15u) & ~15u) : (256u << p.P) * 2u;
        if (prim_bytes + 4096u > uint32_t(T_LDS_BYTES)) return hipErrorInvalidValue;
    }
    // one workgroup per CU (the first-level table takes most of the LDS); with few pieces, fewer workgroups
    const uint64_t want = (p.ntiles + T_WAVES - 1) / T_WAVES;
    const unsigned grid = unsigned(want < 1 ? 1 : (want > uint64_t(cus) ? uint64_t(cus) : want));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T_THREADS), T_LDS_BYTES, st, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    DecParams r = legacy;
    r.status = p.status;
    r.redo = p.redo;
    return launch_decode_redo(r, st);
}

template <int K, int OUT, int WIN = 0>
static hipError_t launch_tile_k(TileParams p, const DecParams &legacy, void *d_ws, hipStream_t st) {
    void (*kern[9])(TileParams) = {nullptr, nullptr, nullptr, nullptr, nullptr, decode_tile_kernel<K, 5, 0, OUT, WIN>, decode_tile_kernel<K, 6, 0, OUT, WIN>,
                                   decode_tile_kernel<K, 7, 0, OUT, WIN>, decode_tile_kernel<K, 8, 0, OUT, WIN>};
    if (p.P < 5 || p.P > 8) return hipErrorInvalidValue;
    return launch_tile_with<K>(kern[p.P], p, legacy, d_ws, st);
}

hipError_t launch_decode_tile(TileParams p, const DecParams &legacy, void *d_ws, hipStream_t st) {
    if (p.chunk_shift > 12 || p.chunk_shift < T_SUB_SHIFT) return hipErrorInvalidValue;
    if (p.o2) {                                  // order 2: first level of 6 bits per live context, two tiles per wave
        if (p.P != 6 || p.chunk_shift > 10 || !p.ctx2slot || !p.nslots) return hipErrorInvalidValue;
        void (*k2)(TileParams) = decode_tile_kernel<2, 6, 0, 2, 0, 0, 1>;
        return launch_tile_with<2, 1>(k2, p, legacy, d_ws, st);
    }
#ifdef MH_EXP_PROBES
    // Diagnostic builds only (make exp EXPFLAGS=-DMH_EXP_PROBES): the shipped library has none of these switches.
    // MH_TILE_PROBE=1: the second-level table has zero records, so every gather is answered by the bounds check — the
    // instruction stream and the waits stay, the trips to L2 go (results are wrong)
    if (const char *pr = getenv("MH_TILE_PROBE")) { p.probe = uint32_t(atoi(pr)); if (p.probe & 1) p.nsec = 0; }
    const char *e = getenv("MH_TILE_K");                          // tiles per wave
    const int k = e ? atoi(e) : 2;
    const char *eo = getenv("MH_TILE_OUT");
    const int o = eo ? atoi(eo) : 2;
    if (o == 1) return k == 1 ? launch_tile_k<1, 1>(p, legacy, d_ws, st) : launch_tile_k<2, 1>(p, legacy, d_ws, st);
    if (const char *es = getenv("MH_TILE_STAMP")) {
        if (atoi(es)) {
            void (*ks)(TileParams) = decode_tile_kernel<2, 7, 0, 2, 0, 1>;
            if (p.P != 7) return hipErrorInvalidValue;
            return launch_tile_with<2>(ks, p, legacy, d_ws, st);
        }
    }
    if (const char *eg = getenv("MH_TILE_G")) {                  // mh_tile_probes.hpp: other ways to the second level (P = 7 only)
        const int g = atoi(eg), w = getenv("MH_TILE_WIN") ? atoi(getenv("MH_TILE_WIN")) : 0;
        if (g && p.P != 7) return hipErrorInvalidValue;
        void (*kg)(TileParams) = nullptr;
#define MH_G_CASE(GV) case GV: kg = k == 1 ? (w ? decode_tile_kernel<1, 7, 0, 2, 1, 0, 0, GV> : decode_tile_kernel<1, 7, 0, 2, 0, 0, 0, GV>) \
                                           : (w ? decode_tile_kernel<2, 7, 0, 2, 1, 0, 0, GV> : decode_tile_kernel<2, 7, 0, 2, 0, 0, 0, GV>); break;
        switch (g) { MH_G_CASE(1) MH_G_CASE(2) MH_G_CASE(3) MH_G_CASE(4) MH_G_CASE(5) MH_G_CASE(6) MH_G_CASE(7) default: break; }
#undef MH_G_CASE
        if (kg) return k == 1 ? launch_tile_with<1>(kg, p, legacy, d_ws, st) : launch_tile_with<2>(kg, p, legacy, d_ws, st);
    }
    const char *ew = getenv("MH_TILE_WIN");                       // 1: the register-window variant
    if (ew && atoi(ew) == 1) return k == 4 ? launch_tile_k<4, 2, 1>(p, legacy, d_ws, st) : k == 3 ? launch_tile_k<3, 2, 1>(p, legacy, d_ws, st) : launch_tile_k<2, 2, 1>(p, legacy, d_ws, st);
    return k == 1 ? launch_tile_k<1, 2>(p, legacy, d_ws, st) : k == 4 ? launch_tile_k<4, 2>(p, legacy, d_ws, st) : k == 3 ? launch_tile_k<3, 2>(p, legacy, d_ws, st) : launch_tile_k<2, 2>(p, legacy, d_ws, st);
#else
    return launch_tile_k<2, 2>(p, legacy, d_ws, st);             // two tiles per wave, output through LDS, window from LDS
#endif
}

}  // namespace mhk
