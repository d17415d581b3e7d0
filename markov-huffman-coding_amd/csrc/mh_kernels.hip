// mh_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the Markov-Huffman hot path.
//
//   hist_o1_kernel     256x256 conditional histogram, LDS-resident packed counters (+ slab reduce)   (a1)
//   hist_o0_kernel     256-bin histogram                                                            (a2)
//   enc_len_kernel     code-length sum per 4 KiB wave-tile (LDS length table)                       (a9-a10)
//   scan_*             exclusive prefix over the wave-tile sums -> absolute bit offsets
//   enc_emit_kernel    LDS codeword table, wave prefix-sum of bit lengths (DPP), bits OR-ed into a
//                      wave-private LDS image, byte-swapped coalesced dword stores, seam atomics   (a9-a12)
//   decode_kernel      two-level decode tables (level 1 in LDS = the reference's 8-bit LUT), K chunks
//                      per lane, granule FIFO input, redo pass with the tree walk                  (a13-a15)
//   index_sync/fill    parallel index builder for streams that come without an index              (N1)
//
// (aN) = row of SURVEY.md §8(a).  All integer/bit work: no MFMA.  No CUDA idioms: waves are 64 wide,
// cross-lane traffic uses DPP / __shfl / __ballot on 64 lanes; workgroups never wait for each other.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "mh_kernels.h"
#include "mh_model.hpp"

namespace mhk {

using mh::DEC16_LEAF;
using mh::TREE_LEAF;
using mh::TREE_STRIDE;

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / row broadcasts (7 VALU
// instructions; the __shfl_up form costs six LDS-crossbar round trips).
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    // within each row of 16 lanes
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xF, 0xF, true));   // row_shr:1
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xF, 0xF, true));   // row_shr:2
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xF, 0xF, true));   // row_shr:4
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xF, 0xF, true));   // row_shr:8
    // carry row totals forward: lane 15 of a row into the next row, then lane 31 into rows 2 and 3
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xA, 0xF, false));  // row_bcast:15, rows 1 and 3
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xC, 0xF, false));  // row_bcast:31, rows 2 and 3
    return v;
}

// The 16 table slots (mh::enc_slot) of a lane's 16 consecutive input bytes, without forming the
// windows: per dword the mixed low bytes of its four symbols are bfi(0xF8F8F8F8, x << 3, x >> 5) ^
// (x << 8 | previous byte), and one byte shuffle per symbol pairs each with its symbol.
__device__ __forceinline__ void slots16(const uint4 &x4, uint32_t pb, uint32_t (&slot)[16]) {
    const uint32_t x[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t rot = (((x[k] << 3) & 0xF8F8F8F8u) | ((x[k] >> 5) & 0x07070707u));
        const uint32_t y = rot ^ ((x[k] << 8) | pb);
        pb = x[k] >> 24;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            slot[4 * k + j] = __builtin_amdgcn_perm(x[k], y, 0x0C0C0400u + uint32_t(j) * 0x0101u);   // x.byte j << 8 | y.byte j
    }
}

__global__ void set_word_kernel(uint32_t *p, uint32_t v) { *p = v; }
hipError_t launch_set_word(uint32_t *d_word, uint32_t v, hipStream_t st) {
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, st, d_word, v);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// histogram, order 1
// ------------------------------------------------------------------------------------------------
// 65536 counters do not fit LDS as u32 (256 KiB > 160 KiB), so two 16-bit fields share one LDS word, each
// a 14-bit counter under two guard bits (bits 14 and 15 of its half).  A returning ds_add tells a lane that
// its add carried a field across a multiple of 0x4000; that lane subtracts 0x4000 again and credits 16384 to
// the 64-bit counter in HBM — one fix-up per crossing, whichever lane caused it, so the field's value plus
// 16384 x (fix-ups done) is always the true count and late fix-ups only let the field run higher for a while.
// A field spills into its neighbour only at 0x10000, i.e. with FOUR crossings (49152 adds) still un-applied.
// What can be un-applied: a fix-up trails its add by two trips through the CU's LDS queue (the adding wave's
// own batch has to return, then its subtract queues up), and the queue holds at most 16 waves x 16
// outstanding instructions x 64 lanes = 16384 adds, so about 33 K adds to ONE counter in the worst case (a
// run of one repeated pair, e.g. zero pages, where every lane of the workgroup hits the same word).
// The first version had a single guard bit (room for 32768): tests/test_gpu_scale.py
// test_histogram_guard_bit_fixups_many_per_workgroup caught it losing 8 x 32768 counts on 64 MiB of zeros.
constexpr int HIST_THREADS = 1024;
constexpr int HIST_LDS_BYTES = 32768 * 4;

// Counter slot of a (prev, sym) pair: sym << 8 | (prev ^ mix(sym)).  Word = slot & 0x7FFF, half = bit 15
// (the symbol's top bit).  The low byte decides the LDS bank.  prev ^ sym alone piles the frequent pairs of
// small ranks (or of one ASCII block) onto a few banks; mixing sym << 3 in spreads them (simulated worst-bank
// load per 64-lane add: Zipf(1.1) 5.8 -> 4.8, text 6.4 -> 5.0, random would be 4.0; measured: 4 GiB text 1.61
// vs 1.69 ms, 16 GiB Zipf unchanged at 5.8).  Any function of sym keeps the mapping invertible; for four
// packed symbols this one costs four instructions.
__device__ __forceinline__ uint32_t hist_mix(uint32_t sym) { return (sym ^ (sym << 3)) & 255u; }
__device__ __forceinline__ uint32_t hist_slot(uint32_t prev, uint32_t sym) { return (sym << 8) | (prev ^ hist_mix(sym)); }
__device__ __forceinline__ uint32_t hist_slot_prev(uint32_t slot) { return (slot & 255u) ^ hist_mix(slot >> 8); }

// cross (region mode): the workgroup's list of crossings, [0] = count, then the slots — with the slab it gives
// the workgroup's own exact pair counts (field + 16384 per listed crossing), which is what lets the encoder
// price its region without a length pass (enc_region_kernel)
// GUARD = counter bits of a 16-bit field: 14 (two guard bits, the product) or 15 (one guard bit: round 1's
// version, which loses counts on runs of one pair — kept ONLY as MH_DEBUG_HIST_GUARD1=1, so that a test can watch
// the conservation check of hist_reduce_kernel catch a spill)
template <int GUARD>
__device__ __forceinline__ void hist_fixup(uint32_t *h, unsigned long long *counts, uint32_t slot, uint32_t *cross, uint32_t cross_cap) {
    atomicSub(&h[slot & 0x7FFFu], (slot >> 15) ? (GUARD < 16 ? (0x10000u << (GUARD & 15)) : 0u) : (1u << GUARD));
    atomicAdd(&counts[hist_slot_prev(slot) * 256u + (slot >> 8)], (unsigned long long)(1u << GUARD));
    if (cross) {
        const uint32_t i = atomicAdd(&cross[0], 1u);
        if (i < cross_cap) cross[1u + i] = slot;
    }
}

template <int GUARD>
__device__ __forceinline__ void hist_add(uint32_t *h, unsigned long long *counts, uint32_t prev, uint32_t sym, uint32_t *cross,
                                         uint32_t cross_cap) {
    constexpr uint32_t CROSS = ((0x10000u - (1u << GUARD)) & 0xFFFFu) * 0x10001u;   // 0xC000C000 for 14 bits
    const uint32_t slot = hist_slot(prev, sym);
    const uint32_t inc = (slot >> 15) ? 0x10000u : 1u;
    const uint32_t old = atomicAdd(&h[slot & 0x7FFFu], inc);
    if (((old + inc) ^ old) & CROSS) hist_fixup<GUARD>(h, counts, slot, cross, cross_cap);
}

// slab: when not null, every workgroup stores its 32768 LDS words there (plain coalesced stores) and
// hist_reduce_kernel sums the slabs afterwards; 16.7 M device-scope 64-bit atomics on the same 512 KiB
// of counters (256 workgroups x 65536) cost ~0.55 ms per call whatever the input size.
// region_vecs != 0 (region mode, needs the slab): workgroup w counts the CONTIGUOUS vectors
// [w * region_vecs, (w + 1) * region_vecs) instead of a grid-strided share, and lists its crossings in
// cross_all + w * (cross_cap + 1).
template <int GUARD>
__global__ __launch_bounds__(HIST_THREADS) void hist_o1_kernel(const uint8_t *__restrict__ data, uint64_t n,
                                                              uint32_t prev0, unsigned long long *counts, uint32_t *slab,
                                                              uint64_t region_vecs, uint32_t *cross_all, uint32_t cross_cap) {
    constexpr uint32_t CROSS = ((0x10000u - (1u << GUARD)) & 0xFFFFu) * 0x10001u;   // 0xC000C000 for 14 bits
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    for (int i = threadIdx.x; i < 32768 / 4; i += HIST_THREADS) reinterpret_cast<uint4 *>(h)[i] = make_uint4(0, 0, 0, 0);
    uint32_t *cross = cross_all ? cross_all + size_t(blockIdx.x) * (cross_cap + 1u) : nullptr;
    if (cross && threadIdx.x == 0) cross[0] = 0;
    __syncthreads();

    const uint64_t nvec = n >> 4;  // whole 16-byte vectors
    const uint4 *vdata = reinterpret_cast<const uint4 *>(data);
    const uint64_t v_begin = region_vecs ? uint64_t(blockIdx.x) * region_vecs + threadIdx.x : uint64_t(blockIdx.x) * HIST_THREADS + threadIdx.x;
    const uint64_t v_end = region_vecs ? ((blockIdx.x + 1ull) * region_vecs < nvec ? (blockIdx.x + 1ull) * region_vecs : nvec) : nvec;
    const uint64_t v_step = region_vecs ? uint64_t(HIST_THREADS) : uint64_t(gridDim.x) * HIST_THREADS;
    // the next trip's vector is loaded before this trip's adds; the byte in front of a lane's vector is the
    // previous lane's last byte (a lane shuffle) except in lane 0 of a wave, which loads it
    const bool lane0 = (threadIdx.x & 63u) == 0;
    uint4 nx4 = make_uint4(0, 0, 0, 0);
    uint32_t nhead = prev0;
    if (v_begin < v_end) {
        nx4 = vdata[v_begin];
        if (lane0 && v_begin) nhead = uint32_t(data[v_begin * 16 - 1]);
    }
    for (uint64_t v = v_begin; v < v_end; v += v_step) {
        const uint4 x4 = nx4;
        const uint32_t head = nhead;
        if (v + v_step < v_end) {
            nx4 = vdata[v + v_step];
#ifndef MH_HIST_PROBE_NOHEAD      /* diagnostic build (counts wrong): without the one-byte load in front of every wave's KiB — is it the 9 % of extra read requests? */
            // (this one-byte load in front of every wave's KiB is what the counters show as 4-9 % more read requests than the
            // input has lines: profiles/r04/hist_head_byte_*.txt — the line is the neighbouring wave's and gets fetched twice.
            // Giving every wave its own contiguous sixteenth of the region, so that the byte is a lane read, removed the
            // requests and cost 4 % in time (5.70 vs 5.48 ms per 16 GiB: sixteen streams per workgroup 4 MiB apart); not kept.)
            if (lane0) nhead = uint32_t(data[(v + v_step) * 16 - 1]);
#endif
        }
        const uint32_t up = __shfl_up(x4.w >> 24, 1);
        uint32_t pb = lane0 ? head : up;
        const uint32_t x[4] = {x4.x, x4.y, x4.z, x4.w};
        // The kernel is VALU-bound (measured: 13.6 instructions per symbol at 77 % VALU utilisation with
        // the previous slot hash), so the per-symbol work is kept to: one byte shuffle for the slot, the
        // word address, the half's increment, the atomic, and three instructions of overflow tracking.
        // All 16 returning adds are issued back to back; the rare fix-ups come afterwards.
        uint32_t slot[16], old[16], inc[16];
        uint32_t newly = 0;                                      // bits that one of this lane's adds flipped
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // prev ^ sym ^ (sym << 3) for the four symbols of the dword (hist_mix, bytewise)
            const uint32_t y = x[k] ^ ((x[k] << 3) & 0xF8F8F8F8u) ^ ((x[k] << 8) | pb);
            const uint32_t xm = x[k] & 0x7F7F7F7Fu;              // the symbols without their top bits: the shuffle below then yields the WORD index
            pb = x[k] >> 24;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * k + j;
                slot[i] = __builtin_amdgcn_perm(xm, y, 0x0C0C0400u + uint32_t(j) * 0x0101u);   // (x.byte j & 0x7F) << 8 | y.byte j: slot & 0x7FFF
                // 1, or 0x10000 for the upper half: top bit of the symbol * 0xFFFF + 1 (the compiler would turn
                // the multiply into compare + select, one instruction more)
                asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(inc[i]) : "v"(__builtin_amdgcn_ubfe(x[k], 8 * j + 7, 1)), "s"(0xFFFFu));
                old[i] = atomicAdd(&h[slot[i]], inc[i]);
            }
        }
        // newly |= (old + inc) ^ old, as one add and one three-input bit operation per symbol (left to itself the compiler
        // keeps all sixteen differences and ORs them three at a time: half an instruction more per symbol, sixteen registers)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t sum = old[i] + inc[i];
            asm("v_bitop3_b32 %0, %1, %2, %0 bitop3:0xbe" : "+v"(newly) : "v"(sum), "v"(old[i]));
        }
        if (newly & CROSS) {                                     // some add of this lane crossed a multiple of 0x4000
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (((old[i] + inc[i]) ^ old[i]) & CROSS)        // (the slot's bit 15, the symbol's top bit, says which half was added to)
                    hist_fixup<GUARD>(h, counts, slot[i] | (inc[i] != 1u ? 0x8000u : 0u), cross, cross_cap);
        }
    }
    // ragged tail (< 16 bytes): one lane of block 0 — in region mode of the workgroup whose region holds that vector
    const uint32_t tail_block = region_vecs ? uint32_t(nvec / region_vecs) : 0u;
    if (blockIdx.x == tail_block && threadIdx.x == 0) {
        uint64_t i = nvec << 4;
        uint32_t prev = i ? uint32_t(data[i - 1]) : prev0;
        for (; i < n; ++i) {
            uint32_t c = data[i];
            hist_add<GUARD>(h, counts, prev, c, cross, cross_cap);
            prev = c;
        }
    }
    __syncthreads();
    if (slab) {
        uint4 *dst = reinterpret_cast<uint4 *>(slab + size_t(blockIdx.x) * 32768u);
        for (uint32_t i = threadIdx.x; i < 32768u / 4u; i += HIST_THREADS) dst[i] = reinterpret_cast<const uint4 *>(h)[i];
        return;
    }
    // flush: one 64-bit atomic per non-zero counter
    for (uint32_t w = threadIdx.x; w < 32768u; w += HIST_THREADS) {
        uint32_t v = h[w];
        uint32_t lo = v & 0xFFFFu, hi = v >> 16;
        if (lo) atomicAdd(&counts[hist_slot_prev(w) * 256u + (w >> 8)], (unsigned long long)lo);
        if (hi) {
            const uint32_t s1 = w | 0x8000u;
            atomicAdd(&counts[hist_slot_prev(s1) * 256u + (s1 >> 8)], (unsigned long long)hi);
        }
    }
}

// Sums the workgroups' slabs into the 64-bit counters (which already hold the 16384-credits of counter
// overflows): thread w owns word w = two counters, reads are coalesced across the block.
// check (the workspace's first 64 bytes: [0] status, [2..3] running total, [4] ticket): the grand total of the counts
// must be the number of bytes counted (the reference's counts sum to the file size, src/main.cpp:176-178); a 16-bit
// field that spilled into its neighbour, a lost fix-up or a damaged slab all break that, and the last block to finish
// says so in the status word (mh_dev_status -> MH_ERR_CORRUPT).
__global__ __launch_bounds__(256) void hist_reduce_kernel(const uint32_t *__restrict__ slab, uint32_t nslab, unsigned long long *counts,
                                                          unsigned int *check, unsigned long long n) {
    __shared__ unsigned long long part[4];
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;          // < 32768
    unsigned long long lo = 0, hi = 0;
    for (uint32_t s = 0; s < nslab; ++s) {
        const uint32_t v = slab[size_t(s) * 32768u + w];
        lo += v & 0xFFFFu;
        hi += v >> 16;
    }
    const uint32_t s1 = w | 0x8000u;
    const unsigned long long c0 = counts[hist_slot_prev(w) * 256u + (w >> 8)] + lo;
    const unsigned long long c1 = counts[hist_slot_prev(s1) * 256u + (s1 >> 8)] + hi;
    counts[hist_slot_prev(w) * 256u + (w >> 8)] = c0;
    counts[hist_slot_prev(s1) * 256u + (s1 >> 8)] = c1;
    if (!check) return;
    unsigned long long t = c0 + c1;
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *total = reinterpret_cast<unsigned long long *>(check + 2);
        atomicAdd(total, part[0] + part[1] + part[2] + part[3]);
        __threadfence();
        if (atomicAdd(check + 4, 1u) == gridDim.x - 1u) {       // the last block: every block's share is in
            const unsigned long long all = atomicAdd(total, 0ull);
            if (all != n) atomicExch(reinterpret_cast<int *>(check), MHK_STATUS_CORRUPT);
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
constexpr int E_THREADS = 1024;
constexpr int E_WAVES = E_THREADS / 64;
constexpr int E_VEC = 16;                              // bytes per lane per sub-step
constexpr int E_SUB = 64 * E_VEC;                      // 1 KiB per wave sub-step
constexpr int E_SUBSTEPS = 4;
constexpr int E_WT = E_SUB * E_SUBSTEPS;               // 4 KiB wave-tile
constexpr int E_STAGE_BITS = E_SUB * mh::ENC16_MAX_LEN;            // 12288 payload bits per sub-step
constexpr int E_STAGE_WORDS = E_STAGE_BITS / 32 + 8;               // + alignment word + pad
constexpr int EMIT_LDS_BYTES = 131072 + E_WAVES * E_STAGE_WORDS * 4;
constexpr int LEN_LDS_BYTES = 65536;

// The lane's 16 bytes at `off` (zero past n), issued early so that the next sub-step's HBM latency
// hides behind the current one's work.  nvalid = bytes < n.  head = the byte before the vector, loaded
// by lane 0 only (the other lanes take their neighbour's last byte at use time).
struct LaneIn { uint4 x; uint32_t nvalid; uint32_t head; };

__device__ __forceinline__ LaneIn load_raw(const uint8_t *__restrict__ data, uint64_t n, uint64_t off, uint32_t prev0) {
    LaneIn r;
    r.x = make_uint4(0, 0, 0, 0);
    r.nvalid = 0;
    r.head = prev0;
    if (off + E_VEC <= n) {
        r.x = *reinterpret_cast<const uint4 *>(data + off);
        r.nvalid = E_VEC;
    } else if (off < n) {
        r.nvalid = uint32_t(n - off);
        uint32_t b[4] = {0, 0, 0, 0};
        for (uint32_t j = 0; j < r.nvalid; ++j) b[j >> 2] |= uint32_t(data[off + j]) << (8u * (j & 3u));
        r.x = make_uint4(b[0], b[1], b[2], b[3]);
    }
    if ((threadIdx.x & 63u) == 0 && off) r.head = (off - 1 < n) ? uint32_t(data[off - 1]) : 0u;
    return r;
}
// byte before the lane's vector: the previous lane's last byte, except in lane 0
__device__ __forceinline__ uint32_t head_byte(const LaneIn &in) {
    uint32_t up = __shfl_up(in.x.w >> 24, 1);
    return (threadIdx.x & 63u) == 0 ? in.head : up;
}
__device__ __forceinline__ void load_lane(const uint8_t *__restrict__ data, uint64_t n, uint64_t off, uint32_t prev0,
                                          uint4 &x, uint32_t &pb, uint32_t &nvalid) {
    LaneIn in = load_raw(data, n, off, prev0);
    x = in.x; nvalid = in.nvalid; pb = head_byte(in);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__device__ __forceinline__ uint32_t ctx_before(const uint8_t *__restrict__ data, uint64_t n, uint64_t off, uint32_t ctx0);
__device__ __forceinline__ uint32_t head_ctx(const LaneIn &in);
__device__ __forceinline__ LaneIn load_raw2(const uint8_t *__restrict__ data, uint64_t n, uint64_t off, uint32_t ctx0);
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

// ---- scan over wave-tile sums --------------------------------------------------------------------
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_PER_THREAD = 4;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_PER_THREAD;

__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t *lds, uint64_t &total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t t = __shfl_up(inc, d);
        if (lane >= uint32_t(d)) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (uint32_t w = 0; w < SCAN_THREADS / 64; ++w) {
        uint64_t a = lds[w];
        if (w < wave) base += a;
        tot += a;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
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
constexpr uint64_t HIST_WS_MAGIC = 0x4D48525247303031ull;                 // "MHRRG001"

struct HistHeader { unsigned long long magic, n, data, region_vecs; uint32_t grid, prev0, cross_cap, pad; };

__global__ void hist_header_kernel(HistHeader *hdr, HistHeader v) { *hdr = v; }

// one workgroup per region: bits = sum over pairs of (slab field + 16384 x crossings) x code length
__global__ __launch_bounds__(1024) void region_bits_kernel(const HistHeader *hdr, HistHeader expect, const uint32_t *slab,
                                                           const uint32_t *cross_all, const uint8_t *len8,
                                                           unsigned long long *region_bits, uint32_t *region_esc, int *status) {
    __shared__ unsigned long long part[16];
    __shared__ uint32_t any_esc;
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
        const uint32_t l0 = len8[hist_slot_prev(s0) * 256u + (s0 >> 8)], l1 = len8[hist_slot_prev(s1) * 256u + (s1 >> 8)];
        acc += (unsigned long long)(v & 0xFFFFu) * l0;
        acc += (unsigned long long)(v >> 16) * l1;
        esc |= ((v & 0xFFFFu) && l0 > uint32_t(mh::ENC16_MAX_LEN)) || ((v >> 16) && l1 > uint32_t(mh::ENC16_MAX_LEN));
    }
    const uint32_t *cross = cross_all + size_t(w) * (expect.cross_cap + 1u);
    const uint32_t nc = cross[0];
    if (nc > expect.cross_cap && tid == 0) atomicExch(status, MHK_STATUS_CAPACITY);
    for (uint32_t i = tid; i < (nc < expect.cross_cap ? nc : expect.cross_cap); i += 1024u) {
        const uint32_t sl2 = cross[1u + i];
        const uint32_t l2 = len8[hist_slot_prev(sl2) * 256u + (sl2 >> 8)];
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
template <bool ESCK>
__global__ __launch_bounds__(E_THREADS) void enc_region_kernel(EmitParams p, RegionParams rp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((rp.region_esc[blockIdx.x] != 0) != ESCK) return;
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
    const uint64_t rounds = (v1 - v0 + E_THREADS - 1) / E_THREADS;
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
        const uint64_t v = v0 + r * E_THREADS + tid;
        LaneIn in = load_raw(p.data, p.n, v < v1 ? v * E_VEC : ~0ull >> 1, p.prev0);   // beyond the region: nothing
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
        const uint64_t v = v0 + r * E_THREADS + tid;
        in.x = reinterpret_cast<const uint4 *>(p.data)[v];
        in.nvalid = E_VEC;
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
        const uint64_t off = (v0 + r * E_THREADS + tid) * E_VEC;
        const uint32_t nvalid = FULL ? uint32_t(E_VEC) : nvalid0;
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
    const uint64_t rounds_full = whole > v0 ? (whole - v0) / E_THREADS : 0;
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

// 64-bit window + one prefetched word: the load for the NEXT refill is always in flight, so a refill
// never waits on memory.
struct BitCursor {
    uint64_t buf;     // next bits, first at bit 63
    uint32_t cnt;     // valid bits in buf
    uint32_t ahead;   // stream word `next - 1`, already loaded
    uint64_t next;    // next word index to fetch
    __device__ __forceinline__ void init(const BitSrc &src, uint64_t bitpos) {
        uint64_t w = bitpos >> 5;
        uint32_t sh = uint32_t(bitpos & 31u);
        buf = ((uint64_t(src.word(w)) << 32) | src.word(w + 1)) << sh;
        cnt = 64u - sh;
        ahead = src.word(w + 2);
        next = w + 3;
    }
    // afterwards cnt >= 33
    __device__ __forceinline__ void refill(const BitSrc &src) {
        if (cnt <= 32u) {
            buf |= uint64_t(ahead) << (32u - cnt);
            cnt += 32u;
            ahead = src.word(next++);
        }
    }
    __device__ __forceinline__ void drop(uint32_t n) { buf <<= n; cnt -= n; }
    __device__ __forceinline__ uint64_t window() const { return buf; }
};

struct DecTables {
    const uint16_t *sec;         // second-level tables: LDS copy (decode_kernel) or global (index builder)
    const uint32_t *tree;        // last-resort walk (HBM/L2)
    uint32_t P;                  // primary width in bits
    uint32_t direct, H;          // uniform L2 tables: inner entry = table id, 2^H entries each
    __amdgpu_buffer_rsrc_t sec_rsrc;   // L2-resident second level: buffer resource over `sec` (a gather then takes a 32-bit
                                       // byte offset instead of a 64-bit address)
};

// Decodes one symbol (sequential index builder; tables read from global memory).  Returns the symbol,
// or 0 with *bad set on a null table entry (corrupt stream / context missing from the table).
// *used accumulates the bits consumed.
template <typename CUR>
__device__ __forceinline__ uint32_t decode_one(const uint16_t *prim, const uint32_t *sec_base, const DecTables &t,
                                               const BitSrc &src, CUR &bc, uint32_t prev, uint32_t &used, bool &bad) {
    bc.refill(src);                                        // >= 33 bits: enough for P + 8
    uint32_t e = prim[(prev << t.P) | uint32_t(bc.window() >> (64u - t.P))];
    if (e & DEC16_LEAF) {                                  // code of <= P bits (src/coding.cpp:150-156)
        uint32_t len = (e >> 8) & 31u;
        bad |= (len == 0);
        bc.drop(len); used += len;
        return e & 255u;
    }
    // longer code (src/coding.cpp:129-149): the node's own table, indexed by the next h bits
    const uint32_t h = t.direct ? t.H : ((e >> 12) & 7u) + 1u;
    bc.drop(t.P);
    const uint32_t tbase = t.direct ? (e << t.H) : sec_base[prev] + (e & 0xFFFu);
    uint32_t e2 = t.sec[tbase + uint32_t(bc.window() >> (64u - h))];
    if (e2 & DEC16_LEAF) {
        uint32_t len = (e2 >> 8) & 31u;                     // total length, P included
        bad |= (len == 0);
        if (len) { bc.drop(len - t.P); used += len; }
        return e2 & 255u;
    }
    // longer than P + h: walk the context's tree bit by bit from that node
    bc.drop(h);
    uint32_t node = e2 & 0x1FFu, n = t.P + h;
    const uint32_t *tr = t.tree + prev * TREE_STRIDE;
    for (int guard = 0; guard < 256; ++guard) {
        bc.refill(src);
        uint32_t bit = uint32_t(bc.window() >> 63);
        bc.drop(1); ++n;
        uint32_t pair = tr[node];
        uint32_t c = bit ? (pair >> 16) : (pair & 0xFFFFu);
        if (c & TREE_LEAF) { used += n; return c & 255u; }
        node = c;
    }
    bad = true;
    used += n;
    return 0;
}

// ---- the hot decoder ---------------------------------------------------------------------------
// One workgroup per CU, one lane per chunk, consecutive lanes on consecutive chunks.  Two measured
// facts shape the input side:
//  (1) a lane's compressed bytes are ~0.7 KiB away from its neighbour's, so a per-lane dword read costs
//      a whole cache line, and with ~1000 streams per CU the lines do not survive in L2 between two
//      reads (16x read amplification, L2 hit rate 32 %).  Each lane therefore pulls its stream in
//      32-byte aligned granules (two 16-byte loads): every byte is fetched once.
//  (2) vmcnt retires in order.  A load issued by SOME lane in a round sits in front of that round's
//      table gather for the WHOLE wave, so per-lane "refill when empty" loads put an HBM latency into
//      every round.  Loads are therefore issued only at block boundaries (every 16 symbols, all lanes
//      together; in the L2 layout right BEHIND the first step's gathers) and consumed a block later:
//      `nxt` (and with DEPTH 2 `pre`) is in flight or resident, `cur` feeds the bit window; inside a
//      block a granule switch is register-to-register.
// A block of 16 symbols decoded through the tables consumes at most 16 * 16 = 256 bits = one granule,
// and a block starts with `nxt` full, so the hot path never runs dry; longer codes (the walk) and the
// set-up use the checked pop.
// GW = dwords per granule: 8 (32 bytes) or 16 (64 bytes: half the read amplification, twice the
// registers).  A block is 2 * GW symbols (<= 16 bits each through the tables = one granule).
// DEPTH = granules a stream keeps beside `cur`: 2 (`nxt` resident + `pre` in flight: a granule is consumed
// one block after it was asked for, so the block-boundary loads never make a pop wait) or 1 (`nxt` alone,
// loaded straight into: eight registers per stream fewer; the switch cur <- nxt may then wait for a load
// that was issued at the last block boundary).  Measured at 16 GiB Zipf (profiles/r02/decode_variants.md):
// DEPTH 1 with 4, 5 or 6 streams per lane and 3 or 4 waves per SIMD all land within 2 % of, or behind,
// DEPTH 2 with 4 streams and 2 waves — the kernel is not short of streams in flight.  The L2 layout runs
// DEPTH 1: the eight registers per stream hold a 64-byte store burst instead (25.3 vs 29.4 ms).
template <int GW, int DEPTH = 2>
struct LaneStream {
    static constexpr int NQ = GW / 4;      // uint4 loads per granule
    static constexpr uint32_t BEHIND = DEPTH + 1;   // `cur` holds granule gnext - BEHIND while nxt is full
    const uint4 *base;    // payload (wave-uniform: lives in SGPRs)
    uint32_t glast;       // last readable granule of the payload (wave-uniform)
    uint32_t gnext;       // granule (counted from the payload start) that is loaded next
    uint32_t cur[GW];     // granule feeding the window, cur[0] is next
    uint32_t ccnt;        // dwords left in cur
    uint32_t nxt[GW];     // following granule (DEPTH 1: possibly still in flight)
    bool nxt_full;
    uint32_t pre[DEPTH == 2 ? GW : 1];     // DEPTH 2: the one after, possibly still in flight

    uint64_t buf;         // next bits, first at bit 63
    uint32_t cnt;         // valid bits in buf

    __device__ __forceinline__ void issue_into(uint32_t (&dst)[GW]) {
        const uint32_t g = gnext < glast ? gnext : glast;
        const uint4 *src = base + uint64_t(NQ) * g;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint4 a = src[q];
            dst[4 * q] = a.x; dst[4 * q + 1] = a.y; dst[4 * q + 2] = a.z; dst[4 * q + 3] = a.w;
        }
        ++gnext;
    }
    __device__ __forceinline__ void issue_pre() {
        if constexpr (DEPTH == 2) issue_into(pre); else issue_into(nxt);
    }
    // wave-synchronous point (block boundary): the only place where loads are issued and awaited
    __device__ __forceinline__ void block_sync() {
        if (!nxt_full) {
            if constexpr (DEPTH == 2) {
#pragma unroll
                for (int i = 0; i < GW; ++i) nxt[i] = pre[i];
            }
            nxt_full = true;
            issue_pre();
        }
    }
    // CHECKED (set-up, walk of over-long codes, tail chunks): keeps `nxt` full around every pop, so any
    // amount may be consumed.  Unchecked (hot path): relies on the per-block budget above.
    // The granule is consumed a quad at a time: 3 moves per pop, GW - 4 more every fourth pop.
    template <bool CHECKED>
    __device__ __forceinline__ uint32_t pop_word() {
        if (CHECKED) block_sync();
        const uint32_t w = cur[0];
        if (GW == 8) {
            // short granule: shifting all of it costs less than a second level of bookkeeping
#pragma unroll
            for (int i = 0; i < GW - 1; ++i) cur[i] = cur[i + 1];
            if (--ccnt == 0) {
#pragma unroll
                for (int i = 0; i < GW; ++i) cur[i] = nxt[i];
                nxt_full = false;
                ccnt = GW;
                if (CHECKED) block_sync();
            }
            return w;
        }

        cur[0] = cur[1]; cur[1] = cur[2]; cur[2] = cur[3];
        --ccnt;
        if ((ccnt & 3u) == 0u) {
            if (ccnt == 0) {
#pragma unroll
                for (int i = 0; i < GW; ++i) cur[i] = nxt[i];
                ccnt = GW;
                nxt_full = false;
                if (CHECKED) block_sync();
            } else {
#pragma unroll
                for (int i = 0; i < GW - 4; ++i) cur[i] = cur[i + 4];
            }
        }
        return w;
    }
    // total_bytes > 0 and bitpos < 8 * total_bytes (checked by the caller)
    __device__ __forceinline__ void init(const uint8_t *payload, uint64_t total_bytes, uint64_t bitpos) {
        const uint64_t w = bitpos >> 5;                      // first stream dword
        base = reinterpret_cast<const uint4 *>(payload);
        glast = uint32_t((total_bytes - 1) / (GW * 4));       // payloads stay below 2^32 granules (128 GiB)
        gnext = uint32_t(w / GW);
        if constexpr (DEPTH == 2) {
            issue_into(pre);
#pragma unroll
            for (int i = 0; i < GW; ++i) cur[i] = pre[i];
            ccnt = GW;
            issue_into(pre);
#pragma unroll
            for (int i = 0; i < GW; ++i) nxt[i] = pre[i];
            nxt_full = true;
            issue_into(pre);
        } else {
            issue_into(nxt);
#pragma unroll
            for (int i = 0; i < GW; ++i) cur[i] = nxt[i];
            ccnt = GW;
            issue_into(nxt);
            nxt_full = true;
        }
        for (uint32_t skip = uint32_t(w % GW); skip; --skip) (void)pop_word<true>();
        const uint32_t hi = __builtin_bswap32(pop_word<true>());
        const uint32_t lo = __builtin_bswap32(pop_word<true>());
        const uint32_t sh = uint32_t(bitpos & 31u);
        buf = ((uint64_t(hi) << 32) | lo) << sh;
        cnt = 64u - sh;
    }
    template <bool CHECKED>
    __device__ __forceinline__ void refill() {               // afterwards cnt >= 33
        if (cnt <= 32u) {
            buf |= uint64_t(__builtin_bswap32(pop_word<CHECKED>())) << (32u - cnt);
            cnt += 32u;
        }
    }
    // (The same refill as selects instead of masked regions — some lane needs the pop at practically every
    //  refill point, so the regions run anyway — takes 21 % of the loop's instructions and all but 6 of
    //  its 52 branches away and measured SLOWER: 26.6 vs 25.7 ms.  The kernel is not bound by what it issues.)
    // position of the next unread bit in the payload, modulo 2^32: `cur` holds granule gnext - BEHIND
    // (one more while `nxt` is empty), GW - ccnt of its dwords have gone into the window, cnt bits of
    // the window are still unread
    __device__ __forceinline__ uint32_t position() const {
        const uint32_t gran = gnext - (nxt_full ? BEHIND : BEHIND - 1u);
        return (gran * GW + (GW - ccnt)) * 32u - cnt;
    }
};

// BitCursor's face over the 32-byte-granule FIFO of the order-1 decoder: a lane's payload arrives in aligned
// 32-byte pieces (a dword at a time costs a 128-byte line per load once the lanes' streams are a chunk apart)
struct GranuleCursor {
    LaneStream<8, 2> ls;
    __device__ __forceinline__ void init(const BitSrc &src, uint64_t bitpos) { ls.init(src.p, src.bytes, bitpos); }
    __device__ __forceinline__ void refill(const BitSrc &) { ls.template refill<true>(); }
    __device__ __forceinline__ void drop(uint32_t n) { ls.buf <<= n; ls.cnt -= n; }
    __device__ __forceinline__ uint64_t window() const { return ls.buf; }
};

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

// SEC_LDS  both table levels in LDS (else the second level is gathered from L2)
// SPR      symbols per window refill (2, or 4 when no code exceeds 8 bits)
// DIRECT   L2 mode with uniform, directly addressed second-level tables
// K        independent streams (chunks) per lane
// GW       dwords per input granule (8 = 32 B, 16 = 64 B)
// OUTB     16-byte stores per output burst (1 = 16 B, 4 = 64 B contiguous per stream)
// Models whose tables live in LDS and whose codes are all <= 8 bits are bound by how the streams touch HBM
// (measured: 32-byte granules re-fetch every 128-byte line four times, 16-byte stores double the write
// traffic), so they run K = 2 with 64-byte granules and 64-byte store bursts; with a second level the
// dependent lookups dominate and K = 4 with 32-byte granules wins (32-byte bursts with both levels in LDS,
// 64-byte bursts on a two-slot FIFO with the second level in L2: see launch_decode).
template <bool SEC_LDS, int SPR, bool DIRECT, int K, int GW, int OUTB, int PC, int HC, bool REDO = false, int NT = 512, int DEPTH = 2>
__global__ __launch_bounds__(NT) void decode_kernel(DecParams p) {
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
                                                 uint64_t seg_end, uint32_t &count, bool &bad, F on_symbol) {
    uint64_t pos = st_pos(p, start);
    uint32_t prev = st_ctx(p, start);
    count = 0;
    bad = false;
    if (pos >= seg_end) return start;
    GranuleCursor bc;                                       // (dword reads cost a cache line each here: lanes are a segment apart)
    bc.init(src, pos);
    while (pos < seg_end) {
        on_symbol(count, prev, pos);
        uint32_t used = 0;
        uint32_t sym = decode_one(p.prim, p.sec_base, tabs, src, bc, prev, used, bad);
        if (bad) break;
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
    uint32_t count_sym;
    bool bad;
    uint64_t end = walk_segment(p, tabs, src, start, seg_end, count_sym, bad, [](uint32_t, uint32_t, uint64_t) {});
    if (bad) end = st_make(p, st_ctx(p, end), seg_end);                // (the fill pass reports it if the state was the true one)
    const uint64_t over = st_pos(p, end) - seg_end;
    __hip_atomic_store(&p.e16[i], uint16_t((st_ctx(p, end) << 8) | uint32_t(over > 254 ? 254 : over)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p.c16[i] = uint16_t(count_sym);
    p.s16[i] = uint16_t(pe == IX_INVALID ? 0xFFFEu : pe);              // (an end state is never IX_INVALID: its overshoot is under 255)
}

// symbols per tile of IX_TILE_SEGS segments (the input of the prefix sum): one wave per tile
__global__ __launch_bounds__(256) void index_tile_count_kernel(IdxParams p) {
    const uint64_t t = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (t >= p.ntile5) return;
    uint32_t sum = 0;
    const uint64_t s0 = t * IX_TILE_SEGS;
    for (uint32_t j = lane; j < IX_TILE_SEGS; j += 64u) if (s0 + j < p.nseg5) sum += p.c16[s0 + j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    if (lane == 0) p.tile_cnt[t] = sum;
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

// ------------------------------------------------------------------------------------------------
// ORDER 2 — context = the previous TWO bytes (BASELINE config 5).  An extension the
// reference only speculates about (README.md:158-166): PARITY UNPINNED, the spec is the generalised
// oracle (oracle/mh_oracle.h, order-2 section).  65536 contexts x 256 symbols = 16.7 M counters and
// codewords: nothing of that fits LDS, so the tables live in HBM and are served from L2 / the Infinity
// Cache (text-like sources touch a few thousand contexts: a few MiB of hot table).
//   hist_o2_kernel    LDS-resident tagged counter cache in front of 64-bit global atomics
//   enc2_len_kernel   code-length sum per 4 KiB wave-tile, lengths gathered from HBM/L2
//   enc2_emit_kernel  the emit loop of enc_emit_kernel with every codeword gathered from the full tables
//   decode2_kernel    one lane per chunk, both table levels + the walk tree gathered from HBM/L2
// Context convention: ctx = (byte before previous) << 8 | previous byte; both are ' ' before the stream.
// Index entries carry the 16-bit context in bits 48..63 (the bit offset keeps 48 bits).
// ------------------------------------------------------------------------------------------------
constexpr uint64_t IDX2_POS = 0x0000FFFFFFFFFFFFull;

// ctx of the byte at `off` for lane 0 of a wave (off is a multiple of 16): the two bytes before it
__device__ __forceinline__ uint32_t ctx_before(const uint8_t *__restrict__ data, uint64_t n, uint64_t off, uint32_t ctx0) {
    if (off == 0 || off > n) return ctx0;
    return (uint32_t(data[off - 2]) << 8) | uint32_t(data[off - 1]);
}
// the lane's vector + the context in front of it (previous lane's last two bytes, except in lane 0)
__device__ __forceinline__ uint32_t head_ctx(const LaneIn &in) {
    const uint32_t v = __shfl_up(in.x.w >> 16, 1);               // byte 14 | byte 15 << 8 of the previous lane
    const uint32_t up = ((v & 255u) << 8) | (v >> 8);
    return (threadIdx.x & 63u) == 0 ? in.head : up;
}
__device__ __forceinline__ LaneIn load_raw2(const uint8_t *__restrict__ data, uint64_t n, uint64_t off, uint32_t ctx0) {
    LaneIn r;
    r.x = make_uint4(0, 0, 0, 0); r.nvalid = 0; r.head = ctx0;
    if (off + E_VEC <= n) {
        r.x = *reinterpret_cast<const uint4 *>(data + off);
        r.nvalid = E_VEC;
    } else if (off < n) {
        r.nvalid = uint32_t(n - off);
        uint32_t b[4] = {0, 0, 0, 0};
        for (uint32_t j = 0; j < r.nvalid; ++j) b[j >> 2] |= uint32_t(data[off + j]) << (8u * (j & 3u));
        r.x = make_uint4(b[0], b[1], b[2], b[3]);
    }
    if ((threadIdx.x & 63u) == 0) r.head = ctx_before(data, n, off, ctx0);
    return r;
}

// ---- histogram: counts[ctx * 256 + sym] (64-bit, HBM).  A workgroup keeps 16384 (key, count) slots in LDS,
// an open-addressed table with linear probing: the first key to claim a slot owns it for the whole launch
// (tags never change once set, so a claim is one compare-and-swap and there is no eviction race); every
// later occurrence of that key is one LDS add.  A key that finds no slot within its probe limit goes
// straight to a 64-bit global atomic.  The probing matters more than it looks: text-like sources have a
// few thousand live keys, and ONE frequent key that loses its slot to an earlier one sends ~1 % of the
// stream to a single HBM address, where memory-side atomics serialise (first version, no probing: 83 ms
// per 4 GiB of text with 1.5 % of the symbols on 31 such addresses).  Flat sources (millions of live keys)
// fill the table at once; from 3/4 occupancy on a key gets two probes, so the misses stay cheap and the
// kernel degrades to the global-atomic rate over many addresses.
constexpr int H2_THREADS = 1024;
constexpr uint32_t H2_SLOTS = 16384;
constexpr uint32_t H2_EMPTY = 0xFFFFFFFFu;
constexpr int H2_LDS_BYTES = int(H2_SLOTS) * 8 + 64 * 4 + 16;     // tags, counters + one dummy word per lane, the claim counter
constexpr uint32_t H2_PROBES = 8, H2_PROBES_FULL = 2, H2_FULL = H2_SLOTS * 3 / 4;

// the two slots a key may own without probing: 14 bits each of one 32-bit product
__device__ __forceinline__ void hist2_slots(uint32_t key, uint32_t &s1, uint32_t &s2) {
    const uint32_t h = key * 0x9E3779B1u;
    s1 = h >> 18;
    s2 = (h >> 4) & (H2_SLOTS - 1u);
}
// the whole story for one key: its first slot, its second, then linear probing behind the first, then memory
__device__ __forceinline__ void hist2_add(uint32_t *tag, uint32_t *cnt, uint32_t *used, unsigned long long *counts, uint32_t key,
                                          uint32_t probes) {
    uint32_t s1, s2;
    hist2_slots(key, s1, s2);
    uint32_t slot = s1;
    for (uint32_t p = 0; p < probes + 1u; ++p) {
        // a plain read first: once its tag is set (tags never change) a key costs one read, which the LDS broadcasts
        // to all the lanes that ask for the same slot, and one add — not a compare-and-swap that serialises them
        uint32_t t = __hip_atomic_load(&tag[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (t == H2_EMPTY) {
            t = atomicCAS(&tag[slot], H2_EMPTY, key);
            if (t == H2_EMPTY) { atomicAdd(used, 1u); t = key; }
        }
        if (t == key) { atomicAdd(&cnt[slot], 1u); return; }
        slot = p == 0 ? s2 : ((p == 1 ? s1 : slot) + 1u) & (H2_SLOTS - 1u);
    }
    atomicAdd(&counts[key], 1ull);
}

__global__ __launch_bounds__(H2_THREADS) void hist_o2_kernel(const uint8_t *__restrict__ data, uint64_t n, uint32_t ctx0,
                                                             unsigned long long *counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tag = reinterpret_cast<uint32_t *>(smem);
    uint32_t *cnt = tag + H2_SLOTS;                              // (+ 64 dummy words behind the slots)
    uint32_t *used = cnt + H2_SLOTS + 64;                        // slots claimed so far
    for (uint32_t i = threadIdx.x; i < H2_SLOTS; i += H2_THREADS) { tag[i] = H2_EMPTY; cnt[i] = 0; }
    if (threadIdx.x == 0) *used = 0;
    __syncthreads();
    const uint64_t nvec = n >> 4;
    const uint4 *vdata = reinterpret_cast<const uint4 *>(data);
    // whole waves stay in the loop together (the neighbour's bytes come by shuffle)
    const uint64_t nvec_up = (nvec + 63) & ~uint64_t(63);
    const uint64_t vstep = uint64_t(gridDim.x) * H2_THREADS;
    const uint32_t dummy = H2_SLOTS + (threadIdx.x & 63u);       // where a lane's add goes when the slot it read is not its key's
    uint64_t v = uint64_t(blockIdx.x) * H2_THREADS + threadIdx.x;
    uint4 ahead = v < nvec ? vdata[v] : make_uint4(0, 0, 0, 0);
    for (; v < nvec_up; v += vstep) {
        const bool live = v < nvec;
        const uint4 x4 = ahead;
        if (v + vstep < nvec) ahead = vdata[v + vstep];
        const uint32_t up = __shfl_up(x4.w >> 16, 1);
        uint32_t ctx = ((up & 255u) << 8) | (up >> 8);
        if ((threadIdx.x & 63u) == 0) ctx = live ? ctx_before(data, n, v << 4, ctx0) : ctx0;
        if (!live) continue;
        const uint32_t ctx_in = ctx;
        const uint32_t x[4] = {x4.x, x4.y, x4.z, x4.w};
        // The usual case without a branch: all sixteen tags are read at once, a key that finds its own tag adds to its slot,
        // any other adds to a dummy word of the lane.  (The counters said what the symbol-by-symbol form below was waiting
        // for: 30 scalar instructions and 8.5 branches per symbol, the exec-mask bookkeeping of sixteen divergent probe
        // loops in a row, with the LDS 25 % and the vector ALU 29 % busy.)  Keys that miss — every key once per workgroup,
        // and what the table cannot hold — go through hist2_add afterwards.
        uint32_t key[16], sa[16], sb[16], ta[16], tb[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            key[j] = (ctx << 8) | ((x[j >> 2] >> (8 * (j & 3))) & 255u);
            ctx = key[j] & 0xFFFFu;
            hist2_slots(key[j], sa[j], sb[j]);
            ta[j] = __hip_atomic_load(&tag[sa[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            tb[j] = __hip_atomic_load(&tag[sb[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        uint32_t missed = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool ha = ta[j] == key[j], hb = tb[j] == key[j];
            atomicAdd(&cnt[ha ? sa[j] : hb ? sb[j] : dummy], 1u);
            missed |= (ha || hb) ? 0u : (1u << j);
        }
        if (missed) {                                            // (divergent, rare once the table is warm)
            const uint32_t probes = *used < H2_FULL ? H2_PROBES : H2_PROBES_FULL;
            // every lane takes ITS next missed symbol per trip: as many trips as the worst lane has misses, not sixteen
            const uint64_t xlo = uint64_t(x4.x) | uint64_t(x4.y) << 32, xhi = uint64_t(x4.z) | uint64_t(x4.w) << 32;
            while (missed) {
                const uint32_t j = uint32_t(__builtin_ctz(missed));
                missed &= missed - 1u;
                // bytes j - 2, j - 1, j of the lane's stream: the incoming context supplies what lies before byte 0
                const uint32_t b0 = uint32_t(((j < 8u ? xlo : xhi) >> (8u * (j & 7u))) & 255u);
                const uint32_t j1 = j - 1u, j2 = j - 2u;
                const uint32_t b1 = j >= 1u ? uint32_t(((j1 < 8u ? xlo : xhi) >> (8u * (j1 & 7u))) & 255u) : (ctx_in & 255u);
                const uint32_t b2 = j >= 2u ? uint32_t(((j2 < 8u ? xlo : xhi) >> (8u * (j2 & 7u))) & 255u) : j == 1u ? (ctx_in & 255u) : (ctx_in >> 8);
                hist2_add(tag, cnt, used, counts, (b2 << 16) | (b1 << 8) | b0, probes);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {                   // ragged tail (< 16 bytes)
        uint64_t i = nvec << 4;
        uint32_t ctx = ctx_before(data, n, i, ctx0);
        for (; i < n; ++i) {
            const uint32_t key = (ctx << 8) | data[i];
            atomicAdd(&counts[key], 1ull);
            ctx = key & 0xFFFFu;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < H2_SLOTS; i += H2_THREADS)
        if (tag[i] != H2_EMPTY && cnt[i]) atomicAdd(&counts[tag[i]], (unsigned long long)cnt[i]);
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

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// Per-device launch state.  hipFuncSetAttribute (dynamic LDS above 64 KiB) is a per-device setting and the
// CU count differs between devices, so both are keyed by the device current at the call; a mutex makes
// the first call on a device safe from several threads (include/mh.h: models are thread-shareable and
// mh_set_device() may switch devices inside one process).
constexpr int MAX_DEVICES = 64;
struct DeviceState {
    int cu_count = 0;
    bool hist_ready = false, hist2_ready = false, encode_ready = false, chain_ready = false, region_ready = false, decode_ready = false, redo_ready = false, index_ready = false;
};
static DeviceState g_dev[MAX_DEVICES];
static std::mutex g_dev_mu;

static DeviceState &device_state() {          // caller holds g_dev_mu
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    DeviceState &d = g_dev[dev];
    if (d.cu_count == 0) {
        hipDeviceProp_t prop;
        d.cu_count = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return d;
}

static int cu_count() {
    std::lock_guard<std::mutex> lock(g_dev_mu);
    return device_state().cu_count;
}

static hipError_t allow_lds(const void *fn, int bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// Runs `setup` once per device (under the lock); `flag` selects the DeviceState member.
template <typename F>
static hipError_t once_per_device(bool DeviceState::*flag, F setup) {
    std::lock_guard<std::mutex> lock(g_dev_mu);
    DeviceState &d = device_state();
    if (d.*flag) return hipSuccess;
    hipError_t e = setup();
    if (e == hipSuccess) d.*flag = true;
    return e;
}

// Region mode of the order-1 histogram (and of the encoder that follows it): the input's 16-byte vectors are
// split into `grid` contiguous regions of region_vecs vectors (a multiple of 1024 = one encoder round).
// Workspace: [0,256) header | slabs grid x 32768 u32 | crossing lists grid x (1 + cross_cap) u32.
struct RegionGeom { int grid; uint64_t region_vecs, nvec_up; uint32_t cross_cap; size_t off_slab, off_cross, total; };
static RegionGeom region_geom(uint64_t n) {
    RegionGeom g;
    g.nvec_up = (n + 15) >> 4;
    const uint64_t want = (g.nvec_up + HIST_THREADS - 1) / HIST_THREADS;
    const uint64_t cus = uint64_t(cu_count());
    g.grid = int(want < 1 ? 1 : (want > cus ? cus : want));
    const uint64_t per = (g.nvec_up + uint64_t(g.grid) - 1) / uint64_t(g.grid);
    g.region_vecs = ((per + HIST_THREADS - 1) / HIST_THREADS) * HIST_THREADS;
    if (g.region_vecs == 0) g.region_vecs = HIST_THREADS;
    g.cross_cap = uint32_t(g.region_vecs * 16 / 16384) + 16u;       // a crossing takes 16384 adds of the workgroup
    g.off_slab = 256;
    g.off_cross = g.off_slab + size_t(g.grid) * 32768u * 4u;
    g.total = (g.off_cross + size_t(g.grid) * (g.cross_cap + 1u) * 4u + 255) & ~size_t(255);
    return g;
}
size_t hist_workspace_bytes(uint64_t n) { return region_geom(n).total; }

hipError_t launch_hist_o1(const uint8_t *d_data, uint64_t n, uint32_t prev0, unsigned long long *d_counts, void *d_ws, size_t ws_bytes,
                          hipStream_t st) {
    hipError_t e = hipMemsetAsync(d_counts, 0, 65536 * sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    unsigned char *ws = static_cast<unsigned char *>(d_ws);
    const bool ws_ok = d_ws && (reinterpret_cast<uintptr_t>(d_ws) & 15u) == 0;
    const RegionGeom g = region_geom(n);
    const bool regions = ws_ok && ws_bytes >= g.total;
    // debug (tests of the conservation check): MH_DEBUG_HIST_GUARD_BITS=1 one guard bit (round 1's kernel: can lose counts
    // on long runs of one pair, depending on timing), =0 none (a field that wraps carries into its neighbour: always does)
#ifdef MH_EXP_PROBES                         /* diagnostic library only (libmhc_diag.so): the shipped one always has two guard bits */
    const char *dbg = getenv("MH_DEBUG_HIST_GUARD_BITS");
    const int guard_bits = dbg ? atoi(dbg) : 2;
#else
    const int guard_bits = 2;
#endif
    const bool guard1 = guard_bits == 0 || guard_bits == 1;
    // workspace: [0,64) status block (status word, the conservation check's total and ticket) | [64,256) header
    if (ws_ok && ws_bytes >= 256) {
        e = hipMemsetAsync(ws, 0, 128, st);                     // status OK; no longer the histogram of anything
        if (e != hipSuccess) return e;
    }
    unsigned int *check = (ws_ok && ws_bytes >= 256) ? reinterpret_cast<unsigned int *>(ws) : nullptr;
    if (regions && !guard1) {   // the header says whose histogram the workspace holds (the region encoder checks it)
        HistHeader h{HIST_WS_MAGIC, n, reinterpret_cast<unsigned long long>(d_data), g.region_vecs, uint32_t(g.grid), prev0, g.cross_cap, 0};
        hipLaunchKernelGGL(hist_header_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<HistHeader *>(ws + 64), h);
    }
    if (n == 0) {
        if (regions) { e = hipMemsetAsync(ws + g.off_cross, 0, size_t(g.grid) * (g.cross_cap + 1u) * 4u, st); if (e != hipSuccess) return e;
                       e = hipMemsetAsync(ws + g.off_slab, 0, size_t(g.grid) * 32768u * 4u, st); }
        return e;
    }
    e = once_per_device(&DeviceState::hist_ready, [] {
        hipError_t r = allow_lds(reinterpret_cast<const void *>(hist_o1_kernel<14>), HIST_LDS_BYTES);
#ifdef MH_EXP_PROBES
        if (r == hipSuccess) r = allow_lds(reinterpret_cast<const void *>(hist_o1_kernel<15>), HIST_LDS_BYTES);
        if (r == hipSuccess) r = allow_lds(reinterpret_cast<const void *>(hist_o1_kernel<16>), HIST_LDS_BYTES);
#endif
        return r;
    });
    if (e != hipSuccess) return e;
#ifdef MH_EXP_PROBES
    auto kern = guard_bits == 0 ? hist_o1_kernel<16> : guard_bits == 1 ? hist_o1_kernel<15> : hist_o1_kernel<14>;
#else
    auto kern = hist_o1_kernel<14>;
#endif
    if (regions) {
        uint32_t *slab = reinterpret_cast<uint32_t *>(ws + g.off_slab);
        hipLaunchKernelGGL(kern, dim3(g.grid), dim3(HIST_THREADS), HIST_LDS_BYTES, st, d_data, n, prev0, d_counts, slab,
                           g.region_vecs, reinterpret_cast<uint32_t *>(ws + g.off_cross), g.cross_cap);
        hipLaunchKernelGGL(hist_reduce_kernel, dim3(32768 / 256), dim3(256), 0, st, slab, uint32_t(g.grid), d_counts, check,
                           (unsigned long long)(n));
        return hipGetLastError();
    }
    uint64_t nvec = n >> 4;
    uint64_t want = (nvec + HIST_THREADS - 1) / HIST_THREADS;
    int grid = int(want < 1 ? 1 : (want > uint64_t(cu_count()) ? uint64_t(cu_count()) : want));
    // with a (smaller) workspace the workgroups' counters go out as plain stores and are summed by a second kernel
    uint32_t *slab = (ws_ok && ws_bytes >= 256 + size_t(grid) * 32768u * 4u) ? reinterpret_cast<uint32_t *>(ws + 256) : nullptr;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(HIST_THREADS), HIST_LDS_BYTES, st, d_data, n, prev0, d_counts, slab,
                       uint64_t(0), static_cast<uint32_t *>(nullptr), 0u);
    if (slab) hipLaunchKernelGGL(hist_reduce_kernel, dim3(32768 / 256), dim3(256), 0, st, slab, uint32_t(grid), d_counts, check,
                                 (unsigned long long)(n));
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

hipError_t launch_hist_o2(const uint8_t *d_data, uint64_t n, uint32_t ctx0, unsigned long long *d_counts, hipStream_t st) {
    hipError_t e = hipMemsetAsync(d_counts, 0, (size_t(1) << 24) * sizeof(unsigned long long), st);
    if (e != hipSuccess || n == 0) return e;
    e = once_per_device(&DeviceState::hist2_ready, [] { return allow_lds(reinterpret_cast<const void *>(hist_o2_kernel), H2_LDS_BYTES); });
    if (e != hipSuccess) return e;
    const uint64_t nvec = n >> 4;
    const uint64_t want = (nvec + H2_THREADS - 1) / H2_THREADS;
    const int grid = int(want < 1 ? 1 : (want > uint64_t(cu_count()) ? uint64_t(cu_count()) : want));
    hipLaunchKernelGGL(hist_o2_kernel, dim3(grid), dim3(H2_THREADS), H2_LDS_BYTES, st, d_data, n, ctx0, d_counts);
    return hipGetLastError();
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
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<uint32_t *>(ws + 8), uint32_t(ENC_PATH_LENGTH_PASS));
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
        hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<uint32_t *>(ws + 8), uint32_t(ENC_PATH_CHAIN));
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
        return r != hipSuccess ? r : allow_lds(reinterpret_cast<const void *>(enc_region_kernel<true>), REGION_LDS_BYTES);
    });
    if (e != hipSuccess) return e;
    const unsigned char *hws = static_cast<const unsigned char *>(d_hist_ws);
    unsigned long long *region_bits = reinterpret_cast<unsigned long long *>(ws + 64);
    unsigned long long *region_start = region_bits + 1024;
    uint32_t *region_esc = reinterpret_cast<uint32_t *>(region_start + 1024);
    const HistHeader expect{HIST_WS_MAGIC, a.n, reinterpret_cast<unsigned long long>(a.data), g.region_vecs, uint32_t(g.grid), a.prev0, g.cross_cap, 0};
    hipLaunchKernelGGL(region_bits_kernel, dim3(g.grid), dim3(1024), 0, st, reinterpret_cast<const HistHeader *>(hws + 64), expect,
                       reinterpret_cast<const uint32_t *>(hws + g.off_slab), reinterpret_cast<const uint32_t *>(hws + g.off_cross), a.len8,
                       region_bits, region_esc, status);
    hipLaunchKernelGGL(region_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, region_bits, uint32_t(g.grid), region_start, a.start_bit,
                       a.out, a.cap & ~uint64_t(3), a.nbits, status);
    EmitParams ep{a.data, a.n, a.prev0, a.chunk_shift, a.out, a.enc16, a.len8, a.code64, nullptr, nullptr, 0, a.index, status, a.fine, nullptr, 0};
    RegionParams rp{region_start, region_bits, region_esc, g.region_vecs, g.nvec_up, (a.cap & ~uint64_t(3)) >> 2, status};
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<uint32_t *>(ws + 8),
                       uint32_t(a.max_len > mh::ENC16_MAX_LEN ? ENC_PATH_REGIONS_ESCAPES : ENC_PATH_REGIONS));
    hipLaunchKernelGGL(enc_region_kernel<false>, dim3(g.grid), dim3(E_THREADS), REGION_LDS_BYTES, st, ep, rp);
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
hipError_t launch_decode_redo(DecParams p, hipStream_t st) {
    if (p.order == 2) {                                      // order 2: the one-lane-per-chunk decoder over the list
        p.redo_list = 1;
        const uint64_t want2 = (p.nchunks + 255) / 256;
        const uint64_t cap2 = uint64_t(cu_count()) * 8;
        hipLaunchKernelGGL(decode2_kernel, dim3(unsigned(want2 > cap2 ? cap2 : (want2 < 1 ? 1 : want2))), dim3(256), 0, st, p);
        return hipGetLastError();
    }
    auto r_lds = decode_kernel<true, 2, false, 1, 8, 1, 0, 0, true>;
    auto r_l2 = decode_kernel<false, 2, false, 1, 8, 1, 0, 0, true>;
    auto r_l2d = decode_kernel<false, 2, true, 1, 8, 1, 0, 0, true>;
    hipError_t e = once_per_device(&DeviceState::redo_ready, [&] {
        const void *all[] = {(const void *)r_lds, (const void *)r_l2, (const void *)r_l2d};
        for (const void *f : all) {
            hipError_t r = allow_lds(f, DEC_LDS_MAX);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    });
    if (e != hipSuccess) return e;
    if (!p.sec_lds && p.P != 8) return hipErrorInvalidValue;
    const size_t lds = 1024 + (size_t(256) << p.P) * 2 + (p.sec_lds ? ((size_t(p.nsec) * 2 + 15) & ~size_t(15)) : 0);
    if (lds > size_t(DEC_LDS_MAX)) return hipErrorInvalidValue;
    const uint64_t rwant = (p.nchunks + DEC_THREADS - 1) / DEC_THREADS;
    const int rgrid = int(rwant > uint64_t(cu_count()) ? uint64_t(cu_count()) : rwant);
    hipLaunchKernelGGL(p.sec_lds ? r_lds : p.direct ? r_l2d : r_l2, dim3(rgrid < 1 ? 1 : rgrid), dim3(DEC_THREADS), lds, st, p);
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
    // instantiations: <SEC_LDS, SPR, DIRECT, K, GW, OUTB, PC, HC[, REDO, NT, DEPTH]>
    // tables in LDS -> wide (2 streams, 64-byte granules and store bursts) or light (4 streams, 32-byte
    // granules and store bursts: 16-byte stores reach HBM as 32-byte writes, measured -14 %; 64-byte bursts
    // measured the same as 32-byte ones here: 4 GiB text 4.24 vs 4.27 ms); L2 gathers -> 4 streams, 32-byte
    // granules, 64-byte bursts.  The MH_LIGHT_* / MH_L2D_* macros exist for A/B builds (csrc/Makefile `exp`).
#ifndef MH_LIGHT_OUTB
#define MH_LIGHT_OUTB 2
#endif
#ifndef MH_LIGHT_DEPTH
#define MH_LIGHT_DEPTH 2
#endif
    void (*k_lds2_light[2])(DecParams) = {decode_kernel<true, 2, false, 4, 8, MH_LIGHT_OUTB, 0, 0, false, 512, MH_LIGHT_DEPTH>,
                                          decode_kernel<true, 2, false, 4, 8, MH_LIGHT_OUTB, 8, 0, false, 512, MH_LIGHT_DEPTH>};
    auto k_lds4 = decode_kernel<true, 4, false, 2, 16, 4, 8, 0>;
    auto k_lds4_light = decode_kernel<true, 4, false, 4, 8, MH_LIGHT_OUTB, 8, 0, false, 512, MH_LIGHT_DEPTH>;
    auto k_l2 = decode_kernel<false, 2, false, 4, 8, MH_LIGHT_OUTB, 8, 0, false, 512, MH_LIGHT_DEPTH>;
    // second-level height H as a template constant where it is common (max code length 10..12 and >= 16): the
    // table index is then two instructions with immediate operands
#ifndef MH_L2D_OUTB
#define MH_L2D_OUTB 4
#endif
#ifndef MH_L2D_DEPTH
#define MH_L2D_DEPTH 1
#endif
#ifndef MH_L2D_K
#define MH_L2D_K 4
#endif
#ifndef MH_L2D_GW
#define MH_L2D_GW 8
#endif
#ifndef MH_L2D_NT
#define MH_L2D_NT 512
#endif

#define L2D(SPRV, HCV) decode_kernel<false, SPRV, true, MH_L2D_K, MH_L2D_GW, MH_L2D_OUTB, 8, HCV, false, MH_L2D_NT, MH_L2D_DEPTH>
    void (*k_l2d[9])(DecParams) = {L2D(2, 0), L2D(2, 0), L2D(2, 2), L2D(2, 3), L2D(2, 4), L2D(2, 0), L2D(2, 0), L2D(2, 0), L2D(2, 8)};
#undef L2D
    e = once_per_device(&DeviceState::decode_ready, [&] {
        const void *all[] = {(const void *)k_lds2_light[0], (const void *)k_lds2_light[1],
                             (const void *)k_lds4, (const void *)k_lds4_light, (const void *)k_l2, (const void *)k_l2d[0], (const void *)k_l2d[2],
                             (const void *)k_l2d[3], (const void *)k_l2d[4], (const void *)k_l2d[8]};
        for (const void *f : all) {
            hipError_t r = allow_lds(f, DEC_LDS_MAX);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    });
    if (e != hipSuccess) return e;
    if (!p.sec_lds && p.P != 8) return hipErrorInvalidValue;       // the L2 layouts are built with P = 8
    size_t lds = 1024 + (size_t(256) << p.P) * 2 + (p.sec_lds ? ((size_t(p.nsec) * 2 + 15) & ~size_t(15)) : 0);
    if (lds > size_t(DEC_LDS_MAX)) return hipErrorInvalidValue;
    // With the tables in LDS and NO second level (every code <= 8 bits) the kernel is bound by how the
    // streams touch HBM once the payload is a large part of the traffic: uniform data runs 1.6x faster
    // with the wide streams.  With a second level the dependent lookups dominate and four streams per
    // lane win (measured with Zipf code lengths and all tables in LDS: 5.5 ms light, 7.7 ms wide per
    // 4 GiB), as they do for low-ratio data (41 %-ratio text 8 % slower with the wide streams).
    const bool short_codes = p.nsec == 0 && p.P == 8;
    const bool wide = p.sec_lds && short_codes && p.n > 0 && p.nbits * 10 > p.n * 8 * 6;      // ratio > 0.6
    const bool l2d = !p.sec_lds && p.direct;
    const uint64_t per_block = l2d ? uint64_t(MH_L2D_NT) * MH_L2D_K : uint64_t(DEC_THREADS) * (wide ? 2 : 4);
    uint64_t want = (p.nchunks + per_block - 1) / per_block;
    int grid = int(want > uint64_t(cu_count()) ? uint64_t(cu_count()) : want);
    const int p8 = p.P == 8;
    if (wide) hipLaunchKernelGGL(k_lds4, dim3(grid), dim3(DEC_THREADS), lds, st, p);
    else if (p.sec_lds) hipLaunchKernelGGL(short_codes ? k_lds4_light : k_lds2_light[p8], dim3(grid), dim3(DEC_THREADS), lds, st, p);
    else if (p.direct) hipLaunchKernelGGL(k_l2d[p.H <= 8 ? p.H : 0], dim3(grid), dim3(MH_L2D_NT), lds, st, p);
    else hipLaunchKernelGGL(k_l2, dim3(grid), dim3(DEC_THREADS), lds, st, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // chunks with a code longer than both table levels: normally none, and the pass returns at once
    return launch_decode_redo(p, st);
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
constexpr uint32_t IDX_MAX_PASSES = 96;
struct IdxWs { size_t off_changed, off_end, off_used, off_count, off_start, off_blk, total; uint64_t nseg, nblk;
               size_t off_e16, off_s16, off_c16, off_dirty, off_tcnt, off_tbase, off_tblk; uint64_t nseg5, ntile5, ntblk, dirty_cap; };
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
    // the fast path (index_tile_kernel): 6 bytes per 256-bit segment, a list of the segments to repair (a quarter of them at
    // most: beyond that the stream does not synchronise this way) and 12 bytes per tile, in the same space (one path runs at a time)
    w.nseg5 = (nbits + IX_SEG_BITS - 1) / IX_SEG_BITS;
    w.ntile5 = (w.nseg5 + IX_TILE_SEGS - 1) / IX_TILE_SEGS;
    w.ntblk = (w.ntile5 + SCAN_BLOCK - 1) / SCAN_BLOCK;
    w.dirty_cap = w.nseg5 / 4 + 64;
    w.off_e16 = w.off_end;
    w.off_s16 = up(w.off_e16 + size_t(w.nseg5) * 2);
    w.off_c16 = up(w.off_s16 + size_t(w.nseg5) * 2);
    w.off_dirty = up(w.off_c16 + size_t(w.nseg5) * 2);
    w.off_tcnt = up(w.off_dirty + size_t(w.dirty_cap) * 4);
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
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<uint32_t *>(ws + 8), path);
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
    p.seg_bits = IDX_SEG_BITS - IDX_SEG_BITS % g;
    p.nseg = (p.nbits + p.seg_bits - 1) / p.seg_bits;          // <= L.nseg
    hipError_t e = hipMemsetAsync(ws, 0, L.off_end, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(p.n_symbols, 0, 8, st);
    if (e != hipSuccess || p.nbits == 0) return e;
    e = once_per_device(&DeviceState::index_ready, [] { return allow_lds(reinterpret_cast<const void *>(build_index_kernel<1>), 131072); });
    if (e != hipSuccess) return e;
    if (p.P > 8) return hipErrorInvalidValue;
    // ---- the fast path: order 1, every code within the tile decoder's two table levels, no code-length lattice (g == 1),
    // a stream worth a launch of 256 workgroups.  Given up (and the segment iteration below started from scratch) when more
    // than an eighth of the segments did not synchronise within their warm-up, or the repairs do not die out.
    if (p.order != 2 && p.tprim && p.tP == 7 && p.max_len <= p.tP + p.tH && g == 1 && p.nbits >= (1ull << 20) && L.nseg5 < 0xFFFFFFFFull && !getenv("MH_INDEX_NO_TILES")) {
        IdxParams q = p;
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
            // per wave of the card at least) with 128 bits tells — where that leaves under 2 % of the segments to repair the
            // short warm-up serves the whole stream (the pass decodes warm-up + segment: 1.44 instead of 1.89 segment lengths)
            IdxParams sq = q;
            sq.ntile5 = q.ntile5 / 256 > 4096 ? q.ntile5 / 256 : 4096;
            sq.nseg5 = sq.ntile5 * IX_TILE_SEGS;
            sq.warm_bits = 128;
            sq.iter = it;
            e = launch_index_tile(sq, 0, st);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(index_tile_dirty_kernel, dim3(256), dim3(256), 0, st, sq);
            unsigned int dirty = ~0u;
            e = hipMemcpyAsync(&dirty, q.changed + it, 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
            ++it;
            if (uint64_t(dirty) * 50u < sq.nseg5) q.warm_bits = 128;
        }
        for (int attempt = 0; attempt < 2 && !ok; ++attempt) {
            e = launch_index_tile(q, 0, st);
            if (e != hipSuccess) return e;
            unsigned int prev_dirty = ~0u;
            bool hopeless = false;
            for (const uint32_t it_end = it + 24u; it < it_end && !ok && !hopeless; ++it) {
                q.iter = it;
                hipLaunchKernelGGL(index_tile_dirty_kernel, dim3(dgrid), dim3(256), 0, st, q);
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
        if (ok) {
            note_index_path(ws, IDX_PATH_TILES, st);
            unsigned long long *tblk = reinterpret_cast<unsigned long long *>(ws + L.off_tblk);
            hipLaunchKernelGGL(index_tile_count_kernel, dim3(unsigned((q.ntile5 + 3) / 4)), dim3(256), 0, st, q);
            hipLaunchKernelGGL(scan_local_kernel, dim3(unsigned(L.ntblk)), dim3(SCAN_THREADS), 0, st, q.tile_cnt, q.ntile5, q.tile_base, tblk);
            hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, tblk, L.ntblk, static_cast<const unsigned long long *>(nullptr));
            hipLaunchKernelGGL(index_scan_add_kernel, dim3(unsigned(L.ntblk)), dim3(SCAN_THREADS), 0, st, q.tile_base, tblk, q.ntile5, L.ntblk, q.n_symbols);
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
    hipLaunchKernelGGL(scan_local_kernel, dim3(unsigned(nblk)), dim3(SCAN_THREADS), 0, st, p.seg_count, p.nseg, p.seg_sym_start, blk_sum);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, blk_sum, nblk, static_cast<const unsigned long long *>(nullptr));
    hipLaunchKernelGGL(index_scan_add_kernel, dim3(unsigned(nblk)), dim3(SCAN_THREADS), 0, st, p.seg_sym_start, blk_sum, p.nseg, nblk, p.n_symbols);
    hipLaunchKernelGGL(index_fill_kernel, dim3(grid), dim3(256), 0, st, p);
    return hipGetLastError();
}

}  // namespace mhk
