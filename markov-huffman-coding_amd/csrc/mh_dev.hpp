// mh_dev.hpp — what the kernel files of the hot path share (round 4: mh_kernels.hip, 3 700 lines, became mh_hist.hip,
// mh_encode.hip, mh_decode.hip and mh_index.hip beside mh_tile.hip and mh_tree.hip):
//   * wave helpers (64 lanes: DPP / __shfl scans), the lane's 16-byte input vector and the byte(s) in front of it,
//   * the block scan the prefix kernels are built from,
//   * the per-device launch state (CU count, dynamic-LDS attributes made once per device) and the region geometry that the
//     histogram and the region encoder must agree on.
// Device functions are inline in every translation unit (no relocatable device code); kernels live in exactly one file
// and are reached from the others through host launchers declared in mh_kernels.h or at the end of this header.
#pragma once
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


// ---- what the order-1 histogram and the region encoder that reads its workspace share -----------------------
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

constexpr uint64_t HIST_WS_MAGIC = 0x4D48525247303031ull;                 // "MHRRG001"

struct HistHeader { unsigned long long magic, n, data, region_vecs; uint32_t grid, prev0, cross_cap, pad; };

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
inline DeviceState g_dev[MAX_DEVICES];
inline std::mutex g_dev_mu;

inline DeviceState &device_state() {          // caller holds g_dev_mu
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    DeviceState &d = g_dev[dev];
    if (d.cu_count == 0) {
        hipDeviceProp_t prop;
        d.cu_count = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return d;
}

inline int cu_count() {
    std::lock_guard<std::mutex> lock(g_dev_mu);
    return device_state().cu_count;
}

inline hipError_t allow_lds(const void *fn, int bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// Runs `setup` once per device (under the lock); `flag` selects the DeviceState member.
template <typename F>
inline hipError_t once_per_device(bool DeviceState::*flag, F setup) {
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
inline RegionGeom region_geom(uint64_t n) {
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

// kernels of mh_encode.hip that the index builder (mh_index.hip) uses too
hipError_t launch_scan_local(const uint32_t *d_vals, uint64_t n, unsigned long long *d_start, unsigned long long *d_blk_sum, hipStream_t st);
hipError_t launch_scan_top(unsigned long long *d_blk_sum, uint64_t nblk, const unsigned long long *d_carry0, hipStream_t st);

}  // namespace mhk
