// mh_hist.hip — histograms of the Markov-Huffman hot path for gfx950 (order 2: N4).
//   hist_o1_kernel     256x256 conditional histogram, LDS-resident packed counters (+ slab reduce)   (a1)
//   hist_o0_kernel     256-bin histogram                                                            (a2)
//   (order 2: mh_hist2.hip)
// All integer work: no MFMA.  Waves are 64 wide; workgroups never wait for each other.
#include "mh_dev.hpp"

namespace mhk {

__global__ void set_word_kernel(uint32_t *p, uint32_t v) { *p = v; }
__global__ void hist_header_kernel(HistHeader *hdr, HistHeader v) { *hdr = v; }
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


// (hist_fixup / hist_add: mh_dev.hpp — the order-2 bucket kernel counts with the same fields)

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
// [r4] A block is 64 words x 4 slices of the slabs (a thread's loop over all 256 slabs was a chain of 256 loads on 2 waves per
// CU: 72 us whatever the input size, 1.5 % of a 2 GiB shard's step); the four partial sums meet in LDS.
constexpr uint32_t HIST_REDUCE_WORDS = 64, HIST_REDUCE_SLICES = 4;
__global__ __launch_bounds__(256) void hist_reduce_kernel(const uint32_t *__restrict__ slab, uint32_t nslab, unsigned long long *counts,
                                                          unsigned int *check, unsigned long long n) {
    __shared__ unsigned long long plo[HIST_REDUCE_SLICES][HIST_REDUCE_WORDS], phi[HIST_REDUCE_SLICES][HIST_REDUCE_WORDS];
    const uint32_t col = threadIdx.x & (HIST_REDUCE_WORDS - 1u), slice = threadIdx.x / HIST_REDUCE_WORDS;
    const uint32_t w = blockIdx.x * HIST_REDUCE_WORDS + col;     // < 32768
    const uint32_t per = (nslab + HIST_REDUCE_SLICES - 1u) / HIST_REDUCE_SLICES;
    const uint32_t s_begin = slice * per, s_end = s_begin + per < nslab ? s_begin + per : nslab;
    unsigned long long lo = 0, hi = 0;
    for (uint32_t s = s_begin; s < s_end; ++s) {
        const uint32_t v = slab[size_t(s) * 32768u + w];
        lo += v & 0xFFFFu;
        hi += v >> 16;
    }
    plo[slice][col] = lo; phi[slice][col] = hi;
    __syncthreads();
    if (slice != 0) return;
#pragma unroll
    for (uint32_t k = 1; k < HIST_REDUCE_SLICES; ++k) { lo += plo[k][col]; hi += phi[k][col]; }
    const uint32_t s1 = w | 0x8000u;
    const unsigned long long c0 = counts[hist_slot_prev(w) * 256u + (w >> 8)] + lo;
    const unsigned long long c1 = counts[hist_slot_prev(s1) * 256u + (s1 >> 8)] + hi;
    counts[hist_slot_prev(w) * 256u + (w >> 8)] = c0;
    counts[hist_slot_prev(s1) * 256u + (s1 >> 8)] = c1;
    if (!check) return;
    unsigned long long t = c0 + c1;
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);     // (slice 0 is one wave)
    if (threadIdx.x == 0) {
        unsigned long long *total = reinterpret_cast<unsigned long long *>(check + 2);
        atomicAdd(total, t);
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
        hipLaunchKernelGGL(hist_reduce_kernel, dim3(32768 / HIST_REDUCE_WORDS), dim3(256), 0, st, slab, uint32_t(g.grid), d_counts, check,
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
    if (slab) hipLaunchKernelGGL(hist_reduce_kernel, dim3(32768 / HIST_REDUCE_WORDS), dim3(256), 0, st, slab, uint32_t(grid), d_counts, check,
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

// workspace: [0,64) status | wt_bits u32[nwt] | wt_start u64[nwt] | blk_sum u64[nblk + 1]

}  // namespace mhk
