// mh_hist2.hip — the order-2 histogram (an extension the reference only speculates about,
// README.md:158-166 — PARITY UNPINNED, the spec is the generalised oracle).  counts[ctx * 256 + sym], ctx = (byte before
// previous) << 8 | previous byte: 16.7 M 64-bit counters, 128 MiB in HBM.  Two ways to fill them:
//   hist_o2_kernel      an LDS tag cache of 16 384 (key, count) slots per workgroup in front of 64-bit global atomics:
//                       sources with a few thousand live keys (text) never leave LDS
//   hist2_* (r4)        sources with millions of live keys (Zipf or uniform bytes) miss the cache and would run at the rate
//                       of global atomics over 128 MiB (36 ms per GiB).  Instead the positions are PARTITIONED by the
//                       context's high byte — 256 buckets of (previous byte, symbol) pairs, two bytes per position, staged
//                       through LDS so that every bucket is written in runs — and each bucket is then an order-1 problem:
//                       65 536 packed counters in LDS (the order-1 kernel's two-guard-bit fields), flushed once per chunk.
// Which one runs is decided on the device, per slab of the input, from the cache misses of the slab's first 4 MiB
// (which are counted either way): no host round trip.  All integer work, 64-wide waves, no workgroup waits for another.
#include "mh_dev.hpp"

namespace mhk {

// ------------------------------------------------------------------------------------------------
// the tag cache
// ------------------------------------------------------------------------------------------------
// A workgroup keeps 16384 (key, count) slots in LDS,
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
constexpr int H2_LDS_BYTES = int(H2_SLOTS) * 8 + 64 * 4 + 16;     // tags, counters + one dummy word per lane, the claim counter, the miss counter
constexpr uint32_t H2_PROBES = 8, H2_PROBES_FULL = 2, H2_FULL = H2_SLOTS * 3 / 4;

// control block = the first 64 bytes of the workspace: [0] status, [1] the path chosen for the current slab (1 tag cache,
// 2 partition), [2] the choices so far (bit 0: a slab stayed in the tag cache, bit 1: a slab was partitioned — the word
// mh_dev_index_path reads), [4..5] cache
// misses of the sample (64-bit)
enum : int { H2_MODE_PLAIN = 0, H2_MODE_SAMPLE = 1, H2_MODE_IF_CACHE = 2, H2_MODE_IF_PARTITION = 3 };
constexpr uint32_t H2_SEL_CACHE = 1, H2_SEL_PARTITION = 2;

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
    atomicAdd(used + 1, 1u);                                     // the miss counter (what the choice of path is made from)
}

// Counts the bytes [lo, hi) of base[0, n_total) (lo a multiple of 16); ctx0 = the context in front of base[0].
__global__ __launch_bounds__(H2_THREADS) void hist_o2_kernel(const uint8_t *__restrict__ base, uint64_t n_total, uint64_t lo, uint64_t hi,
                                                             uint32_t ctx0, unsigned long long *counts, uint32_t *ctl, int mode) {
    if (mode == H2_MODE_IF_CACHE && ctl[1] != H2_SEL_CACHE) return;
    if (mode == H2_MODE_IF_PARTITION && ctl[1] != H2_SEL_PARTITION) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tag = reinterpret_cast<uint32_t *>(smem);
    uint32_t *cnt = tag + H2_SLOTS;                              // (+ 64 dummy words behind the slots)
    uint32_t *used = cnt + H2_SLOTS + 64;                        // [0] slots claimed so far, [1] keys sent to memory
    for (uint32_t i = threadIdx.x; i < H2_SLOTS; i += H2_THREADS) { tag[i] = H2_EMPTY; cnt[i] = 0; }
    if (threadIdx.x == 0) { used[0] = 0; used[1] = 0; }
    __syncthreads();
    const uint8_t *data = base + lo;
    const uint64_t n = hi - lo;
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
        if ((threadIdx.x & 63u) == 0) ctx = live ? ctx_before(base, n_total, lo + (v << 4), ctx0) : ctx0;
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
        uint32_t ctx = ctx_before(base, n_total, lo + i, ctx0);
        for (; i < n; ++i) {
            const uint32_t key = (ctx << 8) | data[i];
            atomicAdd(&counts[key], 1ull);
            ctx = key & 0xFFFFu;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < H2_SLOTS; i += H2_THREADS)
        if (tag[i] != H2_EMPTY && cnt[i]) atomicAdd(&counts[tag[i]], (unsigned long long)cnt[i]);
    if (mode == H2_MODE_SAMPLE && threadIdx.x == 0 && used[1])
        atomicAdd(reinterpret_cast<unsigned long long *>(ctl + 4), (unsigned long long)used[1]);
}

// after the sample: more than 1 key in 32 went to memory -> partition the rest of the slab
__global__ void hist2_select_kernel(uint32_t *ctl, unsigned long long sample_n, uint32_t force) {
    unsigned long long *miss = reinterpret_cast<unsigned long long *>(ctl + 4);
    uint32_t sel = *miss * 32ull > sample_n ? H2_SEL_PARTITION : H2_SEL_CACHE;
    if (force == H2_SEL_CACHE || force == H2_SEL_PARTITION) sel = force;
    ctl[1] = sel;
    ctl[2] |= sel;
    *miss = 0;
}

// ------------------------------------------------------------------------------------------------
// the partition path.  One slab = the bytes [lo, hi) (whole 64 KiB tiles), G = 256 workgroups, each with a contiguous
// range of tiles.  Bucket of position i = byte i - 2; its pair = byte i - 1 << 8 | byte i.
//   hist2_count_kernel    cnt[bucket * G + g] = positions of workgroup g's range that fall into the bucket
//   hist2_offsets_kernel  exclusive scan of cnt rounded up to units (bucket-major): where each workgroup's share of each
//                         bucket starts; the buckets' bounds; the work items (bucket, share of <= 4 Mi pairs) of the bucket kernel
//   hist2_scatter_kernel  per tile: ranks within the tile by LDS atomics, pairs sorted by bucket in a 135 KiB LDS stage,
//                         copied out in whole 16-byte units (what is left of a bucket waits in LDS for the next tile)
//   hist2_bucket_kernel   per item: 65 536 packed counters in LDS, stored as the item's image
//   hist2_reduce_kernel   the images of a bucket's items summed into its 64-bit counters
// ------------------------------------------------------------------------------------------------
constexpr int H2P_G = 256;
constexpr uint32_t H2P_TILE = 65536;                             // bytes = positions per tile: 64 per thread
constexpr uint32_t H2P_CHUNK = 1u << 22;                         // pairs per work item of the bucket kernel
constexpr uint64_t H2P_PAD_PAIRS = 256ull * H2P_G * 8;                // room for the padding of every (workgroup, bucket) share
constexpr uint64_t H2P_SAMPLE = 4ull << 20;                      // bytes of a slab that go through the tag cache to choose the path
constexpr uint64_t H2P_MIN = 32ull << 20;                        // slabs shorter than this are not worth four more launches
constexpr uint64_t H2P_SLAB = 2ull << 30;                        // bytes per slab: a 4 GiB pair buffer at most

// workspace behind the control block
struct H2Geom { size_t off_cnt, off_off, off_bstart, off_items, off_holes, off_image, off_pairs, total; uint32_t max_items; };
static inline H2Geom hist2_geom(uint64_t n, uint64_t slab) {
    H2Geom g;
    g.off_cnt = 64;
    g.off_off = g.off_cnt + size_t(256) * H2P_G * 4;
    g.off_bstart = g.off_off + (size_t(256) * H2P_G + 1) * 8;
    g.off_items = g.off_bstart + 257 * 8;
    g.off_holes = g.off_items + 260 * 4;
    const uint64_t most = (n < slab ? n : slab) + H2P_PAD_PAIRS;         // pairs: one per position + up to seven per (workgroup, bucket)
    g.max_items = uint32_t((most + H2P_CHUNK - 1) / H2P_CHUNK) + 256u;   // sum over the buckets of ceil(length / chunk)
    g.off_image = (g.off_holes + 256 * 4 + 255) & ~size_t(255);
    g.off_pairs = g.off_image + size_t(g.max_items) * 32768u * 4u;
    g.total = g.off_pairs + ((size_t(most) * 2 + 255) & ~size_t(255));
    return g;
}

// 64 private copies of the 256 counters per workgroup — four per wave, chosen by the lane, 257 words apart so that the four
// land in different banks: a frequent byte value (Zipf: one in six) would otherwise make ten lanes of every add queue up on
// one LDS word (1.5 ms per 2 GiB of Zipf bytes against 0.63 of uniform ones with one copy per wave).
constexpr int H2P_COUNT_STRIDE = 257, H2P_COUNT_LDS = 64 * H2P_COUNT_STRIDE * 4;
__global__ __launch_bounds__(1024) void hist2_count_kernel(const uint8_t *__restrict__ base, uint64_t n_total, uint64_t lo, uint32_t ntiles,
                                                           uint32_t ctx0, uint32_t *cnt, const uint32_t *ctl) {
    if (ctl[1] != H2_SEL_PARTITION) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    for (int i = threadIdx.x; i < 64 * H2P_COUNT_STRIDE; i += 1024) h[i] = 0;
    __syncthreads();
    const uint32_t per = (ntiles + H2P_G - 1) / H2P_G;
    const uint32_t t0 = blockIdx.x * per < ntiles ? blockIdx.x * per : ntiles;
    const uint32_t t1 = t0 + per < ntiles ? t0 + per : ntiles;
    uint32_t *mine = h + ((threadIdx.x >> 6) * 4u + (threadIdx.x & 3u)) * H2P_COUNT_STRIDE;
    const uint64_t b0 = lo + uint64_t(t0) * H2P_TILE, b1 = lo + uint64_t(t1) * H2P_TILE;
    const uint4 *vdata = reinterpret_cast<const uint4 *>(base);
    for (uint64_t v = (b0 >> 4) + threadIdx.x; v < (b1 >> 4); v += 1024) {
        const uint4 x = vdata[v];
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(&mine[w[k] & 255u], 1u);
            atomicAdd(&mine[(w[k] >> 8) & 255u], 1u);
            atomicAdd(&mine[(w[k] >> 16) & 255u], 1u);
            atomicAdd(&mine[w[k] >> 24], 1u);
        }
    }
    __syncthreads();
    // the buckets are the bytes two places EARLIER: the two bytes in front of the range come in, its last two go out
    if (threadIdx.x == 0 && t1 > t0) {
        const uint32_t c = ctx_before(base, n_total, b0, ctx0);
        atomicAdd(&h[c >> 8], 1u);
        atomicAdd(&h[c & 255u], 1u);
        atomicSub(&h[base[b1 - 2]], 1u);
        atomicSub(&h[base[b1 - 1]], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t s = 0;
        for (int w = 0; w < 64; ++w) s += h[w * H2P_COUNT_STRIDE + threadIdx.x];
        cnt[threadIdx.x * H2P_G + blockIdx.x] = s;
    }
}

// Every workgroup's share of every bucket is rounded up to whole 16-byte units of eight pairs (the scatter kernel pads the
// last unit with zero pairs and says how many in holes[bucket]: they are counted as the pair (0, 0) and taken off again
// by the reduce kernel), so every copy of the scatter kernel and every load of the bucket kernel is an aligned 16 bytes.
__global__ __launch_bounds__(1024) void hist2_offsets_kernel(const uint32_t *__restrict__ cnt, unsigned long long *off, unsigned long long *bstart,
                                                             uint32_t *item_start, uint32_t *holes, const uint32_t *ctl) {
    if (ctl[1] != H2_SEL_PARTITION) return;
    __shared__ uint64_t lds[SCAN_THREADS / 64];
    __shared__ unsigned long long sb[257];
    constexpr int PER = 256 * H2P_G / 1024;                      // 64 consecutive entries per thread: a quarter of a bucket
    const uint4 *src = reinterpret_cast<const uint4 *>(cnt + size_t(threadIdx.x) * PER);
    auto up8 = [](uint32_t c) { return uint64_t((c + 7u) & ~7u); };
    uint64_t sum = 0;
    for (int i = 0; i < PER / 4; ++i) { const uint4 x = src[i]; sum += up8(x.x) + up8(x.y) + up8(x.z) + up8(x.w); }
    uint64_t total;
    uint64_t run = block_excl_scan(sum, lds, total);
    if ((threadIdx.x & 3u) == 0) sb[threadIdx.x >> 2] = run;
    if (threadIdx.x == 0) sb[256] = total;
    unsigned long long *dst = off + size_t(threadIdx.x) * PER;
    for (int i = 0; i < PER / 4; ++i) {                          // (read again: sixty-four values per thread do not fit the registers of a 1024-thread block)
        const uint4 x = src[i];
        dst[4 * i] = run; run += up8(x.x);
        dst[4 * i + 1] = run; run += up8(x.y);
        dst[4 * i + 2] = run; run += up8(x.z);
        dst[4 * i + 3] = run; run += up8(x.w);
    }
    if (threadIdx.x == 0) off[size_t(256) * H2P_G] = total;
    __syncthreads();
    uint64_t items = 0;
    if (threadIdx.x < 256) {
        const unsigned long long len = sb[threadIdx.x + 1] - sb[threadIdx.x];
        items = (len + H2P_CHUNK - 1) / H2P_CHUNK;
        bstart[threadIdx.x] = sb[threadIdx.x];
        holes[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) bstart[256] = total;
    uint64_t nitems;
    const uint64_t first = block_excl_scan(items, lds, nitems);
    if (threadIdx.x < 256) item_start[threadIdx.x] = uint32_t(first);
    if (threadIdx.x == 0) item_start[256] = uint32_t(nitems);
}

// bytes S[0..17] of a lane's stream (the two bytes in front of its vector, then the vector) as five dwords y[0..4];
// S[j] | S[j+1] << 8 | S[j+2] << 16 (| S[j+3] << 24) is one byte-funnel shift
#define H2P_TRIPLE(y, k, j) __builtin_amdgcn_alignbyte((y)[5 * (k) + ((j) >> 2) + 1], (y)[5 * (k) + ((j) >> 2)], uint32_t((j) & 3))

// LDS of the scatter kernel: the stage in units of eight pairs (a bucket's slot starts on a unit: the pairs carried over from
// the tile before, then this tile's), and per bucket: the carried unit, stage-to-memory unit offset, next free unit in memory,
// this tile's count, slot start, first free pair, end of the slot's whole units, pairs carried; and the bucket of every unit
constexpr uint32_t H2P_STAGE_UNITS = (H2P_TILE + 256u * 14u) / 8u;
constexpr int H2P_L_CARRY = int(H2P_STAGE_UNITS) * 16, H2P_L_GUN = H2P_L_CARRY + 256 * 16, H2P_L_CURSOR = H2P_L_GUN + 256 * 8,
              H2P_L_COUNT = H2P_L_CURSOR + 256 * 8, H2P_L_SU = H2P_L_COUNT + 256 * 4, H2P_L_LBASE = H2P_L_SU + 260 * 4,
              H2P_L_UEND = H2P_L_LBASE + 256 * 4, H2P_L_CARRYN = H2P_L_UEND + 256 * 4, H2P_L_UOWN = H2P_L_CARRYN + 256 * 4,
              H2P_SCATTER_LDS = H2P_L_UOWN + int(H2P_STAGE_UNITS);

__global__ __launch_bounds__(1024) void hist2_scatter_kernel(const uint8_t *__restrict__ base, uint64_t n_total, uint64_t lo, uint32_t ntiles,
                                                             uint32_t ctx0, const unsigned long long *__restrict__ off, uint4 *pairs,
                                                             uint32_t *holes, const uint32_t *ctl) {
    if (ctl[1] != H2_SEL_PARTITION) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *stage = reinterpret_cast<uint4 *>(smem);
    uint16_t *stage16 = reinterpret_cast<uint16_t *>(smem);
    uint4 *carry = reinterpret_cast<uint4 *>(smem + H2P_L_CARRY);
    unsigned long long *gun = reinterpret_cast<unsigned long long *>(smem + H2P_L_GUN);
    unsigned long long *cursor = reinterpret_cast<unsigned long long *>(smem + H2P_L_CURSOR);
    uint32_t *lcount = reinterpret_cast<uint32_t *>(smem + H2P_L_COUNT);
    uint32_t *su = reinterpret_cast<uint32_t *>(smem + H2P_L_SU);
    uint32_t *lbase = reinterpret_cast<uint32_t *>(smem + H2P_L_LBASE);
    uint32_t *uend = reinterpret_cast<uint32_t *>(smem + H2P_L_UEND);
    uint32_t *carryn = reinterpret_cast<uint32_t *>(smem + H2P_L_CARRYN);
    uint8_t *uown = smem + H2P_L_UOWN;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t per = (ntiles + H2P_G - 1) / H2P_G;
    const uint32_t t0 = blockIdx.x * per < ntiles ? blockIdx.x * per : ntiles;
    const uint32_t t1 = t0 + per < ntiles ? t0 + per : ntiles;
    if (threadIdx.x < 256) {
        lcount[threadIdx.x] = 0; carryn[threadIdx.x] = 0;
        cursor[threadIdx.x] = off[size_t(threadIdx.x) * H2P_G + blockIdx.x] >> 3;
        carry[threadIdx.x] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    uint4 nx[4];
    if (t0 < t1) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            nx[k] = *reinterpret_cast<const uint4 *>(base + lo + uint64_t(t0) * H2P_TILE + (uint64_t(k) * 1024 + threadIdx.x) * 16);
    }
    for (uint32_t tile = t0; tile < t1; ++tile) {
        const uint64_t tb = lo + uint64_t(tile) * H2P_TILE;
        uint32_t y[20];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 x = nx[k];
            uint32_t hd = __shfl_up(x.w >> 16, 1);                                   // bytes 14 | 15 << 8 of the previous lane's vector
            if (lane == 0) {
                const uint32_t c = ctx_before(base, n_total, tb + (uint64_t(k) * 1024 + threadIdx.x) * 16, ctx0);
                hd = (c >> 8) | ((c & 255u) << 8);
            }
            y[5 * k] = hd | (x.x << 16); y[5 * k + 1] = (x.x >> 16) | (x.y << 16); y[5 * k + 2] = (x.y >> 16) | (x.z << 16);
            y[5 * k + 3] = (x.z >> 16) | (x.w << 16); y[5 * k + 4] = x.w >> 16;
        }
        // rank of every position within its bucket (the order inside a bucket does not matter to a histogram)
        uint32_t rk[32];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t r = atomicAdd(&lcount[H2P_TRIPLE(y, k, j) & 255u], 1u);
                const int i = 16 * k + j;
                rk[i >> 1] = (i & 1) ? (rk[i >> 1] | (r << 16)) : r;
            }
            __builtin_amdgcn_sched_barrier(0);                   // (sixteen atomics in flight at a time: all sixty-four spill registers)
        }
        __syncthreads();
        if (tile + 1 < t1) {                                     // the next tile's vectors: on their way during everything below
#pragma unroll
            for (int k = 0; k < 4; ++k)
                nx[k] = *reinterpret_cast<const uint4 *>(base + tb + H2P_TILE + (uint64_t(k) * 1024 + threadIdx.x) * 16);
        }
        if (wave == 0) {                                         // the buckets' slots: a scan of their sizes in units, four buckets per lane
            const uint4 c4 = reinterpret_cast<const uint4 *>(lcount)[lane], k4 = reinterpret_cast<const uint4 *>(carryn)[lane];
            const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w}, k[4] = {k4.x, k4.y, k4.z, k4.w};
            uint32_t s = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) s += (k[i] + c[i] + 7u) >> 3;
            uint32_t u = wave_inclusive_sum(s) - s;
            uint32_t lb[4], ue[4], kn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t a = 4u * lane + uint32_t(i), tot = k[i] + c[i];
                su[a] = u;
                lb[i] = u * 8u + k[i]; ue[i] = u + (tot >> 3); kn[i] = tot & 7u;
                const unsigned long long cur = cursor[a];
                gun[a] = cur - u;
                cursor[a] = cur + (tot >> 3);
                if (k[i]) stage[u] = carry[a];                   // what the tile before left over leads the slot
                u += (tot + 7u) >> 3;
            }
            if (lane == 63) su[256] = u;
            reinterpret_cast<uint4 *>(lbase)[lane] = make_uint4(lb[0], lb[1], lb[2], lb[3]);
            reinterpret_cast<uint4 *>(uend)[lane] = make_uint4(ue[0], ue[1], ue[2], ue[3]);
            reinterpret_cast<uint4 *>(carryn)[lane] = make_uint4(kn[0], kn[1], kn[2], kn[3]);
            reinterpret_cast<uint4 *>(lcount)[lane] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        // (the byte triples are formed again rather than kept: sixty-four more live registers would spill)
#pragma unroll
        for (int i = 0; i < 20; ++i) asm volatile("" : "+v"(y[i]));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t t = H2P_TRIPLE(y, k, j);
                const int i = 16 * k + j;
                const uint32_t r = (i & 1) ? (rk[i >> 1] >> 16) : (rk[i >> 1] & 0xFFFFu);
                stage16[lbase[t & 255u] + r] = uint16_t(t >> 8);                     // the pair: previous byte | symbol << 8
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // whose slot every unit of the stage is (a wave marks sixteen buckets, 64 units per trip)
#pragma unroll 1
        for (uint32_t a = wave * 16u; a < wave * 16u + 16u; ++a) {
            const uint32_t b1 = su[a + 1];
            for (uint32_t u = su[a] + lane; u < b1; u += 64u) uown[u] = uint8_t(a);
        }
        __syncthreads();
        // copy-out: a thread per unit; a slot's WHOLE units go, 16 bytes each, to where the bucket continues in memory (what
        // is left of the slot, under eight pairs, is carried into the next tile)
        {
            const uint32_t total = su[256];
            for (uint32_t u = threadIdx.x; u < total; u += 1024u) {
                const uint32_t a = uown[u];
                if (u < uend[a]) pairs[gun[a] + u] = stage[u];
            }
        }
        if (threadIdx.x < 256 && carryn[threadIdx.x]) carry[threadIdx.x] = stage[uend[threadIdx.x]];
        // (no barrier here: the next tile's ranks go to lcount, which wave 0 cleared before the second barrier above, and
        // nobody writes the stage or the bucket tables again before every wave has passed the next tile's first barrier)
    }
    __syncthreads();
    if (threadIdx.x < 256) {                                     // the last unit of the workgroup's share of a bucket: padded with zero pairs
        const uint32_t k = carryn[threadIdx.x];
        if (k) {
            const uint4 v = carry[threadIdx.x];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t m[4];
#pragma unroll
            for (uint32_t d = 0; d < 4; ++d) m[d] = k >= 2u * d + 2u ? w[d] : k == 2u * d + 1u ? (w[d] & 0xFFFFu) : 0u;
            pairs[cursor[threadIdx.x]] = make_uint4(m[0], m[1], m[2], m[3]);
            atomicAdd(&holes[threadIdx.x], 8u - k);
        }
    }
}

// Counter fields as in hist_o1_kernel (two 14-bit counters under two guard bits each per LDS word), but the PREVIOUS byte is
// the major coordinate here: slot = previous << 8 | (symbol ^ mix(previous)), so that the 256 counters of one (bucket, previous
// byte) row sit in 256 consecutive words of the image and the reduce kernel writes whole rows of the 64-bit counters.
__device__ __forceinline__ void h2p_fixup(uint32_t *h, unsigned long long *mine, uint32_t slot) {
    atomicSub(&h[slot & 0x7FFFu], (slot >> 15) ? 0x40000000u : 0x4000u);
    const uint32_t prev = slot >> 8, sym = (slot & 255u) ^ hist_mix(prev);
    atomicAdd(&mine[prev * 256u + sym], 16384ull);
}
// eight pairs of one vector: a dword holds two pairs, each previous byte | symbol << 8
__device__ __forceinline__ void h2p_add8(uint32_t *h, unsigned long long *mine, const uint4 &x4) {
    const uint32_t x[4] = {x4.x, x4.y, x4.z, x4.w};
    uint32_t slot[8], old[8], inc[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t m = x[k] ^ ((x[k] << 3) & 0xF8F8F8F8u);                    // every byte ^ itself << 3
        const uint32_t y = x[k] ^ (m << 8);                                       // bytes 1 and 3: symbol ^ mix(previous)
        const uint32_t xm = x[k] & 0x7F7F7F7Fu;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 2 * k + j;
            slot[i] = __builtin_amdgcn_perm(xm, y, j ? 0x0C0C0603u : 0x0C0C0401u);   // (previous & 0x7F) << 8 | mixed byte
            asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(inc[i]) : "v"(__builtin_amdgcn_ubfe(x[k], 16 * j + 7, 1)), "s"(0xFFFFu));
            old[i] = atomicAdd(&h[slot[i]], inc[i]);
        }
    }
    uint32_t newly = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) newly |= (old[i] + inc[i]) ^ old[i];
    if (newly & 0xC000C000u) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (((old[i] + inc[i]) ^ old[i]) & 0xC000C000u) h2p_fixup(h, mine, slot[i] | (inc[i] != 1u ? 0x8000u : 0u));
    }
}

// An item = a share of one bucket (the bucket split evenly into ceil(length / 4 Mi) items, whole units each).  Its 32 768
// LDS words go to image[item] as plain stores; hist2_reduce_kernel adds the images of a bucket to the 64-bit counters (one
// 64-bit global atomic per live counter and item was half of this kernel's time: 33 M of them per 2 GiB slab).
__global__ __launch_bounds__(HIST_THREADS) void hist2_bucket_kernel(const uint4 *__restrict__ pairs, const unsigned long long *__restrict__ bstart,
                                                                   const uint32_t *__restrict__ item_start, unsigned long long *counts,
                                                                   uint32_t *image, const uint32_t *ctl) {
    if (ctl[1] != H2_SEL_PARTITION) return;
    if (blockIdx.x >= item_start[256]) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    for (int i = threadIdx.x; i < 32768 / 4; i += HIST_THREADS) reinterpret_cast<uint4 *>(h)[i] = make_uint4(0, 0, 0, 0);
    uint32_t a = 0;
#pragma unroll
    for (uint32_t step = 128; step; step >>= 1)
        if (item_start[a + step] <= blockIdx.x) a += step;
    unsigned long long *mine = counts + (size_t(a) << 16);       // the bucket's 65 536 counters: key = bucket << 16 | pair
    const unsigned long long ub = bstart[a] >> 3, ue = bstart[a + 1] >> 3;       // in units (the bounds are multiples of eight pairs)
    const uint32_t nitems = item_start[a + 1] - item_start[a], mine_i = blockIdx.x - item_start[a];
    const unsigned long long per = (ue - ub + nitems - 1) / nitems;
    const unsigned long long v0 = ub + mine_i * per < ue ? ub + mine_i * per : ue, v1 = v0 + per < ue ? v0 + per : ue;
    __syncthreads();
    // two vectors per trip, the next two already on their way
    unsigned long long v = v0 + threadIdx.x;
    uint4 na = make_uint4(0, 0, 0, 0), nb = make_uint4(0, 0, 0, 0);
    if (v < v1) na = pairs[v];
    if (v + HIST_THREADS < v1) nb = pairs[v + HIST_THREADS];
    for (; v < v1; v += 2 * HIST_THREADS) {
        const uint4 xa = na, xb = nb;
        const bool has_b = v + HIST_THREADS < v1;
        if (v + 2 * HIST_THREADS < v1) na = pairs[v + 2 * HIST_THREADS];
        if (v + 3 * HIST_THREADS < v1) nb = pairs[v + 3 * HIST_THREADS];
        h2p_add8(h, mine, xa);
        if (has_b) h2p_add8(h, mine, xb);
    }
    __syncthreads();
    uint4 *dst = reinterpret_cast<uint4 *>(image + size_t(blockIdx.x) * 32768u);
    for (uint32_t i = threadIdx.x; i < 32768u / 4u; i += HIST_THREADS) dst[i] = reinterpret_cast<const uint4 *>(h)[i];
}

// counts[bucket << 16 | previous << 8 | symbol] += the fields of the bucket's items.  A block = 1024 words of one bucket =
// eight (previous & 0x7F) rows, both halves: plain loads and stores (nobody else touches the counters while this runs), whole
// 2 KiB rows of counters per 256 threads.  The zero pairs that padded the workgroups' last units come off the pair (0, 0).
__global__ __launch_bounds__(1024) void hist2_reduce_kernel(const uint32_t *__restrict__ image, const uint32_t *__restrict__ item_start,
                                                            const uint32_t *__restrict__ holes, unsigned long long *counts, const uint32_t *ctl) {
    if (ctl[1] != H2_SEL_PARTITION) return;
    const uint32_t a = blockIdx.x >> 5, w = (blockIdx.x & 31u) * 1024u + threadIdx.x;
    const uint32_t i0 = item_start[a], i1 = item_start[a + 1];
    if (i0 == i1) return;
    unsigned long long lo = 0, hi = 0;
    for (uint32_t i = i0; i < i1; ++i) {
        const uint32_t v = image[size_t(i) * 32768u + w];
        lo += v & 0xFFFFu;
        hi += v >> 16;
    }
    if (w == 0) lo -= holes[a];                                  // (with the 16384-credits already in the counter this cannot go below zero)
    unsigned long long *mine = counts + (size_t(a) << 16);
    const uint32_t p0 = w >> 8, p1 = p0 | 0x80u, low = w & 255u;
    if (lo) mine[p0 * 256u + (low ^ hist_mix(p0))] += lo;
    if (hi) mine[p1 * 256u + (low ^ hist_mix(p1))] += hi;
}

// Conservation (order 1 has the same in hist_reduce_kernel): whatever path each slab took, the 2^24 counters must
// add up to the n positions counted (src/main.cpp:176-178: the reference's counts add up to the file size).  A unit the scatter
// lost, a packed field that spilled into its neighbour, a wrong hole correction or a key dropped by the tag cache all change
// the total.  One pass over 128 MiB of counters (~40 us), the last block compares and sets the workspace's status word.
// ctl[8..9] = running total, ctl[10] = blocks done (the 64-byte control block is zeroed at the start of every call).
__global__ __launch_bounds__(1024) void hist2_total_kernel(const unsigned long long *__restrict__ counts, unsigned long long n, uint32_t *ctl) {
    __shared__ unsigned long long part[16];
    unsigned long long s = 0;
    for (size_t i = size_t(blockIdx.x) * 1024u + threadIdx.x; i < (size_t(1) << 24); i += size_t(gridDim.x) * 1024u) s += counts[i];   // (8-byte loads: the caller's array need not be 16-byte aligned)
#pragma unroll
    for (int d = 32; d; d >>= 1) s += __shfl_down(s, d);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < 16; ++i) t += part[i];
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(ctl + 8);
        atomicAdd(acc, t);
        __threadfence();
        if (atomicAdd(ctl + 10, 1u) == gridDim.x - 1) {
            __threadfence();
            if (atomicAdd(acc, 0ull) != n) atomicExch(reinterpret_cast<int *>(ctl), MHK_STATUS_CORRUPT);
        }
    }
}
static hipError_t launch_hist2_total(const unsigned long long *d_counts, uint64_t n, uint32_t *ctl, hipStream_t st) {
    if (!ctl) return hipSuccess;                               // (no workspace: nowhere to report to)
    hipLaunchKernelGGL(hist2_total_kernel, dim3(512), dim3(1024), 0, st, d_counts, (unsigned long long)n, ctl);
    return hipGetLastError();
}

size_t hist2_workspace_bytes(uint64_t n) { return n < H2P_MIN ? 64 : hist2_geom(n, H2P_SLAB).total; }

static hipError_t launch_tag(const uint8_t *base, uint64_t n_total, uint64_t lo, uint64_t hi, uint32_t ctx0, unsigned long long *d_counts,
                             uint32_t *ctl, int mode, hipStream_t st) {
    if (hi <= lo) return hipSuccess;
    const uint64_t nvec = (hi - lo) >> 4;
    const uint64_t want = (nvec + H2_THREADS - 1) / H2_THREADS;
    int grid = int(want < 1 ? 1 : (want > uint64_t(cu_count()) ? uint64_t(cu_count()) : want));
    if (mode == H2_MODE_SAMPLE && grid > 64) grid = 64;      // 64 KiB per workgroup says enough, and every workgroup flushes a whole table
    hipLaunchKernelGGL(hist_o2_kernel, dim3(grid), dim3(H2_THREADS), H2_LDS_BYTES, st, base, n_total, lo, hi, ctx0, d_counts, ctl, mode);
    return hipGetLastError();
}

hipError_t launch_hist_o2(const uint8_t *d_data, uint64_t n, uint32_t ctx0, unsigned long long *d_counts, void *d_ws, size_t ws_bytes,
                          hipStream_t st) {
    hipError_t e = hipMemsetAsync(d_counts, 0, (size_t(1) << 24) * sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    const bool ws_ok = d_ws && (reinterpret_cast<uintptr_t>(d_ws) & 255u) == 0 && ws_bytes >= 64;
    uint32_t *ctl = ws_ok ? static_cast<uint32_t *>(d_ws) : nullptr;
    if (ctl) { e = hipMemsetAsync(ctl, 0, 64, st); if (e != hipSuccess) return e; }
    if (n == 0) return hipSuccess;
    e = once_per_device(&DeviceState::hist2_ready, [] {
        hipError_t r = allow_lds(reinterpret_cast<const void *>(hist_o2_kernel), H2_LDS_BYTES);
        if (r == hipSuccess) r = allow_lds(reinterpret_cast<const void *>(hist2_scatter_kernel), H2P_SCATTER_LDS);
        if (r == hipSuccess) r = allow_lds(reinterpret_cast<const void *>(hist2_count_kernel), H2P_COUNT_LDS);
        if (r == hipSuccess) r = allow_lds(reinterpret_cast<const void *>(hist2_bucket_kernel), HIST_LDS_BYTES);
        return r;
    });
    if (e != hipSuccess) return e;
    uint64_t slab = H2P_SLAB;
    uint32_t force = 0;
#ifdef MH_EXP_PROBES                         /* diagnostic library only: tests force a path and shrink the slabs */
    if (const char *s = getenv("MH_HIST2_SLAB")) { const uint64_t v = strtoull(s, nullptr, 0); if (v >= H2P_MIN && v <= H2P_SLAB && v % H2P_TILE == 0) slab = v; }
    if (const char *s = getenv("MH_HIST2_FORCE")) force = uint32_t(atoi(s)) & 3u;
#endif
    const H2Geom g = hist2_geom(n, slab);
    if (!ws_ok || ws_bytes < g.total || n < H2P_MIN) {
        e = launch_tag(d_data, n, 0, n, ctx0, d_counts, ctl, H2_MODE_PLAIN, st);
        if (e == hipSuccess && ctl) e = launch_set_word(ctl + 2, H2_SEL_CACHE, st);
        if (e == hipSuccess) e = launch_hist2_total(d_counts, n, ctl, st);
        return e;
    }
    unsigned char *ws = static_cast<unsigned char *>(d_ws);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(ws + g.off_cnt);
    unsigned long long *off = reinterpret_cast<unsigned long long *>(ws + g.off_off);
    unsigned long long *bstart = reinterpret_cast<unsigned long long *>(ws + g.off_bstart);
    uint32_t *items = reinterpret_cast<uint32_t *>(ws + g.off_items);
    uint32_t *image = reinterpret_cast<uint32_t *>(ws + g.off_image);
    uint32_t *holes = reinterpret_cast<uint32_t *>(ws + g.off_holes);
    uint4 *pairs = reinterpret_cast<uint4 *>(ws + g.off_pairs);
    for (uint64_t lo = 0; lo < n; lo += slab) {
        const uint64_t hi = n - lo < slab ? n : lo + slab;
        if (hi - lo < H2P_MIN) {             // a short last slab (never the first: n >= H2P_MIN): the path of the slab before it
            e = launch_tag(d_data, n, lo, hi, ctx0, d_counts, ctl, H2_MODE_IF_CACHE, st);
            if (e != hipSuccess) return e;
        } else {
            const uint64_t mid = lo + H2P_SAMPLE;
            e = launch_tag(d_data, n, lo, mid, ctx0, d_counts, ctl, H2_MODE_SAMPLE, st);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(hist2_select_kernel, dim3(1), dim3(1), 0, st, ctl, (unsigned long long)H2P_SAMPLE, force);
            e = launch_tag(d_data, n, mid, hi, ctx0, d_counts, ctl, H2_MODE_IF_CACHE, st);
            if (e != hipSuccess) return e;
        }
        const uint64_t mid = hi - lo < H2P_MIN ? lo : lo + H2P_SAMPLE;
        const uint32_t ntiles = uint32_t((hi - mid) / H2P_TILE);
        if (ntiles) {
            hipLaunchKernelGGL(hist2_count_kernel, dim3(H2P_G), dim3(1024), H2P_COUNT_LDS, st, d_data, n, mid, ntiles, ctx0, cnt, ctl);
            hipLaunchKernelGGL(hist2_offsets_kernel, dim3(1), dim3(1024), 0, st, cnt, off, bstart, items, holes, ctl);
            hipLaunchKernelGGL(hist2_scatter_kernel, dim3(H2P_G), dim3(1024), H2P_SCATTER_LDS, st, d_data, n, mid, ntiles, ctx0, off, pairs, holes, ctl);
            const uint32_t grid = uint32_t((uint64_t(ntiles) * H2P_TILE + H2P_PAD_PAIRS + H2P_CHUNK - 1) / H2P_CHUNK) + 256u;     // <= g.max_items
            hipLaunchKernelGGL(hist2_bucket_kernel, dim3(grid), dim3(HIST_THREADS), HIST_LDS_BYTES, st, pairs, bstart, items, d_counts, image, ctl);
            hipLaunchKernelGGL(hist2_reduce_kernel, dim3(256 * 32), dim3(1024), 0, st, image, items, holes, d_counts, ctl);
        }
        e = launch_tag(d_data, n, mid + uint64_t(ntiles) * H2P_TILE, hi, ctx0, d_counts, ctl, H2_MODE_IF_PARTITION, st);
        if (e != hipSuccess) return e;
    }
    return launch_hist2_total(d_counts, n, ctl, st);
}

}  // namespace mhk
