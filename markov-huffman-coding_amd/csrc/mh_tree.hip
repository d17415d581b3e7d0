// mh_tree.hip — per-context Huffman tree build and table packing on the device.
//
//   tree_build_kernel   one wave per context: exact emulation of the reference's heap
//                       (src/min_pq.tpp:4-52) and merge rule (src/huffman.cpp:131-164), the heap held in
//                       the wave's registers and driven by scalar code (RegHeap); then all lanes derive
//                       depths, codewords (src/huffman.cpp:97-123) and the encode tables, and size the
//                       decode tables for every primary width.
//   tree_pack_kernel    one workgroup per context: fills the two decode-table levels and the walk tree
//                       for the layout the host picked from those sizes (same rule as Model::pack()).
//
// The result is bit-identical to the host build in mh_model.cpp (tests/test_gpu_parity.py compares the
// images); it exists so that histogram -> tables -> encode runs without the 512 KiB of counts going to
// the host and ~5 ms of host work in the middle of the pipeline.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mh_kernels.h"
#include "mh_model.hpp"

namespace mhk {

using mh::DEC16_LEAF;
using mh::DEC16_NULL;
using mh::TREE_LEAF;
using mh::TREE_STRIDE;

constexpr uint16_t NONE = 0xFFFF;

// ------------------------------------------------------------------------------------------------
// The reference's binary heap (src/min_pq.tpp:4-52), emulated comparison for comparison, kept in the
// wave's REGISTERS: heap level L (positions 0 .. 2^L - 1) lives in one VGPR, position = lane (level 7 takes
// two, level 8 is the single index 255), so an entry is reached with v_readlane / v_writelane at a
// wave-uniform position and every level of a sift is a handful of scalar instructions — no LDS round trip
// and no divergent lane (the first version ran the heap in LDS on lane 0: ~0.8 ms for 256 contexts, bound
// by the LDS latency of ~4000 dependent accesses per context).  Each entry carries its key (the weight;
// 32 bits when the context's total fits, else 64), and `item` = node id | subtree height << 16, so the
// merge loop (src/huffman.cpp:143-151) needs nothing back from memory.
// All values handled here are wave-uniform; sifts are unrolled over the levels by template recursion.
template <bool K64>
struct RegHeap {
    uint32_t klo[10], khi[10], itm[10];   // slot: levels 0..6 -> 0..6, level 7 -> 7 (positions 0..63) and 8 (64..127), level 8 -> 9
    uint32_t hn = 0;

    static __device__ __forceinline__ uint32_t uni(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
    static __device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return uint32_t(__builtin_amdgcn_readlane(int(v), int(lane))); }
    static __device__ __forceinline__ uint32_t wl(uint32_t val, uint32_t lane, uint32_t old) {
        // (this clang has no __builtin_amdgcn_writelane; both scalar operands are wave-uniform by construction)
        // the lane select goes through M0: two different SGPR operands would exceed gfx9's constant-bus limit.
        // (M0 is a reserved register — the compiler sets it itself before every use of its own — and clang
        // warns about naming it as a clobber; it is named anyway so that no operand is ever placed in it.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        // (readfirstlane: an "s" operand the compiler holds in a VGPR would be passed as that VGPR)
        asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(uni(val)), "s"(uni(lane)) : "m0");
#pragma clang diagnostic pop
        return old;
    }
    // strict a < b on (hi, lo) pairs
    static __device__ __forceinline__ bool lt(uint32_t alo, uint32_t ahi, uint32_t blo, uint32_t bhi) {
        if (K64) return ahi < bhi || (ahi == bhi && alo < blo);
        return alo < blo;
    }
    template <int L>
    __device__ __forceinline__ void get(uint32_t pos, uint32_t &lo, uint32_t &hi, uint32_t &it) const {
        if constexpr (L < 7) {
            lo = rl(klo[L], pos); hi = K64 ? rl(khi[L], pos) : 0u; it = rl(itm[L], pos);
        } else if constexpr (L == 7) {
            const uint32_t ln = pos & 63u;
            const bool up = pos >= 64u;
            lo = up ? rl(klo[8], ln) : rl(klo[7], ln);
            hi = K64 ? (up ? rl(khi[8], ln) : rl(khi[7], ln)) : 0u;
            it = up ? rl(itm[8], ln) : rl(itm[7], ln);
        } else {
            lo = rl(klo[9], 0); hi = K64 ? rl(khi[9], 0) : 0u; it = rl(itm[9], 0);
        }
    }
    template <int L>
    __device__ __forceinline__ void put(uint32_t pos, uint32_t lo, uint32_t hi, uint32_t it) {
        if constexpr (L < 7) {
            klo[L] = wl(lo, pos, klo[L]); if (K64) khi[L] = wl(hi, pos, khi[L]); itm[L] = wl(it, pos, itm[L]);
        } else if constexpr (L == 7) {
            const uint32_t ln = pos & 63u;
            if (pos >= 64u) { klo[8] = wl(lo, ln, klo[8]); if (K64) khi[8] = wl(hi, ln, khi[8]); itm[8] = wl(it, ln, itm[8]); }
            else { klo[7] = wl(lo, ln, klo[7]); if (K64) khi[7] = wl(hi, ln, khi[7]); itm[7] = wl(it, ln, itm[7]); }
        } else {
            klo[9] = wl(lo, 0, klo[9]); if (K64) khi[9] = wl(hi, 0, khi[9]); itm[9] = wl(it, 0, itm[9]);
        }
    }
    // [r3] A compact variant was measured against this one: four register sets of 64 slots (children always side by side
    // in one set), sifts as rolled loops, every write one asm statement that branches inside so that no join sees a
    // register modified on one path only — 4 400 instructions of kernel instead of 16 000, all tests green, and 1.1 ms per
    // launch instead of 0.74.  The time is the length of the dependent scalar chain (~250 instructions per heap operation
    // either way, one wave per CU, nothing to hide behind), not the code size; the unrolled form below stays.
    // Both sifts are written as a read-only phase that follows the whole path with scalar selects, and a
    // write phase that stores ONE entry per level unconditionally — the moved entry, the sifted key, or
    // what the slot held anyway.  (Branches with early exits made the compiler copy all thirty heap registers
    // at every join: ~500 instructions per heap operation, no faster than the first version's LDS heap.)

    // src/min_pq.tpp:4-7 + 29-36: the new entry, appended at (L, pos), moves up while its parent's key is
    // STRICTLY greater.  Heap order along the root path means the parents that move form one bottom segment.
    template <int L>
    __device__ __forceinline__ void swim_flat(uint32_t pos, uint32_t lo, uint32_t hi, uint32_t it) {
        uint32_t alo[L + 1], ahi[L + 1], ait[L + 1];            // ancestors: a*[l] = entry at level l on the root path (l < L)
        uint32_t moves = 0;                                      // parents that move down (counted from the bottom)
        bool going = true;
        alo[L] = lo; ahi[L] = hi; ait[L] = it;
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            get_dyn(l, pos >> (L - l), alo[l], ahi[l], ait[l]);
            going = going && lt(lo, hi, alo[l], ahi[l]);         // parent > key
            moves += going ? 1u : 0u;
        }
        const uint32_t f = uint32_t(L) - moves;                  // level where the new entry comes to rest
#pragma unroll
        for (int l = L; l >= 0; --l) {
            // level l gets: its parent's entry if the parent moved into it, the new entry at level f, else itself
            const bool from_parent = uint32_t(l) > f;
            const bool self = uint32_t(l) < f;
            const uint32_t plo = l > 0 ? alo[l > 0 ? l - 1 : 0] : lo, phi = l > 0 ? ahi[l > 0 ? l - 1 : 0] : hi,
                           pit = l > 0 ? ait[l > 0 ? l - 1 : 0] : it;
            const uint32_t wlo = from_parent ? plo : self ? alo[l] : lo;
            const uint32_t whi = from_parent ? phi : self ? ahi[l] : hi;
            const uint32_t wit = from_parent ? pit : self ? ait[l] : it;
            put_dyn(l, pos >> (L - l), wlo, whi, wit);
        }
    }
    // [r3] the same with an early exit: most entries come to rest one or two levels above where they were appended, and the
    // flat form reads and rewrites the whole root path every time.  One comparison chain, no write until the resting level is
    // known; then only the levels that change are written (switch on the number of moves: every case writes a fixed set).
    template <int L>
    __device__ __forceinline__ void swim_short(uint32_t pos, uint32_t lo, uint32_t hi, uint32_t it) {
        uint32_t alo[L + 1], ahi[L + 1], ait[L + 1];
        uint32_t moves = 0;
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            get_dyn(l, pos >> (L - l), alo[l], ahi[l], ait[l]);
            if (!lt(lo, hi, alo[l], ahi[l])) break;              // parent <= key: the entry rests at level l + 1
            ++moves;
        }
        // parents at levels L-1 .. L-moves move down one level each; the new entry goes to level L - moves
#pragma unroll
        for (int m = 0; m < L; ++m) {
            if (uint32_t(m) < moves) put_dyn(L - m, pos >> m, alo[L - 1 - m], ahi[L - 1 - m], ait[L - 1 - m]);
        }
        put_any_level<L>(L - moves, pos >> moves, lo, hi, it);
    }
    template <int L>
    __device__ __forceinline__ void put_any_level(uint32_t level, uint32_t pos, uint32_t lo, uint32_t hi, uint32_t it) {
        switch (level) {
            case 0: put<0>(pos, lo, hi, it); break;
            case 1: if (L >= 1) put<1>(pos, lo, hi, it); break;
            case 2: if (L >= 2) put<2>(pos, lo, hi, it); break;
            case 3: if (L >= 3) put<3>(pos, lo, hi, it); break;
            case 4: if (L >= 4) put<4>(pos, lo, hi, it); break;
            case 5: if (L >= 5) put<5>(pos, lo, hi, it); break;
            case 6: if (L >= 6) put<6>(pos, lo, hi, it); break;
            case 7: if (L >= 7) put<7>(pos, lo, hi, it); break;
            default: if (L >= 8) put<8>(pos, lo, hi, it); break;
        }
    }
    // level known at compile time through the unrolled loops above: these forward to get<>/put<>
    __device__ __forceinline__ void get_dyn(int l, uint32_t pos, uint32_t &lo, uint32_t &hi, uint32_t &it) const {
        switch (l) {
            case 0: get<0>(pos, lo, hi, it); break; case 1: get<1>(pos, lo, hi, it); break; case 2: get<2>(pos, lo, hi, it); break;
            case 3: get<3>(pos, lo, hi, it); break; case 4: get<4>(pos, lo, hi, it); break; case 5: get<5>(pos, lo, hi, it); break;
            case 6: get<6>(pos, lo, hi, it); break; case 7: get<7>(pos, lo, hi, it); break; default: get<8>(pos, lo, hi, it); break;
        }
    }
    __device__ __forceinline__ void put_dyn(int l, uint32_t pos, uint32_t lo, uint32_t hi, uint32_t it) {
        switch (l) {
            case 0: put<0>(pos, lo, hi, it); break; case 1: put<1>(pos, lo, hi, it); break; case 2: put<2>(pos, lo, hi, it); break;
            case 3: put<3>(pos, lo, hi, it); break; case 4: put<4>(pos, lo, hi, it); break; case 5: put<5>(pos, lo, hi, it); break;
            case 6: put<6>(pos, lo, hi, it); break; case 7: put<7>(pos, lo, hi, it); break; default: put<8>(pos, lo, hi, it); break;
        }
    }
    __device__ __forceinline__ void push(uint32_t lo, uint32_t hi, uint32_t it) {     // src/min_pq.tpp:4-7
        const uint32_t i = hn++;
        const uint32_t level = 31u - uint32_t(__builtin_clz(i + 1u));
        const uint32_t pos = i + 1u - (1u << level);
        switch (level) {
            case 0: swim_short<0>(pos, lo, hi, it); break;
            case 1: swim_short<1>(pos, lo, hi, it); break;
            case 2: swim_short<2>(pos, lo, hi, it); break;
            case 3: swim_short<3>(pos, lo, hi, it); break;
            case 4: swim_short<4>(pos, lo, hi, it); break;
            case 5: swim_short<5>(pos, lo, hi, it); break;
            case 6: swim_short<6>(pos, lo, hi, it); break;
            case 7: swim_short<7>(pos, lo, hi, it); break;
            default: swim_short<8>(pos, lo, hi, it); break;
        }
    }
    // src/min_pq.tpp:38-52: the hole at the root takes the smaller child — the right one only when STRICTLY
    // smaller than the left — while that child is STRICTLY smaller than the sinking key.
    // LMAX = deepest level that still holds entries: the path is followed, and written back, no further (the heap
    // shrinks from 256 entries to one during the merge: on average two levels less than the full eight)
    template <int LMAX>
    __device__ __forceinline__ void sink_flat(uint32_t lo, uint32_t hi, uint32_t it) {
        uint32_t pos[LMAX + 1], clo[LMAX + 1], chi[LMAX + 1], cit[LMAX + 1];   // the min-child path: entry (clo, chi, cit)[l] at (l, pos[l])
        pos[0] = 0; clo[0] = lo; chi[0] = hi; cit[0] = it;       // level 0 is the hole itself
        uint32_t moves = 0;                                      // children that move up
        bool going = true;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            const uint32_t li = ((2u << l) - 1u) + 2u * pos[l];  // heap index of the left child
            uint32_t llo, lhi, lit, rlo, rhi, rit;
            get_dyn(l + 1, 2u * pos[l], llo, lhi, lit);
            get_dyn(l + 1, 2u * pos[l] + 1u, rlo, rhi, rit);     // (level 8 holds one entry: both reads return it, the right one is never valid)
            const bool right = li + 1u < hn && lt(rlo, rhi, llo, lhi);
            clo[l + 1] = right ? rlo : llo; chi[l + 1] = right ? rhi : lhi; cit[l + 1] = right ? rit : lit;
            pos[l + 1] = 2u * pos[l] + (right ? 1u : 0u);
            going = going && li < hn && lt(clo[l + 1], chi[l + 1], lo, hi);
            moves += going ? 1u : 0u;
        }
#pragma unroll
        for (int l = 0; l <= LMAX; ++l) {
            // level l gets: its child's entry if that child moved up, the sinking entry at level `moves`, else itself
            const bool from_child = uint32_t(l) < moves;
            const bool self = uint32_t(l) > moves;
            const int c = l < LMAX ? l + 1 : LMAX;
            const uint32_t wlo = from_child ? clo[c] : self ? clo[l] : lo;
            const uint32_t whi = from_child ? chi[c] : self ? chi[l] : hi;
            const uint32_t wit = from_child ? cit[c] : self ? cit[l] : it;
            put_dyn(l, pos[l], wlo, whi, wit);
        }
    }
    // src/min_pq.tpp:9-15: returns the minimum's item and key; the last entry sinks from the root
    __device__ __forceinline__ uint32_t pop(uint32_t &mlo, uint32_t &mhi) {
        uint32_t top;
        get<0>(0, mlo, mhi, top);
        --hn;
        const uint32_t level = 31u - uint32_t(__builtin_clz(hn + 1u));
        const uint32_t pos = hn + 1u - (1u << level);
        uint32_t lo, hi, it;
        switch (level) {                                          // the last entry leaves (level, pos) and sinks from the root
            case 0: get<0>(pos, lo, hi, it); sink_flat<0>(lo, hi, it); break;
            case 1: get<1>(pos, lo, hi, it); sink_flat<1>(lo, hi, it); break;
            case 2: get<2>(pos, lo, hi, it); sink_flat<2>(lo, hi, it); break;
            case 3: get<3>(pos, lo, hi, it); sink_flat<3>(lo, hi, it); break;
            case 4: get<4>(pos, lo, hi, it); sink_flat<4>(lo, hi, it); break;
            case 5: get<5>(pos, lo, hi, it); sink_flat<5>(lo, hi, it); break;
            case 6: get<6>(pos, lo, hi, it); sink_flat<6>(lo, hi, it); break;
            case 7: get<7>(pos, lo, hi, it); sink_flat<7>(lo, hi, it); break;
            default: get<8>(pos, lo, hi, it); sink_flat<8>(lo, hi, it); break;
        }
        return top;
    }
};

struct TreeLds {
    uint16_t *left, *right, *parent, *height;
    uint8_t *sym;
    unsigned long long *weight;
};

// src/huffman.cpp:131-164 for one context: nleaf leaves are already laid out (ascending symbol order) in
// the node arrays.  Returns nn and the root through the two references; all lanes run the same scalar code
// and lane 0 writes the node arrays.
template <bool K64>
__device__ __forceinline__ void merge_context(const TreeLds &t, uint32_t nleaf, uint32_t lane, uint32_t &nn_out, uint32_t &root_out,
                                              uint32_t &single_out) {
    RegHeap<K64> h;
#pragma unroll
    for (int i = 0; i < 10; ++i) { h.klo[i] = 0; h.khi[i] = 0; h.itm[i] = 0; }
    for (uint32_t i = 0; i < nleaf; ++i) {                            // :134-138 ascending symbol order
        const unsigned long long w = t.weight[i];
        h.push(RegHeap<K64>::uni(uint32_t(w)), RegHeap<K64>::uni(uint32_t(w >> 32)), i);   // height 0
    }
    uint32_t nn = nleaf;
    while (h.hn > 1) {                                                 // :143-151
        uint32_t alo, ahi, blo, bhi;
        uint32_t a = h.pop(alo, ahi), b = h.pop(blo, bhi);
        if ((a >> 16) > (b >> 16)) {                                   // :147-149 the lower subtree goes left
            uint32_t x = a; a = b; b = x;
        }
        const unsigned long long w = ((unsigned long long)(ahi) << 32 | alo) + ((unsigned long long)(bhi) << 32 | blo);
        const uint32_t ha = a >> 16, hb = b >> 16, hnew = (ha > hb ? ha : hb) + 1u;
        const uint32_t ia = a & 0xFFFFu, ib = b & 0xFFFFu;
        if (lane == 0) {
            t.left[nn] = uint16_t(ia); t.right[nn] = uint16_t(ib); t.parent[nn] = NONE; t.sym[nn] = 0;
            t.weight[nn] = w; t.height[nn] = uint16_t(hnew);
            t.parent[ia] = t.parent[ib] = uint16_t(nn);
        }
        h.push(uint32_t(w), uint32_t(w >> 32), nn | (hnew << 16));
        ++nn;
    }
    uint32_t rlo, rhi;
    const uint32_t root = h.pop(rlo, rhi) & 0xFFFFu;                   // :152
    single_out = 0;
    if (nleaf == 1) {                                                  // :154-162 one-symbol context
        if (lane == 0) {
            const uint8_t s = t.sym[root];
            for (int k = 0; k < 2; ++k) {
                t.left[nn + k] = t.right[nn + k] = NONE; t.parent[nn + k] = uint16_t(root); t.height[nn + k] = 0; t.sym[nn + k] = s;
                t.weight[nn + k] = t.weight[root];
            }
            t.left[root] = uint16_t(nn); t.right[root] = uint16_t(nn + 1);
            t.height[root] = 1;
        }
        nn += 2;
        single_out = 1;
    }
    nn_out = nn;
    root_out = root;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tree_build_kernel(const unsigned long long *__restrict__ counts, TreeBuildOut o) {
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    __shared__ unsigned long long cnt[256];
    __shared__ uint16_t left[TB_NODE_STRIDE], right[TB_NODE_STRIDE], parent[TB_NODE_STRIDE], height[TB_NODE_STRIDE];
    __shared__ uint8_t sym[TB_NODE_STRIDE];
    __shared__ unsigned long long weight[TB_NODE_STRIDE];
    __shared__ uint8_t olen[256];
    __shared__ unsigned long long ocode[256];
    __shared__ uint32_t prof[9];
    __shared__ uint32_t s_nn, s_root, s_single, s_ntab8, s_maxlen;

    unsigned long long wsum = 0;
    for (uint32_t i = lane; i < 256; i += 64) {
        cnt[i] = counts[size_t(c) * 256 + i];
        wsum += cnt[i];
        olen[i] = 0;
        ocode[i] = 0;
    }
    for (int d = 32; d >= 1; d >>= 1) wsum += __shfl_xor(wsum, d);
    if (lane < 9) prof[lane] = 0;
    if (lane == 0) { s_ntab8 = 0; s_maxlen = 0; s_single = 0; }
    __syncthreads();

    // ---- leaves, in ascending symbol order (src/huffman.cpp:134-138): node id = rank among the non-zero counts
    uint32_t nleaf = 0;
    for (uint32_t i = 0; i < 4; ++i) {
        const uint32_t s = i * 64 + lane;
        const bool nz = cnt[s] != 0;
        const unsigned long long m = __ballot(nz);
        const uint32_t pos = nleaf + __popcll(m & ((1ull << lane) - 1ull));
        if (nz) {
            left[pos] = right[pos] = NONE; parent[pos] = NONE; height[pos] = 0; sym[pos] = uint8_t(s); weight[pos] = cnt[s];
        }
        nleaf += __popcll(m);
    }
    __syncthreads();
    {
        const TreeLds t{left, right, parent, height, sym, weight};
        uint32_t nn = 0, root = 0xFFFFFFFFu, single = 0;
        if (nleaf > 0) {
            // 32-bit keys when every weight that can appear (the root's is the context total) fits
            const bool wide = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(wsum >> 32)))) != 0;
            if (wide) merge_context<true>(t, nleaf, lane, nn, root, single);
            else merge_context<false>(t, nleaf, lane, nn, root, single);
        }
        if (lane == 0) { s_nn = nn; s_root = root; s_single = single; }
    }
    __syncthreads();

    const uint32_t nn = s_nn, root = s_root;
    if (root != 0xFFFFFFFFu) {
        for (uint32_t node = lane; node < nn; node += 64) {
            // walk up: depth, and for a leaf its codeword (last bit first)
            uint32_t d = 0;
            unsigned long long code = 0;
            uint32_t cur = node;
            while (cur != root) {
                const uint32_t p = parent[cur];
                if (right[p] == cur && d < 64) code |= 1ull << d;
                ++d;
                cur = p;
            }
            if (left[node] == NONE) {
                // in the one-symbol case the right leaf is visited last and wins (src/huffman.cpp:115)
                if (!(s_single && node == left[root])) {
                    olen[sym[node]] = uint8_t(d > 255 ? 255 : d);
                    ocode[sym[node]] = d <= 64 ? code : 0;
                    atomicMax(&s_maxlen, d);
                }
            } else if (d <= 8) {
                const uint32_t h = height[node] < o.hcap ? height[node] : o.hcap;
                atomicAdd(&prof[d], 1u << h);
                if (d == 8) atomicAdd(&s_ntab8, 1u);
            }
        }
    }
    __syncthreads();

    // ---- outputs
    uint32_t lenmask = 0;                    // bit l-1 for every code length l < 32 in use, bit 31 for longer ones
    for (uint32_t s = lane; s < 256; s += 64) {
        const uint32_t l = olen[s];
        if (l) lenmask |= 1u << (l < 32u ? l - 1u : 31u);
        const unsigned long long cd = ocode[s];
        o.len8[c * 256 + s] = uint8_t(l);
        o.code64[c * 256 + s] = cd;
        if (o.enc16) {                                       // order 0/1: the encoder's LDS images
            const uint32_t slot = mh::enc_slot((s << 8) | c);
            uint16_t e = 0;
            if (l > uint32_t(mh::ENC16_MAX_LEN)) e = mh::ENC16_ESCAPE;
            else if (l > 0) e = uint16_t((l << 12) | uint32_t(cd));
            o.enc16[slot] = e;
            o.len_slot[slot] = uint8_t(l);
        }
    }
    for (uint32_t i = lane; i < TB_NODE_STRIDE; i += 64) {
        const bool live = i < nn;
        o.node_left[c * TB_NODE_STRIDE + i] = live ? left[i] : NONE;
        o.node_right[c * TB_NODE_STRIDE + i] = live ? right[i] : NONE;
        o.node_sym[c * TB_NODE_STRIDE + i] = live ? sym[i] : 0;
        o.node_height[c * TB_NODE_STRIDE + i] = live ? uint8_t(height[i] > 255 ? 255 : height[i]) : 0;
    }
    for (int d = 32; d >= 1; d >>= 1) lenmask |= __shfl_xor(lenmask, d);
    if (lane == 0) {
        uint32_t *m = o.ctx_meta + c * TB_META_STRIDE;
        m[0] = nn; m[1] = root; m[2] = s_maxlen; m[3] = s_ntab8;
        for (int d = 0; d < 9; ++d) m[4 + d] = prof[d];
        m[13] = uint32_t(wsum); m[14] = uint32_t(wsum >> 32);
        m[15] = nleaf < 2 ? 0u : lenmask;                       // the 1-bit code of a one-symbol context does not count
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tree_pack_body(const TreePackArgs &a, const uint32_t c) {
    const uint32_t tid = threadIdx.x;
    __shared__ uint16_t left[TB_NODE_STRIDE], right[TB_NODE_STRIDE], nid[TB_NODE_STRIDE];
    __shared__ uint8_t sym[TB_NODE_STRIDE], height[TB_NODE_STRIDE];
    __shared__ uint32_t tr[TREE_STRIDE];
    __shared__ uint32_t scan[256];
    const uint32_t *meta = a.ctx_meta + c * TB_META_STRIDE;
    const uint32_t nn = meta[0], root = meta[1];
    for (uint32_t i = tid; i < TB_NODE_STRIDE; i += 256) {
        left[i] = a.node_left[c * TB_NODE_STRIDE + i];
        right[i] = a.node_right[c * TB_NODE_STRIDE + i];
        sym[i] = a.node_sym[c * TB_NODE_STRIDE + i];
        height[i] = a.node_height[c * TB_NODE_STRIDE + i];
    }
    tr[tid] = 0;
    __syncthreads();
    const uint32_t P = a.P, nprim = 1u << P;
    const uint32_t my_base = a.sec_base_in ? a.sec_base_in[c] : a.sec_base_val[c];
    if (tid == 0 && !a.sec_base_in && a.sec_base) a.sec_base[c] = my_base;
    if (root == 0xFFFFFFFFu) {                       // empty context: null tables
        if (tid < nprim) a.prim[(c << P) | tid] = DEC16_NULL;
        if (a.tree) a.tree[c * TREE_STRIDE + tid] = 0;
        return;
    }
    if (tid == 0) {                                   // inner-node ids for the walk: root = 0, the rest in node order
        uint32_t next = 1;
        for (uint32_t i = 0; i < nn; ++i) nid[i] = (left[i] == NONE) ? NONE : (i == root ? 0 : uint16_t(next++));
    }
    __syncthreads();
    auto enc_child = [&](uint32_t ch) -> uint32_t { return left[ch] == NONE ? (TREE_LEAF | sym[ch]) : uint32_t(nid[ch]); };
    for (uint32_t i = tid; i < nn; i += 256)
        if (left[i] != NONE) tr[nid[i]] = (enc_child(right[i]) << 16) | enc_child(left[i]);

    // first level: thread w follows the P bits of w from the root
    uint32_t node = root, depth = 0, tabsize = 0, h = 0;
    if (tid < nprim) {
        while (depth < P && left[node] != NONE) {
            const uint32_t bit = a.lsb ? (tid >> depth) & 1u : (tid >> (P - 1 - depth)) & 1u;
            node = bit ? right[node] : left[node];
            ++depth;
        }
        if (left[node] != NONE) {                     // internal node at depth P
            h = a.direct ? a.H : (height[node] < a.hcap ? height[node] : a.hcap);
            tabsize = 1u << h;
        }
    }
    // exclusive prefix of the table sizes in w order (tables are laid out by increasing w)
    scan[tid] = tabsize;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t v = tid >= d ? scan[tid - d] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const uint32_t off = scan[tid] - tabsize;
    const uint32_t base = my_base;
    if (tid < nprim) {
        uint16_t e;
        if (left[node] == NONE) e = uint16_t(DEC16_LEAF | (depth << 8) | sym[node]);      // a leaf reached at depth <= P fills its whole range
        else if (a.direct) e = uint16_t((base >> a.H) + (off >> a.H));
        else e = uint16_t(((h - 1) << 12) | off);
        a.prim[(c << P) | tid] = e;
        // second level: this thread fills its own table
        for (uint32_t x = 0; x < tabsize; ++x) {
            uint32_t n2 = node, d2 = 0;
            while (d2 < h && left[n2] != NONE) {
                const uint32_t bit = a.lsb ? (x >> d2) & 1u : (x >> (h - 1 - d2)) & 1u;
                n2 = bit ? right[n2] : left[n2];
                ++d2;
            }
            a.sec[base + off + x] = left[n2] == NONE ? uint16_t(DEC16_LEAF | ((P + d2) << 8) | sym[n2]) : uint16_t(nid[n2]);
        }
    }
    __syncthreads();
    if (a.tree) a.tree[c * TREE_STRIDE + tid] = tr[tid];      // (a second packing of the same trees leaves the walk tree alone)
}

__global__ __launch_bounds__(256) void tree_pack_kernel(TreePackArgs a) { tree_pack_body(a, blockIdx.x); }
// [r4] two packings of the same trees in ONE launch (the chunk decoder's tables and the tile decoder's: blocks 0 .. nctx - 1
// and nctx .. 2 nctx - 1): they ran one after the other, 0.04 ms each, with nothing between them but a launch gap
__global__ __launch_bounds__(256) void tree_pack2_kernel(TreePackArgs a, TreePackArgs b, uint32_t nctx) {
    if (blockIdx.x < nctx) tree_pack_body(a, blockIdx.x);
    else tree_pack_body(b, blockIdx.x - nctx);
}

// ---- order 2: the live contexts' tables ("LDS codeword-table staging") --------------------------
// one wave per slot; the last block (slot == nslots) writes the encoder's all-escape row
__global__ __launch_bounds__(64) void o2_hot_pack_kernel(O2HotArgs a) {
    const uint32_t slot = blockIdx.x, lane = threadIdx.x;
    if (slot == a.nslots) { a.hot[slot * 64u + lane] = mh::ENC16_ESCAPE; return; }
    const uint32_t ctx = a.slot_ctx[slot];
    {   // encoder row: id of the symbol -> len << 12 | code, ENC16_ESCAPE for codes over 12 bits, id 63 and unused ids
        uint16_t e = mh::ENC16_ESCAPE;
        if (lane < 63 && a.id_used[lane]) {
            const uint32_t k = ctx * 256u + a.id_sym[lane];
            const uint32_t l = a.len8[k];
            e = l == 0 ? uint16_t(0) : (l > uint32_t(mh::ENC16_MAX_LEN) ? mh::ENC16_ESCAPE : uint16_t((l << 12) | uint32_t(a.code64[k])));
        }
        // stored at column id ^ (id of the context's second byte): the encoders read it there (bank spreading)
        a.hot[slot * 64u + (lane ^ a.slot_id1[slot])] = e;
    }
    if (!a.tprim) return;
    // tile decoder: first level of P bits (lane = window value, LSB-first), uniform second-level tables of 2^H entries;
    // entry = next slot << 16 | 0x8000 | length << 8 | symbol (leaf) or the table id (inner node at depth P)
    const uint32_t P = a.P, H = a.H;
    const uint16_t *left = a.node_left + size_t(ctx) * TB_NODE_STRIDE, *right = a.node_right + size_t(ctx) * TB_NODE_STRIDE;
    const uint8_t *sym = a.node_sym + size_t(ctx) * TB_NODE_STRIDE;
    const uint32_t root = a.ctx_meta[size_t(ctx) * TB_META_STRIDE + 1];
    auto leaf_entry = [&](uint32_t node, uint32_t len) -> uint32_t {
        const uint32_t s = sym[node];
        uint32_t nxt = a.ctx2slot[((ctx & 255u) << 8) | s];
        if (nxt == 0xFFFFu) nxt = 0;                              // no live context follows: only at the very end of a stream
        return (nxt << 16) | DEC16_LEAF | (len << 8) | s;
    };
    for (uint32_t w = lane; w < (1u << P); w += 64u) {            // (P <= 6: one pass)
        uint32_t node = root, depth = 0;
        if (root != 0xFFFFFFFFu)
            while (depth < P && left[node] != NONE) { node = ((w >> depth) & 1u) ? right[node] : left[node]; ++depth; }
        const bool inner = root != 0xFFFFFFFFu && left[node] != NONE;
        const unsigned long long m = __ballot(inner);
        const uint32_t rank = uint32_t(__popcll(m & ((1ull << lane) - 1ull)));
        const uint32_t id = (slot << P) + rank;                   // sparse ids: at most 2^P tables per slot
        uint32_t e = DEC16_NULL;
        if (root != 0xFFFFFFFFu) e = inner ? id : leaf_entry(node, depth);
        a.tprim[(slot << P) + (w ^ (slot & ((1u << P) - 1u)))] = e;       // stored at column w ^ slot: the decoder reads it there (bank spreading)
        if (inner) {
            for (uint32_t x = 0; x < (1u << H); ++x) {
                uint32_t n2 = node, d2 = 0;
                while (d2 < H && left[n2] != NONE) { n2 = ((x >> d2) & 1u) ? right[n2] : left[n2]; ++d2; }
                a.tsec[(size_t(id) << H) + x] = left[n2] == NONE ? leaf_entry(n2, P + d2) : 0u;   // deeper than P + H: unresolved (redo pass)
            }
        }
    }
}

hipError_t launch_o2_hot_pack(const O2HotArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(o2_hot_pack_kernel, dim3(a.nslots + 1), dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_tree_build(const unsigned long long *d_counts, int nctx, const TreeBuildOut &o, hipStream_t st) {
    hipLaunchKernelGGL(tree_build_kernel, dim3(nctx), dim3(64), 0, st, d_counts, o);
    return hipGetLastError();
}

hipError_t launch_tree_pack(const TreePackArgs &a, int nctx, hipStream_t st) {
    hipLaunchKernelGGL(tree_pack_kernel, dim3(nctx), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_tree_pack2(const TreePackArgs &a, const TreePackArgs &b, int nctx, hipStream_t st) {
    hipLaunchKernelGGL(tree_pack2_kernel, dim3(2 * nctx), dim3(256), 0, st, a, b, uint32_t(nctx));
    return hipGetLastError();
}

}  // namespace mhk
