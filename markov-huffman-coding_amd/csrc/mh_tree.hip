// mh_tree.hip — per-context Huffman tree build and table packing on the device.
//
//   tree_build_kernel   one wave per context: exact emulation of the reference's heap
//                       (src/min_pq.tpp:4-52) and merge rule (src/huffman.cpp:131-164) by lane 0 in
//                       LDS; then all lanes derive depths, codewords (src/huffman.cpp:97-123) and the
//                       encode tables, and size the decode tables for every primary width.
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
__global__ __launch_bounds__(64) void tree_build_kernel(const unsigned long long *__restrict__ counts, TreeBuildOut o) {
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    __shared__ unsigned long long cnt[256];
    __shared__ unsigned long long hkey[260];
    __shared__ uint16_t hitem[260];
    __shared__ uint16_t left[TB_NODE_STRIDE], right[TB_NODE_STRIDE], parent[TB_NODE_STRIDE], height[TB_NODE_STRIDE];
    __shared__ uint8_t sym[TB_NODE_STRIDE];
    __shared__ unsigned long long weight[TB_NODE_STRIDE];
    __shared__ uint8_t olen[256];
    __shared__ unsigned long long ocode[256];
    __shared__ uint32_t prof[9];
    __shared__ uint32_t s_nn, s_root, s_single, s_ntab8, s_maxlen, s_tie;

    unsigned long long wsum = 0;
    for (uint32_t i = lane; i < 256; i += 64) {
        cnt[i] = counts[size_t(c) * 256 + i];
        wsum += cnt[i];
        olen[i] = 0;
        ocode[i] = 0;
    }
    for (int d = 32; d >= 1; d >>= 1) wsum += __shfl_xor(wsum, d);
    if (lane < 9) prof[lane] = 0;
    if (lane == 0) { s_ntab8 = 0; s_maxlen = 0; s_single = 0; s_tie = 0; }
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
    // ---- fast path: with pairwise distinct keys at every extraction the heap's pop order IS the sorted
    // order, whatever its internal mechanics (src/min_pq.tpp), so the merge sequence follows from the
    // sorted leaves and the queue of merged nodes (whose weights are created in non-decreasing order).
    // Any equal pair among the three smallest keys of a step makes the reference's choice depend on
    // the heap layout: such a context is rebuilt by the exact heap emulation below.
    for (uint32_t i = lane; i < 256; i += 64) {
        hkey[i] = i < nleaf ? weight[i] : ~0ull;                  // hkey / hitem double as the sort buffer
        hitem[i] = uint16_t(i);
    }
    __syncthreads();
    for (uint32_t k = 2; k <= 256; k <<= 1) {                      // bitonic sort of 256 (key, id) pairs, ascending
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < 128; t += 64) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const bool up = (lo & k) == 0;
                const unsigned long long a = hkey[lo], b = hkey[hi];
                const uint16_t ia = hitem[lo], ib = hitem[hi];
                const bool gt = a > b || (a == b && ia > ib);
                if (gt == up) { hkey[lo] = b; hkey[hi] = a; hitem[lo] = ib; hitem[hi] = ia; }
            }
            __syncthreads();
        }
    }
    if (lane == 0 && nleaf > 1) {
        uint32_t i1 = 0, i2 = nleaf, nn = nleaf;                   // heads of the leaf queue and of the merged-node queue
        bool tie = false;
        auto key1 = [&](uint32_t i) -> unsigned long long { return i < nleaf ? hkey[i] : ~0ull; };
        auto key2 = [&](uint32_t i) -> unsigned long long { return i < nn ? weight[i] : ~0ull; };
        for (uint32_t step = 0; step + 1 < nleaf; ++step) {
            uint32_t pick[2];
            unsigned long long pk[2];
            for (int q = 0; q < 2; ++q) {
                const unsigned long long k1 = key1(i1), k2 = key2(i2);
                tie |= k1 == k2;                                   // both queues non-empty here or one key is ~0 (counts never reach it)
                if (k1 < k2) { pick[q] = hitem[i1]; pk[q] = k1; ++i1; } else { pick[q] = i2; pk[q] = k2; ++i2; }
            }
            const unsigned long long k3a = key1(i1), k3b = key2(i2);
            const unsigned long long k3 = k3a < k3b ? k3a : k3b;
            tie |= pk[0] == pk[1] || pk[1] == k3;
            uint32_t a = pick[0], b = pick[1];
            if (height[a] > height[b]) { const uint32_t t = a; a = b; b = t; }     // src/huffman.cpp:147-149
            left[nn] = uint16_t(a); right[nn] = uint16_t(b); parent[nn] = NONE; sym[nn] = 0;
            weight[nn] = weight[a] + weight[b];
            height[nn] = uint16_t((height[a] > height[b] ? height[a] : height[b]) + 1);
            parent[a] = parent[b] = uint16_t(nn);
            ++nn;
        }
        s_tie = tie ? 1u : 0u;
        s_nn = nn;
        s_root = nn - 1;
    }
    __syncthreads();

    if (lane == 0 && (nleaf <= 1 || s_tie)) {
        // ---- exact heap emulation (same comparisons in the same order as swap-based swim/sink)
        int nn = int(nleaf), hn = 0;
        for (int i = 0; i < nn; ++i) parent[i] = NONE;             // the fast path may have linked the leaves
        auto push = [&](unsigned long long key, uint16_t item) {
            int i = hn++;
            while (i != 0) {
                int par = (i - 1) / 2;
                if (!(hkey[par] > key)) break;                   // strict >: equal keys do not move
                hkey[i] = hkey[par]; hitem[i] = hitem[par];
                i = par;
            }
            hkey[i] = key; hitem[i] = item;
        };
        // Sift-down with the same comparisons in the same order, two levels per LDS round trip: the keys
        // of both children AND of all four grandchildren are fetched together (the walk is one lane waiting
        // on LDS latency at every level; entries below the current node are not modified while sinking).
        auto pop = [&]() -> uint16_t {
            const uint16_t top = hitem[0];
            --hn;
            const unsigned long long key = hkey[hn];
            const uint16_t item = hitem[hn];
            int i = 0;
            for (;;) {
                const int l = 2 * i + 1, r = l + 1;
                if (l >= hn) break;
                // children and grandchildren (indices clamped for the load, validity decided by < hn)
                auto at = [&](int idx) -> unsigned long long { return hkey[idx < 259 ? idx : 259]; };
                const unsigned long long kl = at(l), kr = at(r);
                const unsigned long long kll = at(2 * l + 1), klr = at(2 * l + 2), krl = at(2 * r + 1), krr = at(2 * r + 2);
                const uint16_t il = hitem[l], ir = hitem[r < 259 ? r : 259];
                const bool right1 = r < hn && kr < kl;               // right only if strictly smaller
                const int pick = right1 ? r : l;
                const unsigned long long k1 = right1 ? kr : kl;
                if (!(k1 < key)) break;
                hkey[i] = k1; hitem[i] = right1 ? ir : il;
                i = pick;
                const int l2 = 2 * i + 1, r2 = l2 + 1;
                if (l2 >= hn) break;
                const unsigned long long kl2 = right1 ? krl : kll, kr2 = right1 ? krr : klr;
                const bool right2 = r2 < hn && kr2 < kl2;
                const int pick2 = right2 ? r2 : l2;
                const unsigned long long k2 = right2 ? kr2 : kl2;
                if (!(k2 < key)) break;
                hkey[i] = k2; hitem[i] = hitem[pick2];
                i = pick2;
            }
            hkey[i] = key; hitem[i] = item;
            return top;
        };
        for (int i = 0; i < nn; ++i) push(weight[i], uint16_t(i));   // src/huffman.cpp:134-138 (ascending symbol order)
        int root = -1;
        if (hn > 0) {
            while (hn > 1) {                                      // :143-151
                uint16_t a = pop(), b = pop();
                if (height[a] > height[b]) { uint16_t t = a; a = b; b = t; }   // :147-149
                left[nn] = a; right[nn] = b; parent[nn] = NONE; sym[nn] = 0;
                weight[nn] = weight[a] + weight[b];
                height[nn] = uint16_t((height[a] > height[b] ? height[a] : height[b]) + 1);
                parent[a] = parent[b] = uint16_t(nn);
                push(weight[nn], uint16_t(nn));
                ++nn;
            }
            root = pop();
            if (left[root] == NONE) {                             // :154-162 one-symbol context
                const uint8_t s = sym[root];
                for (int k = 0; k < 2; ++k) {
                    left[nn] = right[nn] = NONE; parent[nn] = uint16_t(root); height[nn] = 0; sym[nn] = s; weight[nn] = weight[root];
                    if (k == 0) left[root] = uint16_t(nn); else right[root] = uint16_t(nn);
                    ++nn;
                }
                height[root] = 1;
                s_single = 1;
            }
        }
        s_nn = uint32_t(nn);
        s_root = root < 0 ? 0xFFFFFFFFu : uint32_t(root);
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
__global__ __launch_bounds__(256) void tree_pack_kernel(TreePackArgs a) {
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
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
    if (tid == 0 && !a.sec_base_in) a.sec_base[c] = my_base;
    if (root == 0xFFFFFFFFu) {                       // empty context: null tables
        if (tid < nprim) a.prim[(c << P) | tid] = DEC16_NULL;
        a.tree[c * TREE_STRIDE + tid] = 0;
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
            const uint32_t bit = (tid >> (P - 1 - depth)) & 1u;
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
                const uint32_t bit = (x >> (h - 1 - d2)) & 1u;
                n2 = bit ? right[n2] : left[n2];
                ++d2;
            }
            a.sec[base + off + x] = left[n2] == NONE ? uint16_t(DEC16_LEAF | ((P + d2) << 8) | sym[n2]) : uint16_t(nid[n2]);
        }
    }
    __syncthreads();
    a.tree[c * TREE_STRIDE + tid] = tr[tid];
}

hipError_t launch_tree_build(const unsigned long long *d_counts, int nctx, const TreeBuildOut &o, hipStream_t st) {
    hipLaunchKernelGGL(tree_build_kernel, dim3(nctx), dim3(64), 0, st, d_counts, o);
    return hipGetLastError();
}

hipError_t launch_tree_pack(const TreePackArgs &a, int nctx, hipStream_t st) {
    hipLaunchKernelGGL(tree_pack_kernel, dim3(nctx), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace mhk
