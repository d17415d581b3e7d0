// mh_api_model.cpp — models behind the C ABI (include/mh.h): built from counts on the host or on the device, loaded from and
// written to table files, queried; the order-2 extension's builds.
#include "mh_api_internal.hpp"

namespace mhapi {

// Host tables are always built; the device images are uploaded when a device exists.  Without one the
// model still answers table queries (mh_model_write_table, mh_model_get_code, ...) but every
// compute call on it returns MH_ERR_NO_DEVICE.
int upload_model(mh_model *m) {
    m->packed = m->host.pack();
    m->type = m->host.type;
    m->max_len = m->packed.max_len;
    m->dec_bits = m->packed.dec_bits; m->dec_h = m->packed.dec_h;
    m->dec_lds = m->packed.dec_lds; m->dec_direct = m->packed.dec_direct;
    m->nsec = uint32_t(m->packed.dec_sec.size());
    // gcd of the code lengths, the 1-bit code of one-symbol contexts aside (src/huffman.cpp:154-162: such a
    // context shifts the stream's phase once, it does not take the stream off the lattice of the others)
    for (size_t i = 0; i < size_t(256) * 256; ++i) {
        const int l = m->packed.len8[i];
        if (l && (m->min_len == 0 || l < m->min_len)) m->min_len = l;
    }
    for (int c = 0; c < 256; ++c) {
        int live = 0;
        for (int sy = 0; sy < 256; ++sy) live += m->packed.len8[size_t(c) * 256 + sy] != 0;
        if (live < 2) continue;
        for (int sy = 0; sy < 256; ++sy) m->len_gcd = gcd_u32(m->len_gcd, m->packed.len8[size_t(c) * 256 + sy]);
    }
    if (!have_device()) return MH_OK;
    if (m->max_len > mh::MAX_CODE_BITS) return MH_OK;   // compute calls report MH_ERR_CODE_TOO_LONG
    HIP_TRY(hipGetDevice(&m->device));
    const mh::Model::Packed &pk = m->packed;
    // one device allocation and one upload for all images (each piece 256-byte aligned)
    struct Piece { const void *src; size_t bytes; void **dst; };
    const size_t sec_bytes = (pk.dec_sec.size() * 2 + 15) & ~size_t(15);     // kernels copy whole uint4s
    const Piece pieces[] = {
        {pk.enc16.data(), 65536 * 2, reinterpret_cast<void **>(&m->d_enc16)},
        {pk.len8.data(), 65536, reinterpret_cast<void **>(&m->d_len8)},
        {pk.len_slot.data(), 65536, reinterpret_cast<void **>(&m->d_len_slot)},
        {pk.code64.data(), 65536 * 8, reinterpret_cast<void **>(&m->d_code64)},
        {pk.tree.data(), size_t(256) * mh::TREE_STRIDE * 4, reinterpret_cast<void **>(&m->d_tree)},
        {pk.dec_prim.data(), pk.dec_prim.size() * 2, reinterpret_cast<void **>(&m->d_prim)},
        {pk.dec_sec.data(), pk.dec_sec.size() * 2, reinterpret_cast<void **>(&m->d_sec)},
        {pk.sec_base.data(), 256 * 4, reinterpret_cast<void **>(&m->d_sec_base)},
    };
    size_t total = 0, off[8];
    for (int i = 0; i < 8; ++i) { off[i] = total; total += ((i == 6 ? sec_bytes : pieces[i].bytes) + 255) & ~size_t(255); }
    total += 256;
    std::vector<unsigned char> staging(total, 0);
    for (int i = 0; i < 8; ++i)
        if (pieces[i].bytes) std::memcpy(staging.data() + off[i], pieces[i].src, pieces[i].bytes);
    HIP_TRY(hipMalloc(&m->d_block, total));
    HIP_TRY(hipMemcpy(m->d_block, staging.data(), total, hipMemcpyHostToDevice));
    for (int i = 0; i < 8; ++i) *pieces[i].dst = static_cast<unsigned char *>(m->d_block) + off[i];
    if (tile_p_choice()) {
        const mh::Model::TilePacked tp = m->host.pack_tile(tile_p_choice());
        if (tp.P) {
            const size_t pb = (tp.prim.size() * 2 + 255) & ~size_t(255), sb = tp.sec.size() * 2 + 64;
            HIP_TRY(hipMalloc(&m->d_tile_own, pb + sb));
            HIP_TRY(hipMemset(m->d_tile_own, 0, pb + sb));
            m->d_tprim = static_cast<uint16_t *>(m->d_tile_own);
            m->d_tsec = reinterpret_cast<uint16_t *>(static_cast<unsigned char *>(m->d_tile_own) + pb);
            HIP_TRY(hipMemcpy(m->d_tprim, tp.prim.data(), tp.prim.size() * 2, hipMemcpyHostToDevice));
            if (!tp.sec.empty()) HIP_TRY(hipMemcpy(m->d_tsec, tp.sec.data(), tp.sec.size() * 2, hipMemcpyHostToDevice));
            m->tile_p = tp.P; m->tile_h = tp.H; m->tile_nsec = uint32_t(tp.sec.size());
        }
    }
    return MH_OK;
}

// After a device build the trees live in HBM only; table files and code/LUT queries need them on the
// host.  Built once, on demand.
int ensure_mirror(const mh_model *cm) {
    mh_model *m = const_cast<mh_model *>(cm);
    std::lock_guard<std::mutex> lock(m->mu);
    if (m->mirror_ready) return MH_OK;
    const size_t nn = size_t(256) * mhk::TB_NODE_STRIDE;
    std::vector<uint16_t> left(nn), right(nn);
    std::vector<uint8_t> sym(nn);
    std::vector<uint32_t> meta(size_t(256) * mhk::TB_META_STRIDE);
    HIP_TRY(hipMemcpy(left.data(), m->d_node_left, nn * 2, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(right.data(), m->d_node_right, nn * 2, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sym.data(), m->d_node_sym, nn, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(meta.data(), m->d_meta, meta.size() * 4, hipMemcpyDeviceToHost));
    m->host.type = 1;
    m->host.ctx.assign(256, mh::ContextCoder{});
    for (int c = 0; c < 256; ++c) {
        const uint32_t *mt = &meta[size_t(c) * mhk::TB_META_STRIDE];
        const int root = mt[1] == 0xFFFFFFFFu ? -1 : int(mt[1]);
        m->host.ctx[c].adopt(int(mt[0]), root, &left[size_t(c) * mhk::TB_NODE_STRIDE], &right[size_t(c) * mhk::TB_NODE_STRIDE],
                             &sym[size_t(c) * mhk::TB_NODE_STRIDE]);
        m->host.ctx_weight[c] = (uint64_t(mt[14]) << 32) | mt[13];
    }
    m->mirror_ready = true;
    return MH_OK;
}

int finish_model(mh_model *m, mh_model **out) {
    int rc = upload_model(m);
    if (rc != MH_OK) { mh_model_free(m); return rc; }
    *out = m;
    return MH_OK;
}

}  // namespace mhapi

using namespace mhapi;

extern "C" {

/* ---------------------------------------------------------------- model */

static int model2_from_host_counts(const uint64_t *counts, mh_model **out) {
    if (!have_device()) return MH_ERR_NO_DEVICE;                 // the order-2 build has no host twin: it runs on the device
    DevBuf d_counts;
    HIP_TRY(d_counts.alloc((size_t(1) << 24) * 8));
    HIP_TRY(hipMemcpy(d_counts.p, counts, (size_t(1) << 24) * 8, hipMemcpyHostToDevice));
    return mh_dev_model_from_counts(d_counts.as<uint64_t>(), 2, nullptr, out);
}

int mh_model_from_counts(const uint64_t *counts, int order, mh_model **out) {
    if (counts && out && order == 2) return model2_from_host_counts(counts, out);
    if (!counts || !out || (order != 0 && order != 1)) return MH_ERR_ARG;
    mh_model *m = new (std::nothrow) mh_model;
    if (!m) return MH_ERR_NOMEM;
    m->host.build_from_counts(counts, order);
    return finish_model(m, out);
}

static int model_from_device_counts_via_host(const uint64_t *d_counts, int order, hipStream_t st, mh_model **out) {
    size_t ncount = order ? 65536 : 256;
    std::vector<uint64_t> counts(ncount);
    HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, ncount * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return mh_model_from_counts(counts.data(), order, out);
}

// Fixed-size part of a device-built model: every image + the node arrays, each piece 256-byte aligned.
namespace {
// internal: dev_model_build met a model whose second-level tables need the general (non-uniform) L2 layout, which only
// the host packer lays out (more than 32767 depth-8 inner nodes); never returned through the C ABI
constexpr int BUILD_NEEDS_HOST = -1000;
struct BuildLayout { size_t off[12], fixed; };
BuildLayout build_layout() {
    const size_t nn = size_t(256) * mhk::TB_NODE_STRIDE;
    const size_t sizes[12] = {65536 * 2, 65536, 65536, 65536 * 8, size_t(256) * mh::TREE_STRIDE * 4, 65536 * 2, 256 * 4,
                              nn * 2, nn * 2, nn, nn, size_t(256) * mhk::TB_META_STRIDE * 4};
    BuildLayout L;
    size_t total = 0;
    for (int i = 0; i < 12; ++i) { L.off[i] = total; total += (sizes[i] + 255) & ~size_t(255); }
    L.fixed = total;
    return L;
}
// second-level tables: at most 32767 uniform tables of 256 entries in the L2 layout (far less in the LDS layout)
constexpr size_t MODEL_WS_SEC_BYTES = size_t(32768) * 256 * 2 + 64;
// the tile decoder's tables: a first level of at most 256 << 8 entries, at most 32768 second-level tables of 256
constexpr size_t MODEL_WS_TILE_BYTES = size_t(65536) * 2 + 256 + size_t(32768) * 256 * 2 + 64 + 512;

// d_ws == nullptr: the model allocates (and owns) its device memory.  Otherwise it lives in the caller's
// workspace: no allocation, and the stream is synchronised exactly once (16 KiB of table sizes come back
// so that the host can pick the decode-table layout).
int dev_model_build(const uint64_t *d_counts, void *d_ws, size_t ws_bytes, hipStream_t st, mh_model **out) {
    mh_model *m = new (std::nothrow) mh_model;
    if (!m) return MH_ERR_NOMEM;
    m->type = 1;
    m->mirror_ready = false;
    auto fail = [&](int rc) { mh_model_free(m); return rc; };
#define HIP_TRY_M(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(hip_fail(_e)); } while (0)
    HIP_TRY_M(hipGetDevice(&m->device));
    const BuildLayout L = build_layout();
    unsigned char *b;
    if (d_ws) {
        if (!aligned16(d_ws) || ws_bytes < L.fixed + 64) return fail(MH_ERR_CAPACITY);
        b = static_cast<unsigned char *>(d_ws);
    } else {
        HIP_TRY_M(hipMalloc(&m->d_build, L.fixed));
        b = static_cast<unsigned char *>(m->d_build);
    }
    const size_t *off = L.off;
    m->d_enc16 = reinterpret_cast<uint16_t *>(b + off[0]);
    m->d_len8 = b + off[1];
    m->d_len_slot = b + off[2];
    m->d_code64 = reinterpret_cast<uint64_t *>(b + off[3]);
    m->d_tree = reinterpret_cast<uint32_t *>(b + off[4]);
    m->d_prim = reinterpret_cast<uint16_t *>(b + off[5]);
    m->d_sec_base = reinterpret_cast<uint32_t *>(b + off[6]);
    m->d_node_left = reinterpret_cast<uint16_t *>(b + off[7]);
    m->d_node_right = reinterpret_cast<uint16_t *>(b + off[8]);
    m->d_node_sym = b + off[9];
    uint8_t *d_node_height = b + off[10];
    m->d_meta = reinterpret_cast<uint32_t *>(b + off[11]);

    mhk::TreeBuildOut tb{m->d_len8, reinterpret_cast<unsigned long long *>(m->d_code64), m->d_enc16, m->d_len_slot,
                         m->d_node_left, m->d_node_right, m->d_node_sym, d_node_height, m->d_meta, 8u};
    HIP_TRY_M(mhk::launch_tree_build(reinterpret_cast<const unsigned long long *>(d_counts), 256, tb, st));
    // (a pinned landing place, one per thread, kept for the life of the process — 16 KiB; freeing it from a destructor at exit
    // would call into a runtime that may already be gone: a copy into pageable memory is staged by the runtime)
    struct PinnedMeta { uint32_t *p = nullptr; };
    static thread_local PinnedMeta pinned;
    const size_t meta_words = size_t(256) * mhk::TB_META_STRIDE;
    std::vector<uint32_t> meta_pageable;
    if (!pinned.p && hipHostMalloc(reinterpret_cast<void **>(&pinned.p), meta_words * 4, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        pinned.p = nullptr;
    }
    if (!pinned.p) meta_pageable.resize(meta_words);
    uint32_t *meta = pinned.p ? pinned.p : meta_pageable.data();
    HIP_TRY_M(hipMemcpyAsync(meta, m->d_meta, meta_words * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY_M(hipStreamSynchronize(st));                      // 16 KiB of sizes: the one sync of this call

    // same layout rule as mh::Model::pack()
    size_t tot[9] = {0}, worst[9] = {0}, ntab8 = 0;
    uint64_t weight[256];
    for (int c = 0; c < 256; ++c) {
        const uint32_t *mt = &meta[size_t(c) * mhk::TB_META_STRIDE];
        m->max_len = std::max(m->max_len, int(mt[2]));
        note_min_len(m, mt);
        // mt[15]: bit l-1 = a code of l bits exists (bit 31: 32 or more); one-symbol contexts aside, as in upload_model()
        for (uint32_t l = 1; l <= 32; ++l)
            if (mt[15] & (1u << (l - 1))) m->len_gcd = gcd_u32(m->len_gcd, l == 32 ? 1u : l);
        ntab8 += mt[3];
        for (int P = 0; P < 9; ++P) { tot[P] += mt[4 + P]; worst[P] = std::max(worst[P], size_t(mt[4 + P])); }
        weight[c] = (uint64_t(mt[14]) << 32) | mt[13];
    }
    if (m->max_len > mh::MAX_CODE_BITS) { *out = m; return MH_OK; }   // compute calls report MH_ERR_CODE_TOO_LONG
    int P = 0;
    for (int q = 8; q >= 4 && !P; --q)
        if (worst[q] <= size_t(mh::DEC_SEC_MAX_PER_CTX) && (size_t(256) << q) + tot[q] <= size_t(mh::DEC_LDS_ENTRIES)) P = q;
    m->dec_lds = P != 0;
    if (!m->dec_lds) {
        if (ntab8 > 32767) return fail(BUILD_NEEDS_HOST);         // general L2 layout: rare; the caller lets the host do it
        P = 8;
        m->dec_direct = true;
        m->dec_h = std::min(std::max(m->max_len - 8, 1), 8);
    }
    m->dec_bits = P;
    int order_idx[256];
    for (int i = 0; i < 256; ++i) order_idx[i] = i;
    std::stable_sort(order_idx, order_idx + 256, [&](int a, int b2) { return weight[a] > weight[b2]; });
    mhk::TreePackArgs pa{};
    size_t nsec = 0;
    for (int i = 0; i < 256; ++i) {
        const int c = order_idx[i];
        const uint32_t *mt = &meta[size_t(c) * mhk::TB_META_STRIDE];
        pa.sec_base_val[c] = uint32_t(nsec);                     // travels in the kernel arguments: no pageable copy to wait for
        nsec += m->dec_direct ? (size_t(mt[3]) << m->dec_h) : size_t(mt[4 + P]);
    }
    m->nsec = uint32_t(nsec);
    const size_t sec_bytes = ((nsec * 2 + 15) & ~size_t(15)) + 16;
    if (d_ws) {
        if (ws_bytes < L.fixed + sec_bytes) return fail(MH_ERR_CAPACITY);
        m->d_sec = reinterpret_cast<uint16_t *>(b + L.fixed);
    } else {
        HIP_TRY_M(hipMalloc(&m->d_sec_own, sec_bytes));
        m->d_sec = static_cast<uint16_t *>(m->d_sec_own);
    }
    HIP_TRY_M(hipMemsetAsync(m->d_sec, 0, sec_bytes, st));
    pa.node_left = m->d_node_left; pa.node_right = m->d_node_right; pa.node_sym = m->d_node_sym; pa.node_height = d_node_height;
    pa.ctx_meta = m->d_meta; pa.sec_base = m->d_sec_base;
    pa.P = uint32_t(P); pa.direct = m->dec_direct ? 1u : 0u; pa.H = uint32_t(m->dec_h); pa.hcap = 8u;
    pa.prim = m->d_prim; pa.sec = m->d_sec; pa.tree = m->d_tree;
    bool packed = false;                                      // (the tile tables' packing below takes this one along: one launch)
    // ---- the tile decoder's tables: the same trees packed once more, LSB-first, with a first level of tile_p bits
    if (const int tP = tile_p_choice()) {
        const int tH = std::min(std::max(m->max_len - tP, 1), 8);
        size_t ntab = size_t(256) << tP;
        mhk::TreePackArgs pt{};
        if (tP == 8) {
            ntab = 0;
            for (int c = 0; c < 256; ++c) { pt.sec_base_val[c] = uint32_t(ntab << tH); ntab += meta[size_t(c) * mhk::TB_META_STRIDE + 3]; }
        } else {
            for (int c = 0; c < 256; ++c) pt.sec_base_val[c] = uint32_t(size_t(c) << (tP + tH));
        }
        if (ntab <= 32767 || tP < 8) {
            const size_t pb = ((size_t(256) << tP) * 2 + 255) & ~size_t(255), sb = (ntab << tH) * 2 + 64;
            unsigned char *tb;
            if (d_ws) {
                const size_t at = (L.fixed + sec_bytes + 255) & ~size_t(255);
                if (ws_bytes < at + pb + sb) return fail(MH_ERR_CAPACITY);
                tb = b + at;
            } else {
                HIP_TRY_M(hipMalloc(&m->d_tile_own, pb + sb));
                tb = static_cast<unsigned char *>(m->d_tile_own);
            }
            m->d_tprim = reinterpret_cast<uint16_t *>(tb);
            m->d_tsec = reinterpret_cast<uint16_t *>(tb + pb);
            HIP_TRY_M(hipMemsetAsync(m->d_tsec, 0, sb, st));
            pt.node_left = m->d_node_left; pt.node_right = m->d_node_right; pt.node_sym = m->d_node_sym; pt.node_height = d_node_height;
            pt.ctx_meta = m->d_meta; pt.sec_base = nullptr; pt.sec_base_in = nullptr;
            pt.P = uint32_t(tP); pt.direct = 1u; pt.H = uint32_t(tH); pt.hcap = 8u;
            pt.prim = m->d_tprim; pt.sec = m->d_tsec; pt.tree = nullptr; pt.lsb = 1u;
            HIP_TRY_M(mhk::launch_tree_pack2(pa, pt, 256, st));
            packed = true;
            m->tile_p = tP; m->tile_h = tH; m->tile_nsec = uint32_t(ntab << tH);
        }
    }
    if (!packed) HIP_TRY_M(mhk::launch_tree_pack(pa, 256, st));
#undef HIP_TRY_M
    *out = m;
    return MH_OK;
}
}  // namespace

// ---- order 2 (extension; parity unpinned: the spec is the generalised oracle, oracle/mh_oracle.h) ----------
namespace {
constexpr uint32_t O2_CTX = 65536;
constexpr uint32_t O2_HCAP = 4;          // second-level tables of at most 16 entries: <= 4096 entries per context
const unsigned char O2_MAGIC[4] = {'M', 'H', '2', 1};

bool is_o2_table(const uint8_t *b, size_t n) {
    if (n < 37 || b[0] != 0x80) return false;
    for (int i = 1; i < 33; ++i) if (b[i]) return false;
    return std::memcmp(b + 33, O2_MAGIC, 4) == 0;
}

struct Build2Layout { size_t off[12], total; };
Build2Layout build2_layout() {
    const size_t nn = size_t(O2_CTX) * mhk::TB_NODE_STRIDE, ne = size_t(O2_CTX) * 256;
    const size_t sizes[12] = {ne, ne * 8, ne * 4, ne * 2, size_t(O2_CTX) * 4, nn * 2, nn * 2, nn, nn, size_t(O2_CTX) * mhk::TB_META_STRIDE * 4, 256, ne * 8};
    Build2Layout L;
    size_t total = 0;
    for (int i = 0; i < 12; ++i) { L.off[i] = total; total += (sizes[i] + 255) & ~size_t(255); }
    L.total = total;
    return L;
}

void place2(mh_model *m, unsigned char *b, const Build2Layout &L, uint8_t **node_height) {
    m->d_len8 = b + L.off[0];
    m->d_code64 = reinterpret_cast<uint64_t *>(b + L.off[1]);
    m->d_tree = reinterpret_cast<uint32_t *>(b + L.off[2]);
    m->d_prim = reinterpret_cast<uint16_t *>(b + L.off[3]);
    m->d_sec_base = reinterpret_cast<uint32_t *>(b + L.off[4]);
    m->d_node_left = reinterpret_cast<uint16_t *>(b + L.off[5]);
    m->d_node_right = reinterpret_cast<uint16_t *>(b + L.off[6]);
    m->d_node_sym = b + L.off[7];
    *node_height = b + L.off[8];
    m->d_meta = reinterpret_cast<uint32_t *>(b + L.off[9]);
    m->d_enc64 = reinterpret_cast<uint64_t *>(b + L.off[11]);
}

// The live contexts' tables: slots for the heaviest live contexts whose two bytes are both among the 63 most frequent
// byte values (ids 0..62; everything else is id 63 = escape).  Encoder image and tile-decoder tables are filled on the
// device (o2_hot_pack_kernel); the host only ranks (it holds every context's weight after the build's one sync).
//   o2_enc_ok: the slots carry all but 1e-5 of the input (an escape costs a whole wave sub-step the slow path)
//   o2_dec_ok: EVERY live context has a slot (the decoder follows slot -> slot and has no other path)
constexpr uint32_t O2_SLOTS_MAX = 440;   // (440 + 1) rows of 128 B + 8448 B of maps = 64896 B <= the length pass's 64 KiB of LDS
constexpr uint32_t O2_TILE_P = 6;
int o2_hot_setup(mh_model *m, const std::vector<uint64_t> &weight, const std::vector<uint8_t> &live, const uint8_t *, hipStream_t st) {
    if (getenv("MH_O2_NO_HOT")) return MH_OK;
    uint64_t bw[256] = {0};
    long double total = 0;
    uint32_t nlive = 0;
    for (uint32_t c = 0; c < O2_CTX; ++c) {
        if (!live[c]) continue;
        ++nlive;
        const uint64_t w = weight[c] ? weight[c] : 1;              // (a model from a table file has no weights)
        bw[c >> 8] += w; bw[c & 255u] += w;
        total += w;
    }
    if (nlive == 0) return MH_OK;
    int order[256];
    for (int i = 0; i < 256; ++i) order[i] = i;
    std::stable_sort(order, order + 256, [&](int a, int b) { return bw[a] > bw[b]; });
    uint8_t symid[256];
    std::memset(symid, 63, sizeof symid);
    mhk::O2HotArgs a{};
    for (int i = 0; i < 63; ++i)
        if (bw[order[i]]) { symid[order[i]] = uint8_t(i); a.id_sym[i] = uint8_t(order[i]); a.id_used[i] = 1; }
    std::vector<uint32_t> cand;
    for (uint32_t c = 0; c < O2_CTX; ++c)
        if (live[c] && symid[c >> 8] < 63 && symid[c & 255u] < 63) cand.push_back(c);
    std::stable_sort(cand.begin(), cand.end(), [&](uint32_t x, uint32_t y) { return weight[x] > weight[y]; });
    const uint32_t nslots = uint32_t(std::min<size_t>(cand.size(), O2_SLOTS_MAX));
    if (nslots == 0) return MH_OK;
    long double covered = 0;
    std::vector<uint16_t> slot_ctx(nslots), ctx2slot(O2_CTX, 0xFFFF);
    std::vector<uint8_t> slot_id1(nslots);
    std::vector<unsigned char> head(8448, 0);                      // symid | ctxmap
    std::memcpy(head.data(), symid, 256);
    uint16_t *ctxmap = reinterpret_cast<uint16_t *>(head.data() + 256);
    for (int i = 0; i < 64 * 64; ++i) ctxmap[i] = uint16_t(nslots);    // the all-escape row
    for (uint32_t s = 0; s < nslots; ++s) {
        const uint32_t c = cand[s];
        slot_ctx[s] = uint16_t(c);
        slot_id1[s] = symid[c & 255u];
        ctx2slot[c] = uint16_t(s);
        ctxmap[(uint32_t(symid[c >> 8]) << 6) | (symid[c & 255u] ^ symid[c >> 8])] = uint16_t(s);   // column XOR-ed with the row's id (bank spreading)
        covered += weight[c] ? weight[c] : 1;
    }
    const bool all_hot = nslots == nlive;
    const uint32_t P = O2_TILE_P, H = uint32_t(std::min(std::max(m->max_len - int(P), 1), 8));
    const bool tiles = all_hot && m->d_node_left != nullptr;
    const size_t img = 8448 + size_t(nslots + 1) * 128;
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t off_map = up(img), off_sc = off_map + up(size_t(O2_CTX) * 2), off_s1 = off_sc + up(size_t(nslots) * 2), off_tp = off_s1 + up(nslots);
    const size_t off_ts = off_tp + (tiles ? up((size_t(nslots) << P) * 4) : 0);
    const size_t nsec = tiles ? ((size_t(nslots) << P) << H) : 0;
    const size_t tot = off_ts + up(nsec * 4 + 64);
    HIP_TRY(hipMalloc(&m->d_o2hot, tot));
    unsigned char *b = static_cast<unsigned char *>(m->d_o2hot);
    HIP_TRY(hipMemsetAsync(b + off_ts, 0, up(nsec * 4 + 64), st));
    HIP_TRY(hipMemcpyAsync(b, head.data(), head.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + off_map, ctx2slot.data(), size_t(O2_CTX) * 2, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + off_sc, slot_ctx.data(), size_t(nslots) * 2, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + off_s1, slot_id1.data(), size_t(nslots), hipMemcpyHostToDevice, st));
    a.slot_ctx = reinterpret_cast<const uint16_t *>(b + off_sc); a.nslots = nslots;
    a.slot_id1 = b + off_s1;
    a.len8 = m->d_len8; a.code64 = reinterpret_cast<const unsigned long long *>(m->d_code64);
    a.hot = reinterpret_cast<uint16_t *>(b + 8448);
    a.ctx2slot = reinterpret_cast<const uint16_t *>(b + off_map);
    a.P = P; a.H = H;
    if (tiles) {
        a.node_left = m->d_node_left; a.node_right = m->d_node_right; a.node_sym = m->d_node_sym; a.ctx_meta = m->d_meta;
        a.tprim = reinterpret_cast<uint32_t *>(b + off_tp); a.tsec = reinterpret_cast<uint32_t *>(b + off_ts);
    }
    HIP_TRY(mhk::launch_o2_hot_pack(a, st));
    HIP_TRY(hipStreamSynchronize(st));                            // the staging vectors above are on this frame (the caller syncs next anyway: the wait is paid once)
    m->d_o2img = b; m->o2img_bytes = uint32_t(img);
    m->d_ctx2slot = reinterpret_cast<uint16_t *>(b + off_map);
    m->o2_nslots = nslots; m->o2_p = P; m->o2_h = H; m->o2_nsec = uint32_t(nsec);
    m->d_tprim2 = tiles ? a.tprim : nullptr; m->d_tsec2 = tiles ? a.tsec : nullptr;
    m->o2_enc_ok = covered >= total * (1.0L - 1e-5L);
    m->o2_dec_ok = tiles;
    return MH_OK;
}

// Order-2 model build in two steps, so that G ranks can share it (reduce-scatter of the 1 << 24 counts,
// every rank builds the trees of its 65536 / G contexts, all-gather of the per-context arrays):
//   build2_slice   trees, code lengths, codewords and node arrays of contexts [c0, c1) from their counts, written to
//                  their place in the (caller's or the model's own) workspace — every array is laid out by context, so
//                  a rank's share of each is ONE contiguous range that a collective can gather in place
//   build2_finish  with all 65536 contexts in place: the packed encoder entries, the decode tables, the live contexts'
//                  LDS tables; one sync for the 4 MiB of per-context sizes
int build2_slice(const uint64_t *d_counts_slice, uint32_t c0, uint32_t c1, unsigned char *b, hipStream_t st) {
    if (c0 >= c1 || c1 > O2_CTX) return MH_ERR_ARG;
    const Build2Layout L = build2_layout();
    mhk::TreeBuildOut tb{b + L.off[0] + size_t(c0) * 256, reinterpret_cast<unsigned long long *>(b + L.off[1]) + size_t(c0) * 256, nullptr, nullptr,
                         reinterpret_cast<uint16_t *>(b + L.off[5]) + size_t(c0) * mhk::TB_NODE_STRIDE,
                         reinterpret_cast<uint16_t *>(b + L.off[6]) + size_t(c0) * mhk::TB_NODE_STRIDE,
                         b + L.off[7] + size_t(c0) * mhk::TB_NODE_STRIDE, b + L.off[8] + size_t(c0) * mhk::TB_NODE_STRIDE,
                         reinterpret_cast<uint32_t *>(b + L.off[9]) + size_t(c0) * mhk::TB_META_STRIDE, O2_HCAP};
    HIP_TRY(mhk::launch_tree_build(reinterpret_cast<const unsigned long long *>(d_counts_slice), int(c1 - c0), tb, st));
    return MH_OK;
}

int build2_finish(unsigned char *b, bool owned, hipStream_t st, mh_model **out) {
    mh_model *m = new (std::nothrow) mh_model;
    if (!m) return MH_ERR_NOMEM;
    m->type = 2; m->nctx = O2_CTX; m->mirror_ready = false;
    m->dec_bits = 8; m->dec_lds = false; m->dec_direct = false; m->dec_h = 0;
    if (owned) { m->d_build = b; m->build_cached = true; }
    auto fail = [&](int rc) { mh_model_free(m); return rc; };
#define HIP_TRY_M(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(hip_fail(_e)); } while (0)
    HIP_TRY_M(hipGetDevice(&m->device));
    const Build2Layout L = build2_layout();
    uint8_t *d_node_height = nullptr;
    place2(m, b, L, &d_node_height);
    HIP_TRY_M(mhk::launch_enc64_pack(m->d_len8, m->d_code64, m->d_enc64, uint64_t(O2_CTX) * 256, st));
    std::vector<uint32_t> meta(size_t(O2_CTX) * mhk::TB_META_STRIDE);
    HIP_TRY_M(hipMemcpyAsync(meta.data(), m->d_meta, meta.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY_M(hipStreamSynchronize(st));
    std::vector<uint32_t> sec_base(O2_CTX);
    size_t nsec = 0;
    uint32_t lenmask = 0;                                         // union of the contexts' code-length sets (gcd of a union = gcd of its members)
    for (uint32_t c = 0; c < O2_CTX; ++c) {
        const uint32_t *mt = &meta[size_t(c) * mhk::TB_META_STRIDE];
        m->max_len = std::max(m->max_len, int(mt[2]));
        note_min_len(m, mt);
        lenmask |= mt[15];
        sec_base[c] = uint32_t(nsec);
        nsec += mt[4 + 8];                                        // tables under the depth-8 nodes, heights capped at O2_HCAP
    }
    for (uint32_t l = 1; l <= 32; ++l)
        if (lenmask & (1u << (l - 1))) m->len_gcd = gcd_u32(m->len_gcd, l == 32 ? 1u : l);
    if (nsec > 0xFFFFFFFFull - 4096) return fail(MH_ERR_CAPACITY);
    m->nsec = uint32_t(nsec);
    if (m->max_len > mh::MAX_CODE_BITS) { *out = m; return MH_OK; }
    const size_t sec_bytes = ((nsec * 2 + 15) & ~size_t(15)) + 16;
    HIP_TRY_M(hipMalloc(&m->d_sec_own, sec_bytes));
    m->d_sec = static_cast<uint16_t *>(m->d_sec_own);
    HIP_TRY_M(hipMemsetAsync(m->d_sec_own, 0, sec_bytes, st));
    HIP_TRY_M(hipMemcpyAsync(m->d_sec_base, sec_base.data(), size_t(O2_CTX) * 4, hipMemcpyHostToDevice, st));
    mhk::TreePackArgs pa{};
    pa.node_left = m->d_node_left; pa.node_right = m->d_node_right; pa.node_sym = m->d_node_sym; pa.node_height = d_node_height;
    pa.ctx_meta = m->d_meta; pa.sec_base = m->d_sec_base; pa.sec_base_in = m->d_sec_base;
    pa.P = 8; pa.direct = 0; pa.H = 0; pa.hcap = O2_HCAP;
    pa.prim = m->d_prim; pa.sec = m->d_sec; pa.tree = m->d_tree;
    HIP_TRY_M(mhk::launch_tree_pack(pa, int(O2_CTX), st));
    // ---- the live contexts' own tables (text-like sources: a few hundred contexts over a few dozen byte values)
    {
        std::vector<uint64_t> weight(O2_CTX);
        std::vector<uint8_t> live(O2_CTX);
        for (uint32_t c = 0; c < O2_CTX; ++c) {
            const uint32_t *mt = &meta[size_t(c) * mhk::TB_META_STRIDE];
            weight[c] = (uint64_t(mt[14]) << 32) | mt[13];
            live[c] = mt[1] != 0xFFFFFFFFu;
        }
        const int rc2 = o2_hot_setup(m, weight, live, d_node_height, st);
        if (rc2 != MH_OK) return fail(rc2);
    }
    HIP_TRY_M(hipStreamSynchronize(st));                         // sec_base lives in pageable host memory
#undef HIP_TRY_M
    *out = m;
    return MH_OK;
}

// counts (1 << 24, device) -> 65536 trees, codes and decode tables, all on the device
// The ~600 MiB build block of an order-2 model is kept when a model is freed and handed to the next build on the same
// device (a codec that rebuilds its model per stream — bench.py — otherwise pays a hipMalloc / hipFree of that size per step).
struct Build2Cache { std::mutex mu; void *p = nullptr; int device = -1; } g_build2_cache;
void *build2_block_take() {
    std::lock_guard<std::mutex> lock(g_build2_cache.mu);
    int dev = -1;
    if (g_build2_cache.p && hipGetDevice(&dev) == hipSuccess && dev == g_build2_cache.device) {
        void *p = g_build2_cache.p;
        g_build2_cache.p = nullptr;
        (void)hipDeviceSynchronize();             // what hipFree would have waited for: nothing still reads the freed model's tables
        return p;
    }
    return nullptr;
}
void build2_block_give(void *p, int device) {
    std::lock_guard<std::mutex> lock(g_build2_cache.mu);
    if (g_build2_cache.p) (void)hipFree(g_build2_cache.p);
    g_build2_cache.p = p; g_build2_cache.device = device;
}

int dev_model_build2(const uint64_t *d_counts, hipStream_t st, mh_model **out) {
    void *b = build2_block_take();
    if (!b) HIP_TRY(hipMalloc(&b, build2_layout().total));
    const int rc = build2_slice(d_counts, 0, O2_CTX, static_cast<unsigned char *>(b), st);
    if (rc != MH_OK) { (void)hipFree(b); return rc; }
    return build2_finish(static_cast<unsigned char *>(b), true, st, out);     // (the model frees `b`, also when it fails)
}

// order-2 table file -> host-derived images (ContextCoder per non-empty context) -> device
int model2_from_table(const uint8_t *bytes, size_t n, mh_model **out) {
    mh_model *m = new (std::nothrow) mh_model;
    if (!m) return MH_ERR_NOMEM;
    m->type = 2; m->nctx = O2_CTX; m->mirror_ready = true;
    m->dec_bits = 8; m->dec_lds = false; m->dec_direct = false; m->dec_h = 0;
    m->table2.assign(bytes, bytes + n);
    const size_t ne = size_t(O2_CTX) * 256;
    std::vector<uint8_t> len8(ne, 0);
    std::vector<uint64_t> code64(ne, 0);
    std::vector<uint16_t> prim(ne, mh::DEC16_NULL), sec;
    std::vector<uint32_t> tree(ne, 0), sec_base(O2_CTX, 0);
    mh::BitReader in(bytes + 37, n - 37);
    mh::ContextCoder cc;
    std::vector<uint8_t> live2(O2_CTX, 0);
    for (uint32_t c = 0; c < O2_CTX; ++c) {
        sec_base[c] = uint32_t(sec.size());
        if (in.bit()) {
            live2[c] = 1;
            if (!cc.load(in)) { delete m; return MH_ERR_BADTABLE; }
            int live = 0;
            for (int sy = 0; sy < 256; ++sy) {
                const mh::Code &cd = cc.code(sy);
                len8[size_t(c) * 256 + sy] = uint8_t(std::min(cd.len, 255));
                code64[size_t(c) * 256 + sy] = cd.len <= 64 ? cd.right_aligned() : 0;
                live += cd.len != 0;
                if (cd.len && (m->min_len == 0 || cd.len < m->min_len)) m->min_len = cd.len;
            }
            m->max_len = std::max(m->max_len, cc.max_len());
            if (live >= 2) for (int sy = 0; sy < 256; ++sy) m->len_gcd = gcd_u32(m->len_gcd, len8[size_t(c) * 256 + sy]);
            cc.pack_decode(8, int(O2_HCAP), 0, &prim[size_t(c) << 8], sec, sec_base[c], &tree[size_t(c) * 256]);
        }
        if (in.failed()) { delete m; return MH_ERR_BADTABLE; }
    }
    m->nsec = uint32_t(sec.size());
    if (!have_device() || m->max_len > mh::MAX_CODE_BITS) { *out = m; return MH_OK; }
    auto fail = [&](int rc) { mh_model_free(m); return rc; };
#define HIP_TRY_M(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(hip_fail(_e)); } while (0)
    HIP_TRY_M(hipGetDevice(&m->device));
    const Build2Layout L = build2_layout();
    HIP_TRY_M(hipMalloc(&m->d_build, L.total));
    uint8_t *d_node_height = nullptr;
    place2(m, static_cast<unsigned char *>(m->d_build), L, &d_node_height);
    m->d_node_left = m->d_node_right = nullptr; m->d_node_sym = nullptr; m->d_meta = nullptr;   // no trees on the device
    const size_t sec_bytes = ((sec.size() * 2 + 15) & ~size_t(15)) + 16;
    HIP_TRY_M(hipMalloc(&m->d_sec_own, sec_bytes));
    m->d_sec = static_cast<uint16_t *>(m->d_sec_own);
    HIP_TRY_M(hipMemset(m->d_sec_own, 0, sec_bytes));
    HIP_TRY_M(hipMemcpy(m->d_len8, len8.data(), ne, hipMemcpyHostToDevice));
    HIP_TRY_M(hipMemcpy(m->d_code64, code64.data(), ne * 8, hipMemcpyHostToDevice));
    HIP_TRY_M(mhk::launch_enc64_pack(m->d_len8, m->d_code64, m->d_enc64, uint64_t(ne), nullptr));
    HIP_TRY_M(hipStreamSynchronize(nullptr));
    HIP_TRY_M(hipMemcpy(m->d_tree, tree.data(), ne * 4, hipMemcpyHostToDevice));
    HIP_TRY_M(hipMemcpy(m->d_prim, prim.data(), ne * 2, hipMemcpyHostToDevice));
    HIP_TRY_M(hipMemcpy(m->d_sec_base, sec_base.data(), size_t(O2_CTX) * 4, hipMemcpyHostToDevice));
    if (!sec.empty()) HIP_TRY_M(hipMemcpy(m->d_sec, sec.data(), sec.size() * 2, hipMemcpyHostToDevice));
    {   // the encoder's LDS image of the live contexts (no weights in a table file: every live context counts the same)
        const int rc2 = o2_hot_setup(m, std::vector<uint64_t>(O2_CTX, 0), live2, nullptr, nullptr);
        if (rc2 != MH_OK) return fail(rc2);
    }
#undef HIP_TRY_M
    *out = m;
    return MH_OK;
}

// table file of a device-built order-2 model, written from the node arrays
int model2_write_table(const mh_model *m, std::vector<uint8_t> &out) {
    if (!m->table2.empty()) { out = m->table2; return MH_OK; }
    if (!m->d_node_left) return MH_ERR_NO_DEVICE;
    const size_t nn = size_t(O2_CTX) * mhk::TB_NODE_STRIDE;
    std::vector<uint16_t> left(nn), right(nn);
    std::vector<uint8_t> sym(nn);
    std::vector<uint32_t> meta(size_t(O2_CTX) * mhk::TB_META_STRIDE);
    HIP_TRY(hipMemcpy(left.data(), m->d_node_left, nn * 2, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(right.data(), m->d_node_right, nn * 2, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sym.data(), m->d_node_sym, nn, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(meta.data(), m->d_meta, meta.size() * 4, hipMemcpyDeviceToHost));
    mh::BitWriter w;
    w.bit(1);
    for (int i = 0; i < 256 + 7; ++i) w.bit(0);                   // the empty order-1 table, zero padded
    for (int i = 0; i < 4; ++i) w.byte(O2_MAGIC[i]);
    std::vector<uint32_t> stack;
    for (uint32_t c = 0; c < O2_CTX; ++c) {
        const uint32_t root = meta[size_t(c) * mhk::TB_META_STRIDE + 1];
        const uint16_t *l = &left[size_t(c) * mhk::TB_NODE_STRIDE], *r = &right[size_t(c) * mhk::TB_NODE_STRIDE];
        const uint8_t *sy = &sym[size_t(c) * mhk::TB_NODE_STRIDE];
        w.bit(root != 0xFFFFFFFFu);
        if (root == 0xFFFFFFFFu) continue;
        stack.assign(1, root);                                    // pre-order: inner -> 0, leaf -> 1 + symbol (src/huffman.cpp:174-188)
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            if (l[i] == 0xFFFF) { w.bit(1); w.byte(sy[i]); }
            else { w.bit(0); stack.push_back(r[i]); stack.push_back(l[i]); }
        }
    }
    out = w.bytes();
    return MH_OK;
}
}  // namespace

size_t mh_dev_model_workspace(int order) { return order == 1 ? build_layout().fixed + MODEL_WS_SEC_BYTES + MODEL_WS_TILE_BYTES : 0; }

int mh_dev_model_from_counts_ws(const uint64_t *d_counts, int order, void *d_ws, size_t ws_bytes, void *stream, mh_model **out) {
    if (!d_counts || !out || order != 1 || !d_ws) return MH_ERR_ARG;
    if (!have_device()) return MH_ERR_NO_DEVICE;
    const int rc = dev_model_build(d_counts, d_ws, ws_bytes, static_cast<hipStream_t>(stream), out);
    // the rare model the device packer does not lay out: built on the host instead (that model owns its memory)
    if (rc == BUILD_NEEDS_HOST) return model_from_device_counts_via_host(d_counts, order, static_cast<hipStream_t>(stream), out);
    return rc;
}

size_t mh_dev_model2_workspace(void) { return build2_layout().total; }

int mh_dev_model2_array(int which, size_t *offset, size_t *bytes_per_context) {
    // the per-context arrays a slice build fills: 0 code lengths, 1 codewords, 2..5 tree nodes (left, right, symbol,
    // height), 6 per-context sizes
    static const int idx[7] = {0, 1, 5, 6, 7, 8, 9};
    static const size_t per[7] = {256, 256 * 8, size_t(mhk::TB_NODE_STRIDE) * 2, size_t(mhk::TB_NODE_STRIDE) * 2, size_t(mhk::TB_NODE_STRIDE),
                                  size_t(mhk::TB_NODE_STRIDE), size_t(mhk::TB_META_STRIDE) * 4};
    if (which < 0 || which >= 7 || !offset || !bytes_per_context) return MH_ERR_ARG;
    *offset = build2_layout().off[idx[which]];
    *bytes_per_context = per[which];
    return MH_OK;
}

int mh_dev_model2_build_slice(const uint64_t *d_counts_slice, uint32_t ctx_first, uint32_t ctx_end, void *d_ws, size_t ws_bytes, void *stream) {
    if (!d_counts_slice || !d_ws || !aligned16(d_ws) || ws_bytes < build2_layout().total) return MH_ERR_ARG;
    if (!have_device()) return MH_ERR_NO_DEVICE;
    return build2_slice(d_counts_slice, ctx_first, ctx_end, static_cast<unsigned char *>(d_ws), static_cast<hipStream_t>(stream));
}

int mh_dev_model2_finish(void *d_ws, size_t ws_bytes, void *stream, mh_model **out) {
    if (!d_ws || !out || !aligned16(d_ws) || ws_bytes < build2_layout().total) return MH_ERR_ARG;
    if (!have_device()) return MH_ERR_NO_DEVICE;
    return build2_finish(static_cast<unsigned char *>(d_ws), false, static_cast<hipStream_t>(stream), out);
}

int mh_dev_model_from_counts(const uint64_t *d_counts, int order, void *stream, mh_model **out) {
    if (!d_counts || !out || order < 0 || order > 2) return MH_ERR_ARG;
    if (!have_device()) return MH_ERR_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (order == 2) return dev_model_build2(d_counts, st, out);
    if (order == 0) return model_from_device_counts_via_host(d_counts, order, st, out);   // one tree: not worth a kernel
    int rc = dev_model_build(d_counts, nullptr, 0, st, out);
    if (rc == BUILD_NEEDS_HOST) return model_from_device_counts_via_host(d_counts, order, st, out);
    return rc;
}

int mh_model_from_table_bits(const uint8_t *bytes, size_t n, mh_model **out) {
    if ((!bytes && n) || !out) return MH_ERR_ARG;
    if (is_o2_table(bytes, n)) return model2_from_table(bytes, n, out);
    mh_model *m = new (std::nothrow) mh_model;
    if (!m) return MH_ERR_NOMEM;
    if (!m->host.load_table(bytes, n)) { delete m; return MH_ERR_BADTABLE; }
    return finish_model(m, out);
}

int mh_model_write_table(const mh_model *m, uint8_t *out, size_t cap, size_t *nbytes) {
    if (!m || !nbytes) return MH_ERR_ARG;
    std::vector<uint8_t> t;
    if (m->type == 2) { int rc = model2_write_table(m, t); if (rc != MH_OK) return rc; }
    else {
        int rc = ensure_mirror(m); if (rc != MH_OK) return rc;
        t = m->host.save_table();
    }
    *nbytes = t.size();
    if (!out) return MH_OK;
    if (cap < t.size()) return MH_ERR_CAPACITY;
    if (!t.empty()) std::memcpy(out, t.data(), t.size());
    return MH_OK;
}

int mh_model_type(const mh_model *m) { return m ? m->type : MH_ERR_ARG; }

int mh_model_max_code_len(const mh_model *m) { return m ? m->max_len : MH_ERR_ARG; }
int mh_model_min_code_len(const mh_model *m) { return m ? m->min_len : MH_ERR_ARG; }

int mh_model_get_code(const mh_model *m, int prev, int sym, int *len, uint64_t *code) {
    if (!m || !len || !code) return MH_ERR_ARG;
    if (m->type == 2) {                                          // prev = the 16-bit context; read straight from the device tables
        if (!m->d_len8) return MH_ERR_NO_DEVICE;
        const size_t i = (size_t(prev & 0xFFFF) << 8) | size_t(sym & 255);
        uint8_t l = 0;
        HIP_TRY(hipMemcpy(&l, m->d_len8 + i, 1, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(code, m->d_code64 + i, 8, hipMemcpyDeviceToHost));
        *len = l;
        return MH_OK;
    }
    { int rc = ensure_mirror(m); if (rc != MH_OK) return rc; }
    const mh::Code &c = m->host.context(prev).code(sym);
    *len = c.len;
    *code = c.len <= 64 ? c.right_aligned() : 0;
    return MH_OK;
}

int mh_model_get_lut(const mh_model *m, int prev, int w, int *present, int *is_internal, int *value, int *depth) {
    if (!m || !present || !is_internal || !value || !depth || m->type == 2) return MH_ERR_ARG;
    { int rc = ensure_mirror(m); if (rc != MH_OK) return rc; }
    const mh::ContextCoder &c = m->host.context(prev);
    int n = c.lut(w);
    *present = n >= 0;
    *is_internal = *value = *depth = 0;
    if (n >= 0) {
        *is_internal = !c.node(n).leaf;
        *value = c.node(n).sym;
        *depth = c.node(n).depth;
    }
    return MH_OK;
}

int mh_model_decode_layout(const mh_model *m, int *primary_bits, int *secondary_entries, int *in_lds) {
    if (!m || !primary_bits || !secondary_entries || !in_lds) return MH_ERR_ARG;
    *primary_bits = m->dec_bits;
    *secondary_entries = int(m->nsec);
    *in_lds = m->dec_lds ? 1 : 0;
    return MH_OK;
}

int mh_model_tile_layout(const mh_model *m, int *primary_bits, int *secondary_bits, int *secondary_entries) {
    if (!m || !primary_bits || !secondary_bits || !secondary_entries) return MH_ERR_ARG;
    const bool o2 = m->type == 2;
    *primary_bits = o2 ? (m->o2_dec_ok ? int(m->o2_p) : 0) : m->tile_p;
    *secondary_bits = o2 ? int(m->o2_h) : m->tile_h;
    *secondary_entries = o2 ? int(m->o2_nsec) : int(m->tile_nsec);
    return MH_OK;
}

int mh_model_image(const mh_model *m, int which, void *out, size_t cap, size_t *bytes) {
    if (!m || !bytes) return MH_ERR_ARG;
    if (!m->d_len8) return MH_ERR_NO_DEVICE;
    const void *src = nullptr;
    size_t n = 0;
    const size_t nc = m->nctx;                                   // 256, or 65536 for an order-2 model (which has no enc16 / len_slot)
    switch (which) {
        case 0: src = m->d_enc16; n = m->d_enc16 ? 65536 * 2 : 0; break;
        case 1: src = m->d_len8; n = nc * 256; break;
        case 2: src = m->d_len_slot; n = m->d_len_slot ? 65536 : 0; break;
        case 3: src = m->d_code64; n = nc * 256 * 8; break;
        case 4: src = m->d_prim; n = (nc << m->dec_bits) * 2; break;
        case 5: src = m->d_sec; n = size_t(m->nsec) * 2; break;
        case 6: src = m->d_sec_base; n = nc * 4; break;
        case 7: src = m->d_tree; n = nc * mh::TREE_STRIDE * 4; break;
        case 8: src = m->d_tprim; n = m->tile_p ? (size_t(256) << m->tile_p) * 2 : 0; break;
        case 9: src = m->d_tsec; n = size_t(m->tile_nsec) * 2; break;
        default: return MH_ERR_ARG;
    }
    *bytes = n;
    if (!out) return MH_OK;
    if (cap < n) return MH_ERR_CAPACITY;
    if (n) HIP_TRY(hipMemcpy(out, src, n, hipMemcpyDeviceToHost));
    return MH_OK;
}

void mh_model_free(mh_model *m) {
    if (!m) return;
    if (m->d_block) (void)hipFree(m->d_block);
    if (m->d_build) {
        if (m->type == 2 && m->build_cached) build2_block_give(m->d_build, m->device);   // (waits for nothing: the caller has finished with the model)
        else (void)hipFree(m->d_build);
    }
    if (m->d_sec_own) (void)hipFree(m->d_sec_own);
    if (m->d_tile_own) (void)hipFree(m->d_tile_own);
    if (m->d_o2hot) (void)hipFree(m->d_o2hot);
    delete m;
}

}  // extern "C"
