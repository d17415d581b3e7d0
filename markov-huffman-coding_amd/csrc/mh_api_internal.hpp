// mh_api_internal.hpp — what the translation units of the C ABI share (mh_api.cpp: errors, device memory, staging, the mh_dev_*
// compute calls; mh_api_model.cpp: model building, tables, queries; mh_api_host.cpp: the host-buffer calls).  Not part of the
// boundary — that is include/mh.h.  [r5] mh_api.cpp was one file of 1 900 lines.
#pragma once
#include "../../include/mh.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "mh_kernels.h"
#include "mh_model.hpp"

struct mh_model {
    mh::Model host;              // host mirror of the trees (built lazily after a device build)
    mh::Model::Packed packed;    // host-built images (empty after a device build)
    int device = -1;
    // what the entry points need without touching the mirror
    int type = 1, max_len = 0, dec_bits = 8, dec_h = 0;   // type 2: order-2 contexts (extension, parity unpinned)
    uint32_t nctx = 256;         // contexts the device tables are laid out for (65536 for type 2)
    std::vector<uint8_t> table2; // type 2 loaded from a table file: the file itself (write_table returns it)
    uint32_t len_gcd = 0;        // gcd of all code lengths (index builder: segment length is a multiple of it)
    int min_len = 0;             // the shortest code of any context (0: no codes at all): a stream of nbits holds at most nbits / min_len symbols
    bool dec_lds = true, dec_direct = false;
    uint32_t nsec = 0;
    // device build: node arrays stay on the device until somebody asks for the mirror
    bool mirror_ready = true;
    std::mutex mu;
    void *d_build = nullptr;     // enc/dec images + node arrays + meta (device build)
    bool build_cached = false;   // order 2: d_build goes back to the one-entry block cache when the model is freed
    void *d_sec_own = nullptr;   // second-level tables (device build)
    uint16_t *d_node_left = nullptr, *d_node_right = nullptr;
    uint8_t *d_node_sym = nullptr;
    uint32_t *d_meta = nullptr;
    // device images (owned)
    uint16_t *d_enc16 = nullptr;
    uint8_t *d_len8 = nullptr;
    uint8_t *d_len_slot = nullptr;
    uint64_t *d_code64 = nullptr;
    uint64_t *d_enc64 = nullptr;           // order 2: len << 56 | code (mhk::launch_enc64_pack), part of d_build
    uint16_t *d_prim = nullptr;
    uint16_t *d_sec = nullptr;
    uint32_t *d_sec_base = nullptr;
    uint32_t *d_tree = nullptr;
    void *d_block = nullptr;     // the one allocation all of the above point into
    // tile decoder tables (mh_tile.hip; LSB-first indexed; tile_p == 0: none)
    int tile_p = 0, tile_h = 0;
    uint32_t tile_nsec = 0;
    uint16_t *d_tprim = nullptr, *d_tsec = nullptr;
    void *d_tile_own = nullptr;  // their allocation when the model owns it
    // order 2: tables of the live contexts, one slot each (dev_model_build2): the encoder's LDS image and the tile
    // decoder's tables; o2_enc_ok / o2_dec_ok say whether they cover the model well enough to be used
    void *d_o2hot = nullptr;     // one allocation: image | ctx2slot | slot_ctx | tprim | tsec
    uint8_t *d_o2img = nullptr; uint32_t o2img_bytes = 0;
    uint16_t *d_ctx2slot = nullptr;
    uint32_t *d_tprim2 = nullptr, *d_tsec2 = nullptr;
    uint32_t o2_nslots = 0, o2_p = 0, o2_h = 0, o2_nsec = 0;
    bool o2_enc_ok = false, o2_dec_ok = false;
};

namespace mhapi {


extern thread_local int g_last_hip;
extern thread_local int g_encode_retries;    // segments of the calling thread's last mh_encode* that the one-pass order-2 encoder gave up on
extern std::atomic<uint64_t> g_encode_retries_total;   // ... of all threads since the library was loaded
extern thread_local int g_last_index_path;   // how the calling thread's last index-free mh_decode* built its index (mh_last_index_path)

inline int hip_fail(hipError_t e) {
    g_last_hip = int(e);
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) ? MH_ERR_NO_DEVICE : MH_ERR_HIP;
}

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return hip_fail(_e);      \
    } while (0)

inline bool have_device() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline uint32_t gcd_u32(uint32_t a, uint32_t b) {
    while (b) { const uint32_t t = a % b; a = b; b = t; }
    return a;
}

// first-level width of the tile decoder's tables (mh_tile.hip): the LDS left beside 256 << P entries is what the
// waves stage their input in, so P trades table hits against waves in flight.  MH_TILE_P overrides (5..8; 0: no
// tile tables).
inline int tile_p_choice() {                     // (read at every model build: tests vary it inside one process)
    const char *e = getenv("MH_TILE_P");
    const int p = e ? atoi(e) : 7;
    return p == 0 ? 0 : (p < 5 ? 5 : (p > 8 ? 8 : p));
}

inline int chunk_shift_of(uint32_t chunk_symbols) {
    if (chunk_symbols < MH_CHUNK_MIN || chunk_symbols > MH_CHUNK_MAX) return -1;
    if (chunk_symbols & (chunk_symbols - 1)) return -1;
    int s = 0;
    while ((1u << s) != chunk_symbols) ++s;
    return s;
}

inline int status_from_device(int s) {
    switch (s) {
        case mhk::MHK_STATUS_OK: return MH_OK;
        case mhk::MHK_STATUS_TIMEOUT: return MH_ERR_TIMEOUT;
        case mhk::MHK_STATUS_CAPACITY: return MH_ERR_CAPACITY;
        default: return MH_ERR_CORRUPT;
    }
}

// RAII device buffer for the host-buffer convenience calls
// the shortest code of a device-built context from its meta record (mh_kernels.h, TB_META_STRIDE): mt[15] has bit l - 1 set
// for every code length l in use, and is 0 for a one-symbol context, whose only code is one bit long (mt[2] = 1)
inline void note_min_len(mh_model *m, const uint32_t *mt) {
    int l = 0;
    if (mt[15]) l = __builtin_ctz(mt[15]) + 1;
    else if (mt[2] >= 1) l = 1;
    if (l && (m->min_len == 0 || l < m->min_len)) m->min_len = l;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T> T *as() { return static_cast<T *>(p); }
};

// the context in front of a stream's first symbol as the kernels want it (order 2: two bytes)
inline uint32_t ctx_of_prev0(const mh_model *m, uint8_t prev0) { return m && m->type == 2 ? (uint32_t(prev0) << 8 | prev0) : prev0; }

// MH_TIMING=1: the host-buffer calls account their time to three phases — upload (caller memory -> HBM, the page-cache
// read of a mapped file included), device (kernels, waited for), download (HBM -> caller memory, the page faults of
// a fresh file mapping included) — and print one stderr line per phase in the CLI's [mh-timing] format, so that
// tools/cli_rate.py can tell the pipeline's rate from the file system's.  Each phase is waited for before the next
// starts when timing is on (the calls overlap them otherwise).
struct PhaseClock {
    bool on = getenv("MH_TIMING") != nullptr;
    double upload = 0, device = 0, download = 0;
    size_t up_bytes = 0, down_bytes = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void report(const char *call, size_t n) const {
        if (!on) return;
        fprintf(stderr, "[mh-timing] %s.upload %zu bytes %.4f s %.2f GB/s\n", call, up_bytes, upload, upload > 0 ? up_bytes / upload / 1e9 : 0.0);
        fprintf(stderr, "[mh-timing] %s.device %zu bytes %.4f s %.2f GB/s\n", call, n, device, device > 0 ? n / device / 1e9 : 0.0);
        fprintf(stderr, "[mh-timing] %s.download %zu bytes %.4f s %.2f GB/s\n", call, down_bytes, download, download > 0 ? down_bytes / download / 1e9 : 0.0);
        if (retries) fprintf(stderr, "[mh-timing] %s.retries %d (one-pass encoder timed out: segments encoded again with the two-pass pair)\n", call, retries);
    }
    int retries = 0;
};

extern thread_local PhaseClock *g_phase;
// set around a retry of the one-pass order-2 encoder (mh_api.cpp, dev_encode_ctx)
extern thread_local bool t_no_chain;

// staging between the callers' pageable buffers and HBM (mh_api.cpp: pinned ring, several copy threads)
hipError_t stage_h2d(void *d_dst, const void *h_src, size_t n, hipStream_t st);
hipError_t stage_d2h(void *h_dst, const void *d_src, size_t n, hipStream_t st);
size_t segment_bytes();
// mh_api_model.cpp
int upload_model(mh_model *m);
int ensure_mirror(const mh_model *cm);
int finish_model(mh_model *m, mh_model **out);

}  // namespace mhapi
