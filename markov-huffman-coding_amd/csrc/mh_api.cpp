// mh_api.cpp — the C ABI of include/mh.h on top of the HIP kernels: errors, device memory, staging, and the mh_dev_* compute
// calls (device pointers + stream).  Models: mh_api_model.cpp; host-buffer calls: mh_api_host.cpp.  No CPU compute fallback: every
// compute entry point needs a device and returns MH_ERR_NO_DEVICE without one.
#include "mh_api_internal.hpp"

namespace mhapi {

thread_local int g_last_hip = 0;
thread_local int g_encode_retries = 0;    // segments of the calling thread's last mh_encode* that the one-pass order-2 encoder gave up on
std::atomic<uint64_t> g_encode_retries_total{0};   // ... of all threads since the library was loaded
thread_local int g_last_index_path = 0;   // how the calling thread's last index-free mh_decode* built its index (mh_last_index_path)

// ---- pinned staging for the host-buffer calls ---------------------------------------
// The callers' buffers are pageable (the CLI hands over mmapped files): a plain hipMemcpy from such memory
// is a single-threaded bounce through the runtime's own staging buffers.  Here a ring of pinned pieces is
// filled by several host threads (the page-cache copy is what bounds a transfer, not the bus) while the DMA
// of the previous piece runs, and the same in the other direction.  One ring per process, guarded by a mutex
// (the host-buffer calls of one process take turns on the bus anyway).
class PinnedRing {
public:
    static constexpr size_t PIECE = size_t(16) << 20;
    static constexpr int SLOTS = 4;
    // pageable host -> device, stream-ordered on `st` for the device side; returns when the last piece has been queued
    hipError_t upload(void *d_dst, const void *h_src, size_t n, hipStream_t st) {
        std::lock_guard<std::mutex> lock(mu_);
        hipError_t e = ensure();
        if (e != hipSuccess) return e;
        const unsigned char *src = static_cast<const unsigned char *>(h_src);
        unsigned char *dst = static_cast<unsigned char *>(d_dst);
        for (size_t off = 0; off < n; off += PIECE, ++seq_) {
            const size_t len = std::min(PIECE, n - off);
            const int s = int(seq_ % SLOTS);
            if ((e = hipEventSynchronize(ev_[s])) != hipSuccess) return e;      // the slot's previous transfer has left it
            parallel_copy(buf_[s], src + off, len);
            if ((e = hipMemcpyAsync(dst + off, buf_[s], len, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
            if ((e = hipEventRecord(ev_[s], st)) != hipSuccess) return e;
        }
        return hipSuccess;
    }
    // device -> pageable host; everything queued on `st` before the call is waited for; returns when h_dst is complete
    hipError_t download(void *h_dst, const void *d_src, size_t n, hipStream_t st) {
        std::lock_guard<std::mutex> lock(mu_);
        hipError_t e = ensure();
        if (e != hipSuccess) return e;
        unsigned char *dst = static_cast<unsigned char *>(h_dst);
        const unsigned char *src = static_cast<const unsigned char *>(d_src);
        const size_t pieces = (n + PIECE - 1) / PIECE;
        // piece i is copied out of its slot while pieces i+1 .. i+SLOTS-1 are on the bus
        for (size_t i = 0; i < pieces + SLOTS - 1; ++i) {
            if (i < pieces) {
                const size_t off = i * PIECE, len = std::min(PIECE, n - off);
                const int s = int((seq_ + i) % SLOTS);
                if (i < size_t(SLOTS) && (e = hipEventSynchronize(ev_[s])) != hipSuccess) return e;
                if ((e = hipMemcpyAsync(buf_[s], src + off, len, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
                if ((e = hipEventRecord(ev_[s], st)) != hipSuccess) return e;
            }
            if (i + 1 >= size_t(SLOTS)) {
                const size_t j = i + 1 - SLOTS;                                 // oldest piece in flight
                const size_t off = j * PIECE, len = std::min(PIECE, n - off);
                const int s = int((seq_ + j) % SLOTS);
                if ((e = hipEventSynchronize(ev_[s])) != hipSuccess) return e;
                parallel_copy(dst + off, buf_[s], len);
            }
        }
        seq_ += pieces;
        return hipSuccess;
    }

private:
    hipError_t ensure() {
        if (ready_) return hipSuccess;
        for (int i = 0; i < SLOTS; ++i) {
            hipError_t e = hipHostMalloc(&buf_[i], PIECE, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming)) != hipSuccess) return e;
        }
        unsigned hw = std::thread::hardware_concurrency();
        threads_ = hw >= 16 ? 6 : hw >= 8 ? 4 : hw >= 4 ? 2 : 1;
        if (const char *t = getenv("MH_COPY_THREADS")) { const int v = atoi(t); if (v >= 1 && v <= 32) threads_ = v; }
        ready_ = true;
        return hipSuccess;
    }
    void parallel_copy(void *dst, const void *src, size_t n) const {
        if (threads_ <= 1 || n < (size_t(1) << 20)) { std::memcpy(dst, src, n); return; }
        const size_t part = ((n / size_t(threads_)) + 4095) & ~size_t(4095);
        std::vector<std::thread> pool;
        for (int t = 1; t < threads_; ++t) {
            const size_t off = part * size_t(t);
            if (off >= n) break;
            pool.emplace_back([=] { std::memcpy(static_cast<unsigned char *>(dst) + off, static_cast<const unsigned char *>(src) + off, std::min(part, n - off)); });
        }
        std::memcpy(dst, src, std::min(part, n));
        for (std::thread &t : pool) t.join();
    }
    std::mutex mu_;
    bool ready_ = false;
    void *buf_[SLOTS] = {};
    hipEvent_t ev_[SLOTS] = {};
    size_t seq_ = 0;
    int threads_ = 1;
};
// one ring per device: its events belong to the device that was current when they were created (an event recorded
// on another device's stream is refused), and a process may drive several cards (mh_set_device)
constexpr int MAX_RING_DEVICES = 16;
PinnedRing g_rings[MAX_RING_DEVICES];
int ring_slot_for_device(int dev) { return dev >= 0 && dev < MAX_RING_DEVICES ? dev : 0; }
PinnedRing &ring_for_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return g_rings[ring_slot_for_device(dev)];
}

thread_local PhaseClock *g_phase = nullptr;

// small transfers keep the plain call (the ring pays from a few MiB on)
hipError_t stage_h2d(void *d_dst, const void *h_src, size_t n, hipStream_t st) {
    const double t0 = g_phase && g_phase->on ? PhaseClock::now() : 0;
    hipError_t e = n < (size_t(4) << 20) ? hipMemcpyAsync(d_dst, h_src, n, hipMemcpyHostToDevice, st)   // pageable: returns after staging
                                        : ring_for_current_device().upload(d_dst, h_src, n, st);
    if (g_phase && g_phase->on && e == hipSuccess) {
        e = hipStreamSynchronize(st);
        g_phase->upload += PhaseClock::now() - t0;
        g_phase->up_bytes += n;
    }
    return e;
}
// Device -> caller memory.  Measured on the GPU box (tools/cli_rate.py, 4 GiB): going through the ring costs
// MORE than the runtime's own pageable path when the destination is a freshly grown file mapping — the
// runtime pins the destination pages in place (they are allocated in bulk inside the kernel) and lets the
// DMA engine write them, whereas copying out of the ring takes a user-space page fault per 4 KiB, and several
// threads doing so contend (compress 2.3 GB/s with the ring against 4.9 GB/s without).  MH_D2H_RING=1 selects
// the ring for destinations that are already resident.
hipError_t stage_d2h(void *h_dst, const void *d_src, size_t n, hipStream_t st) {
    static const bool use_ring = getenv("MH_D2H_RING") && atoi(getenv("MH_D2H_RING")) != 0;
    const bool timing = g_phase && g_phase->on;
    double t0 = 0;
    if (timing) {                                                // what is still running on the device belongs to the device phase
        const double td = PhaseClock::now();
        (void)hipStreamSynchronize(st);
        t0 = PhaseClock::now();
        g_phase->device += t0 - td;
    }
    hipError_t e;
    if (use_ring && n >= (size_t(4) << 20)) e = ring_for_current_device().download(h_dst, d_src, n, st);
    else {
        e = hipMemcpyAsync(h_dst, d_src, n, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (timing) { g_phase->download += PhaseClock::now() - t0; g_phase->down_bytes += n; }
    return e;
}

// Host-buffer calls stage their data through HBM in segments, so their device footprint is bounded
// whatever the input size (a 16 GiB file does not need 16 GiB + its worst-case payload on the card).
// MH_SEGMENT_BYTES overrides the 256 MiB default (tests use small values to put seams everywhere).
// The segment size is a multiple of the largest chunk size, so chunk boundaries fall on segment
// boundaries.
size_t segment_bytes() {
    size_t s = size_t(256) << 20;
    if (const char *e = getenv("MH_SEGMENT_BYTES")) {
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v) s = size_t(v);
    }
    s &= ~size_t(MH_CHUNK_MAX - 1);
    return s < MH_CHUNK_MAX ? size_t(MH_CHUNK_MAX) : s;
}

}  // namespace mhapi

using namespace mhapi;

extern "C" {

const char *mh_strerror(int status) {
    switch (status) {
        case MH_OK: return "ok";
        case MH_ERR_ARG: return "invalid argument";
        case MH_ERR_NO_DEVICE: return "no usable HIP device (the codec has no CPU fallback)";
        case MH_ERR_HIP: return "HIP runtime error";
        case MH_ERR_CORRUPT: return "Input appears corrupt";
        case MH_ERR_TYPE: return "File encoding method does not match provided encoding table";
        case MH_ERR_BADTABLE: return "encoding table not parseable";
        case MH_ERR_CODE_TOO_LONG: return "codeword longer than 64 bits";
        case MH_ERR_CAPACITY: return "output buffer or workspace too small";
        case MH_ERR_TIMEOUT: return "device-side wait expired";
        case MH_ERR_NOMEM: return "out of memory";
        default: return "unknown error";
    }
}

int mh_last_hip_error(void) { return g_last_hip; }
int mh_last_index_path(void) { return g_last_index_path; }
int mh_last_encode_retries(void) { return g_encode_retries; }
uint64_t mh_total_encode_retries(void) { return g_encode_retries_total.load(std::memory_order_relaxed); }

int mh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mh_set_device(int ordinal) {
    HIP_TRY(hipSetDevice(ordinal));
    return MH_OK;
}

int mh_dev_malloc(void **d_ptr, size_t bytes) {
    if (!d_ptr) return MH_ERR_ARG;
    if (!have_device()) return MH_ERR_NO_DEVICE;
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 16));
    return MH_OK;
}
int mh_dev_free(void *d_ptr) {
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return MH_OK;
}
int mh_dev_upload(void *d_dst, const void *h_src, size_t bytes) {
    if ((!d_dst || !h_src) && bytes) return MH_ERR_ARG;
    if (bytes) HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return MH_OK;
}
int mh_dev_download(void *h_dst, const void *d_src, size_t bytes) {
    if ((!h_dst || !d_src) && bytes) return MH_ERR_ARG;
    if (bytes) HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return MH_OK;
}

/* ------------------------------------------------------------ device calls */

size_t mh_dev_histogram_workspace(size_t n) { return have_device() ? mhk::hist_workspace_bytes(n) : 64; }

int mh_dev_histogram_o1(const uint8_t *d_data, size_t n, uint8_t prev0, uint64_t *d_counts, void *d_ws, size_t ws_bytes, void *stream) {
    if ((!d_data && n) || !d_counts || !aligned16(d_data)) return MH_ERR_ARG;
    HIP_TRY(mhk::launch_hist_o1(d_data, n, prev0, reinterpret_cast<unsigned long long *>(d_counts), d_ws, ws_bytes,
                                static_cast<hipStream_t>(stream)));
    return MH_OK;
}

int mh_dev_histogram_o0(const uint8_t *d_data, size_t n, uint64_t *d_counts, void *, size_t, void *stream) {
    if ((!d_data && n) || !d_counts || !aligned16(d_data)) return MH_ERR_ARG;
    HIP_TRY(mhk::launch_hist_o0(d_data, n, reinterpret_cast<unsigned long long *>(d_counts), static_cast<hipStream_t>(stream)));
    return MH_OK;
}

size_t mh_dev_histogram_o2_workspace(size_t n) { return mhk::hist2_workspace_bytes(n); }

int mh_dev_histogram_o2_ws(const uint8_t *d_data, size_t n, uint16_t ctx0, uint64_t *d_counts, void *d_ws, size_t ws_bytes, void *stream) {
    if ((!d_data && n) || !d_counts || !aligned16(d_data)) return MH_ERR_ARG;
    if (d_ws && (reinterpret_cast<uintptr_t>(d_ws) & 255u)) return MH_ERR_ARG;
    HIP_TRY(mhk::launch_hist_o2(d_data, n, ctx0, reinterpret_cast<unsigned long long *>(d_counts), d_ws, ws_bytes,
                                static_cast<hipStream_t>(stream)));
    return MH_OK;
}

int mh_dev_histogram_o2(const uint8_t *d_data, size_t n, uint16_t ctx0, uint64_t *d_counts, void *stream) {
    return mh_dev_histogram_o2_ws(d_data, n, ctx0, d_counts, nullptr, 0, stream);
}

// The encode workspace also has room for a region-mode histogram of the input (workspace + 65536 counts): an
// order-0/1 encode that comes without one (mh_dev_encode, mh_dev_encode_at, mh_encode: the reference's `-e table`
// flow, src/main.cpp:137-161 + 208-212) takes it first and then runs the region encoder — 5.8 + 10 ms per 16 GiB
// against 4.1 + 13.7 ms for the length pass + emit pair, and the same bytes out.
static size_t enc_ws_core(size_t n) { return (mhk::encode_workspace_bytes(n) + 255) & ~size_t(255); }
static size_t enc_ws_hist(size_t n) { return (mhk::hist_workspace_bytes(n) + 255) & ~size_t(255); }
constexpr size_t ENC_OWN_HIST_MIN = size_t(4) << 20;   // below this the length pass costs nothing worth a histogram
size_t mh_dev_encode_workspace(size_t n) {
    return enc_ws_core(n) + (n >= ENC_OWN_HIST_MIN && have_device() ? enc_ws_hist(n) + 65536 * 8 : 0);
}

int mh_dev_encode(const mh_model *m, const uint8_t *d_data, size_t n, uint8_t prev0, uint8_t *d_payload, size_t cap,
                  uint64_t *d_nbits, uint64_t *d_index, uint32_t chunk_symbols, void *d_ws, size_t ws_bytes, void *stream) {
    return mh_dev_encode_at(m, d_data, n, prev0, nullptr, d_payload, cap, d_nbits, d_index, chunk_symbols, d_ws, ws_bytes, stream);
}

int mh_dev_payload_bits(const mh_model *m, const uint64_t *d_counts, uint64_t *d_nbits, void *stream) {
    if (!m || !d_counts || !d_nbits) return MH_ERR_ARG;
    if (!m->d_len8) return MH_ERR_NO_DEVICE;
    HIP_TRY(mhk::launch_payload_bits(reinterpret_cast<const unsigned long long *>(d_counts), m->d_len8, m->type == 2 ? (1u << 24) : m->type ? 65536u : 256u,
                                     reinterpret_cast<unsigned long long *>(d_nbits), static_cast<hipStream_t>(stream)));
    return MH_OK;
}

static int dev_encode_hist(const mh_model *m, const uint8_t *d_data, size_t n, uint8_t prev0, const uint64_t *d_start_bit,
                           uint8_t *d_payload, size_t cap, uint64_t *d_nbits, uint64_t *d_index, uint32_t chunk_symbols,
                           const void *d_hist_ws, size_t hist_ws_bytes, void *d_ws, size_t ws_bytes, void *stream, uint32_t *d_fine);

// ctx0: the context before the first byte — the previous byte (orders 0/1) or, for an order-2 model,
// (byte before previous) << 8 | previous byte
// set around a retry: the one-pass order-2 encoder gave up waiting (its workgroups were not all resident: a device shared
// with a long-running kernel of somebody else), the host-side callers that synchronise anyway run the two-pass pair instead
}  // extern "C" (two items the host-buffer calls share: mh_api_internal.hpp)
namespace mhapi { thread_local bool t_no_chain = false; }
extern "C" {

static int dev_encode_ctx(const mh_model *m, const uint8_t *d_data, size_t n, uint32_t ctx0, const uint64_t *d_start_bit,
                          uint8_t *d_payload, size_t cap, uint64_t *d_nbits, uint64_t *d_index, uint32_t chunk_symbols,
                          void *d_ws, size_t ws_bytes, void *stream, uint32_t *d_fine) {
    if (!m || (!d_data && n) || !d_payload || !d_nbits || !d_ws) return MH_ERR_ARG;
    if (!aligned16(d_data) || !aligned16(d_payload) || !aligned16(d_ws)) return MH_ERR_ARG;
    int shift = chunk_shift_of(d_index ? chunk_symbols : MH_CHUNK_DEFAULT);
    if (shift < 0) return MH_ERR_ARG;
    if (ws_bytes < mhk::encode_workspace_bytes(n)) return MH_ERR_CAPACITY;
    if (m->max_len > mh::MAX_CODE_BITS) return MH_ERR_CODE_TOO_LONG;
    if (!m->d_len8) return MH_ERR_NO_DEVICE;
    if (m->type != 2 && n >= ENC_OWN_HIST_MIN && ws_bytes >= enc_ws_core(n) + enc_ws_hist(n) + 65536 * 8 && !getenv("MH_ENCODE_LENGTH_PASS")) {
        // no histogram came with the call: take one of this very buffer (region mode) and let the region encoder price
        // its regions from it — one more read of the input, but no length pass and the faster emit
        unsigned char *w = static_cast<unsigned char *>(d_ws);
        void *hws = w + enc_ws_core(n);
        uint64_t *cnt = reinterpret_cast<uint64_t *>(w + enc_ws_core(n) + enc_ws_hist(n));
        HIP_TRY(mhk::launch_hist_o1(d_data, n, ctx0, reinterpret_cast<unsigned long long *>(cnt), hws, enc_ws_hist(n), static_cast<hipStream_t>(stream)));
        return dev_encode_hist(m, d_data, n, uint8_t(ctx0), d_start_bit, d_payload, cap, d_nbits, d_index, chunk_symbols, hws, enc_ws_hist(n),
                               d_ws, enc_ws_core(n), stream, d_fine);
    }
    mhk::EncodeArgs p{};
    p.order = m->type == 2 ? 2 : 1;
    p.data = d_data; p.n = n; p.prev0 = ctx0; p.chunk_shift = uint32_t(shift);
    p.out = d_payload; p.cap = cap;
    p.enc16 = m->d_enc16; p.len_slot = m->d_len_slot; p.len8 = m->d_len8; p.code64 = m->d_code64; p.enc64 = m->type == 2 ? m->d_enc64 : nullptr;
    p.nbits = reinterpret_cast<unsigned long long *>(d_nbits);
    p.index = reinterpret_cast<unsigned long long *>(d_index);
    p.start_bit = reinterpret_cast<const unsigned long long *>(d_start_bit);
    p.fine = d_fine;
    if (m->type == 2 && m->o2_enc_ok) { p.o2hot = m->d_o2img; p.o2hot_bytes = m->o2img_bytes; }
    p.no_chain = t_no_chain;
    HIP_TRY(mhk::launch_encode(p, d_ws, static_cast<hipStream_t>(stream)));
    return MH_OK;
}

static int dev_encode_hist(const mh_model *m, const uint8_t *d_data, size_t n, uint8_t prev0, const uint64_t *d_start_bit,
                           uint8_t *d_payload, size_t cap, uint64_t *d_nbits, uint64_t *d_index, uint32_t chunk_symbols,
                           const void *d_hist_ws, size_t hist_ws_bytes, void *d_ws, size_t ws_bytes, void *stream, uint32_t *d_fine) {
    if (!m || (!d_data && n) || !d_payload || !d_nbits || !d_ws) return MH_ERR_ARG;
    // order-2 models take the regular path; so does a caller without the workspace (codes over 12 bits are escapes
    // inside the region encoder: src/bitbuffer.cpp:45-73 appends descriptors of any length)
    if (m->type == 2 || !d_hist_ws || hist_ws_bytes < mhk::hist_workspace_bytes(n))
        return dev_encode_ctx(m, d_data, n, m->type == 2 ? (uint32_t(prev0) << 8 | prev0) : prev0, d_start_bit, d_payload, cap, d_nbits, d_index,
                              chunk_symbols, d_ws, ws_bytes, stream, d_fine);
    if (!aligned16(d_data) || !aligned16(d_payload) || !aligned16(d_ws) || !aligned16(d_hist_ws)) return MH_ERR_ARG;
    int shift = chunk_shift_of(d_index ? chunk_symbols : MH_CHUNK_DEFAULT);
    if (shift < 0) return MH_ERR_ARG;
    if (ws_bytes < mhk::encode_workspace_bytes(n)) return MH_ERR_CAPACITY;
    if (m->max_len > mh::MAX_CODE_BITS) return MH_ERR_CODE_TOO_LONG;
    if (!m->d_enc16) return MH_ERR_NO_DEVICE;
    mhk::EncodeArgs p{};
    p.order = 1;
    p.data = d_data; p.n = n; p.prev0 = prev0; p.chunk_shift = uint32_t(shift);
    p.out = d_payload; p.cap = cap;
    p.enc16 = m->d_enc16; p.len_slot = m->d_len_slot; p.len8 = m->d_len8; p.code64 = m->d_code64; p.enc64 = nullptr;
    p.nbits = reinterpret_cast<unsigned long long *>(d_nbits);
    p.index = reinterpret_cast<unsigned long long *>(d_index);
    p.start_bit = reinterpret_cast<const unsigned long long *>(d_start_bit);
    p.fine = d_fine;
    p.max_len = m->max_len;
    HIP_TRY(mhk::launch_encode_regions(p, d_hist_ws, hist_ws_bytes, d_ws, static_cast<hipStream_t>(stream)));
    return MH_OK;
}

int mh_dev_encode_hist(const mh_model *m, const uint8_t *d_data, size_t n, uint8_t prev0, const uint64_t *d_start_bit,
                       uint8_t *d_payload, size_t cap, uint64_t *d_nbits, uint64_t *d_index, uint32_t chunk_symbols,
                       const void *d_hist_ws, size_t hist_ws_bytes, void *d_ws, size_t ws_bytes, void *stream) {
    return dev_encode_hist(m, d_data, n, prev0, d_start_bit, d_payload, cap, d_nbits, d_index, chunk_symbols, d_hist_ws, hist_ws_bytes,
                           d_ws, ws_bytes, stream, nullptr);
}

int mh_dev_encode_fine(const mh_model *m, const uint8_t *d_data, size_t n, uint8_t prev0, const uint64_t *d_start_bit,
                       uint8_t *d_payload, size_t cap, uint64_t *d_nbits, uint64_t *d_index, uint32_t chunk_symbols, uint32_t *d_fine,
                       const void *d_hist_ws, size_t hist_ws_bytes, void *d_ws, size_t ws_bytes, void *stream) {
    return dev_encode_hist(m, d_data, n, prev0, d_start_bit, d_payload, cap, d_nbits, d_index, chunk_symbols, d_hist_ws, hist_ws_bytes,
                           d_ws, ws_bytes, stream, d_fine);
}


int mh_dev_encode_ctx(const mh_model *m, const uint8_t *d_data, size_t n, uint32_t ctx0, const uint64_t *d_start_bit,
                      uint8_t *d_payload, size_t cap, uint64_t *d_nbits, uint64_t *d_index, uint32_t chunk_symbols,
                      void *d_ws, size_t ws_bytes, void *stream) {
    if (m && ctx0 > (m->type == 2 ? 0xFFFFu : 0xFFu)) return MH_ERR_ARG;
    return dev_encode_ctx(m, d_data, n, ctx0, d_start_bit, d_payload, cap, d_nbits, d_index, chunk_symbols, d_ws, ws_bytes, stream, nullptr);
}

int mh_dev_encode_ctx_fine(const mh_model *m, const uint8_t *d_data, size_t n, uint32_t ctx0, const uint64_t *d_start_bit,
                           uint8_t *d_payload, size_t cap, uint64_t *d_nbits, uint64_t *d_index, uint32_t chunk_symbols, uint32_t *d_fine,
                           void *d_ws, size_t ws_bytes, void *stream) {
    if (m && ctx0 > (m->type == 2 ? 0xFFFFu : 0xFFu)) return MH_ERR_ARG;
    return dev_encode_ctx(m, d_data, n, ctx0, d_start_bit, d_payload, cap, d_nbits, d_index, chunk_symbols, d_ws, ws_bytes, stream, d_fine);
}

int mh_dev_encode_at(const mh_model *m, const uint8_t *d_data, size_t n, uint8_t prev0, const uint64_t *d_start_bit,
                     uint8_t *d_payload, size_t cap, uint64_t *d_nbits, uint64_t *d_index, uint32_t chunk_symbols,
                     void *d_ws, size_t ws_bytes, void *stream) {
    return dev_encode_ctx(m, d_data, n, ctx_of_prev0(m, prev0), d_start_bit, d_payload, cap, d_nbits, d_index, chunk_symbols, d_ws,
                          ws_bytes, stream, nullptr);
}

size_t mh_dev_decode_workspace(uint64_t, uint64_t n_symbols, uint32_t chunk_symbols) {
    // status block + the redo list (a count and up to one u32 per chunk)
    if (chunk_shift_of(chunk_symbols) < 0) return 0;
    return (64 + 4 * (size_t(mh_index_entries(n_symbols, chunk_symbols)) + 1) + 15) & ~size_t(15);
}

size_t mh_dev_build_index_workspace(uint64_t nbits) { return mhk::build_index_workspace_bytes(nbits); }

// MH_DECODE_PATH=tile|chunk forces the decoder choice (tests, A/B runs); otherwise the tile decoder runs whenever
// a fine index came with the call, the model has tile tables and the stream is large enough to fill the card
static int decode_path_choice() {        // (read at every call: tests switch inside one process)
    const char *e = getenv("MH_DECODE_PATH");
    return !e ? 0 : (e[0] == 't' ? 1 : (e[0] == 'c' ? 2 : 0));
}

static int dev_decode(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, const uint64_t *d_nbits, uint8_t *d_out,
                      uint64_t n_symbols, const uint64_t *d_index, uint32_t chunk_symbols, void *d_ws, size_t ws_bytes, void *stream,
                      const uint32_t *d_fine = nullptr) {
    if (!m || !d_ws || ws_bytes < 64) return MH_ERR_ARG;
    if (ws_bytes < mh_dev_decode_workspace(nbits, n_symbols, chunk_symbols)) return MH_ERR_ARG;
    if (n_symbols && (!d_payload || !d_out || !d_index)) return MH_ERR_ARG;
    if (!aligned16(d_payload) || !aligned16(d_out)) return MH_ERR_ARG;
    int shift = chunk_shift_of(chunk_symbols);
    if (shift < 0) return MH_ERR_ARG;
    if (m->max_len > mh::MAX_CODE_BITS) return MH_ERR_CODE_TOO_LONG;
    if (!m->d_prim) return MH_ERR_NO_DEVICE;
    mhk::DecParams p{};
    p.order = m->type == 2 ? 2 : 1;
    p.payload = d_payload; p.payload_bytes = (nbits + 7) / 8; p.nbits = nbits;
    p.d_nbits = reinterpret_cast<const unsigned long long *>(d_nbits);
    p.out = d_out; p.n = n_symbols;
    p.index = reinterpret_cast<const unsigned long long *>(d_index);
    p.nchunks = mh_index_entries(n_symbols, chunk_symbols);
    p.chunk_shift = uint32_t(shift);
    p.prim = m->d_prim; p.sec = m->d_sec; p.sec_base = m->d_sec_base; p.tree = m->d_tree;
    p.P = uint32_t(m->dec_bits); p.nsec = m->nsec; p.sec_lds = m->dec_lds ? 1u : 0u;
    p.direct = m->dec_direct ? 1u : 0u; p.H = uint32_t(m->dec_h);
    const int path = decode_path_choice();
    const bool tile_ok = d_fine && path != 2 && (m->type == 2 ? (m->o2_dec_ok && shift <= 10) : (m->tile_p && shift <= 12));
    // The tile decoder's first level is tile_p (7) bits wide: when the average code is about that long (near-uniform
    // bytes: 8-bit codes everywhere), nearly every symbol takes the second-level gather and every tile stages a full
    // 4 KiB — the chunk decoder with its 8-bit first level in LDS is 4x faster there (measured: 4 GiB uniform 3.0 vs 13.4
    // ms; Zipf 6.3 vs 4.5 ms; text 4.4 vs 4.0 ms).  nbits == 0 (unknown): the tile decoder.
    const bool long_codes = m->type != 2 && nbits && n_symbols && double(nbits) > (double(m->tile_p) - 0.5) * double(n_symbols);
    if (tile_ok && (path == 1 || (n_symbols >= (uint64_t(8) << 20) && !long_codes))) {
        mhk::TileParams t{};
        t.payload = d_payload; t.payload_bytes = p.payload_bytes; t.nbits = nbits; t.d_nbits = p.d_nbits;
        t.out = d_out; t.n = n_symbols; t.index = p.index; t.nchunks = p.nchunks; t.chunk_shift = p.chunk_shift;
        t.fine = d_fine;
        t.prim = m->d_tprim; t.sec = m->d_tsec; t.P = uint32_t(m->tile_p); t.H = uint32_t(m->tile_h); t.nsec = m->tile_nsec;
        if (m->type == 2) {                                      // the live contexts' tables (32-bit entries)
            t.o2 = 1; t.nslots = m->o2_nslots; t.ctx2slot = m->d_ctx2slot;
            t.prim = reinterpret_cast<const uint16_t *>(m->d_tprim2); t.sec = reinterpret_cast<const uint16_t *>(m->d_tsec2);
            t.P = m->o2_p; t.H = m->o2_h; t.nsec = m->o2_nsec;
        }
        HIP_TRY(mhk::launch_decode_tile(t, p, d_ws, static_cast<hipStream_t>(stream)));
        HIP_TRY(mhk::launch_set_word(reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(d_ws) + 40), mhk::DEC_PATH_TILE, static_cast<hipStream_t>(stream)));
        return MH_OK;
    }
    HIP_TRY(mhk::launch_decode(p, d_ws, static_cast<hipStream_t>(stream)));
    HIP_TRY(mhk::launch_set_word(reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(d_ws) + 40), mhk::DEC_PATH_CHUNK, static_cast<hipStream_t>(stream)));
    return MH_OK;
}

int mh_dev_decode_fine(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, const uint64_t *d_nbits, uint8_t *d_out,
                       uint64_t n_symbols, const uint64_t *d_index, uint32_t chunk_symbols, const uint32_t *d_fine,
                       void *d_ws, size_t ws_bytes, void *stream) {
    return dev_decode(m, d_payload, nbits, d_nbits, d_out, n_symbols, d_index, chunk_symbols, d_ws, ws_bytes, stream, d_fine);
}

int mh_dev_decode(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, uint8_t *d_out, uint64_t n_symbols,
                  const uint64_t *d_index, uint32_t chunk_symbols, void *d_ws, size_t ws_bytes, void *stream) {
    return dev_decode(m, d_payload, nbits, nullptr, d_out, n_symbols, d_index, chunk_symbols, d_ws, ws_bytes, stream);
}

int mh_dev_decode_dn(const mh_model *m, const uint8_t *d_payload, const uint64_t *d_nbits, uint64_t nbits_hint, uint8_t *d_out,
                     uint64_t n_symbols, const uint64_t *d_index, uint32_t chunk_symbols, void *d_ws, size_t ws_bytes, void *stream) {
    if (!d_nbits) return MH_ERR_ARG;
    return dev_decode(m, d_payload, nbits_hint, d_nbits, d_out, n_symbols, d_index, chunk_symbols, d_ws, ws_bytes, stream);
}

// the model's and the stream's part of an index-builder / stream-decoder launch
static int idx_params(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, uint8_t prev0, uint32_t chunk_symbols, uint64_t *d_n_symbols,
                      mhk::IdxParams &p) {
    int shift = chunk_shift_of(chunk_symbols);
    if (shift < 0 || !aligned16(d_payload)) return MH_ERR_ARG;
    if (m->max_len > mh::MAX_CODE_BITS) return MH_ERR_CODE_TOO_LONG;
    if (!m->d_prim) return MH_ERR_NO_DEVICE;
    p = mhk::IdxParams{};
    p.payload = d_payload; p.payload_bytes = (nbits + 7) / 8; p.nbits = nbits;
    p.order = m->type == 2 ? 2 : 1;
    p.prev0 = ctx_of_prev0(m, prev0); p.chunk_shift = uint32_t(shift);
    p.n_symbols = reinterpret_cast<unsigned long long *>(d_n_symbols);
    p.prim = m->d_prim; p.sec = m->d_sec; p.sec_base = m->d_sec_base; p.tree = m->d_tree;
    p.P = uint32_t(m->dec_bits);
    p.direct = m->dec_direct ? 1u : 0u; p.H = uint32_t(m->dec_h);
    p.len_gcd = m->len_gcd;
    p.max_len = uint32_t(m->max_len > 0 ? m->max_len : 1);
    if (m->type != 2 && m->tile_p) {                         // the tile decoder's tables: the index builder's fast path
        p.tprim = m->d_tprim; p.tsec = m->d_tsec; p.tP = uint32_t(m->tile_p); p.tH = uint32_t(m->tile_h); p.tnsec = m->tile_nsec;
    }
    return MH_OK;
}

static int dev_build_index(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, uint8_t prev0, uint64_t *d_index,
                           uint64_t index_cap, uint32_t chunk_symbols, uint64_t *d_n_symbols, void *d_ws, size_t ws_bytes, void *stream,
                           uint32_t *d_fine, uint64_t fine_cap) {
    if (!m || !d_index || !d_n_symbols || !d_ws || (!d_payload && nbits)) return MH_ERR_ARG;
    if (ws_bytes < mhk::build_index_workspace_bytes(nbits)) return MH_ERR_CAPACITY;
    mhk::IdxParams p;
    const int rc = idx_params(m, d_payload, nbits, prev0, chunk_symbols, d_n_symbols, p);
    if (rc != MH_OK) return rc;
    p.index = reinterpret_cast<unsigned long long *>(d_index); p.index_cap = index_cap;
    p.fine = m->type == 2 ? nullptr : d_fine;
    p.fine_cap = fine_cap;
    HIP_TRY(mhk::launch_build_index(p, d_ws, static_cast<hipStream_t>(stream)));
    return MH_OK;
}

// Streams without an index in two passes over the payload (mh.h): states and counts, then the segment decoder.
int mh_dev_decode_stream_states(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, uint8_t prev0, uint64_t *d_n_symbols,
                                void *d_ws, size_t ws_bytes, void *stream) {
    if (!m || !d_n_symbols || !d_ws || (!d_payload && nbits)) return MH_ERR_ARG;
    if (ws_bytes < mhk::build_index_workspace_bytes(nbits)) return MH_ERR_CAPACITY;
    mhk::IdxParams p;
    const int rc = idx_params(m, d_payload, nbits, prev0, MH_CHUNK_DEFAULT, d_n_symbols, p);
    if (rc != MH_OK) return rc;
    HIP_TRY(mhk::launch_stream_states(p, d_ws, static_cast<hipStream_t>(stream)));
    return MH_OK;
}

int mh_dev_decode_stream_emit(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, uint8_t prev0, uint8_t *d_out, uint64_t out_cap,
                              void *d_ws, size_t ws_bytes, void *stream) {
    if (!m || !d_ws || (!d_payload && nbits) || (!d_out && out_cap)) return MH_ERR_ARG;
    if (ws_bytes < mhk::build_index_workspace_bytes(nbits)) return MH_ERR_CAPACITY;
    if (nbits == 0) return MH_OK;
    if (mh_dev_index_path(d_ws, stream) != mhk::IDX_PATH_STATES) return MH_ERR_ARG;      // (no states in this workspace)
    mhk::IdxParams p;
    const int rc = idx_params(m, d_payload, nbits, prev0, MH_CHUNK_DEFAULT, nullptr, p);
    if (rc != MH_OK) return rc;
    HIP_TRY(mhk::launch_stream_emit(p, d_ws, d_out, out_cap, static_cast<hipStream_t>(stream)));
    return MH_OK;
}

int mh_dev_build_index(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, uint8_t prev0, uint64_t *d_index,
                       uint64_t index_cap, uint32_t chunk_symbols, uint64_t *d_n_symbols, void *d_ws, size_t ws_bytes, void *stream) {
    return dev_build_index(m, d_payload, nbits, prev0, d_index, index_cap, chunk_symbols, d_n_symbols, d_ws, ws_bytes, stream, nullptr, 0);
}

int mh_dev_build_index_fine(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, uint8_t prev0, uint64_t *d_index,
                            uint64_t index_cap, uint32_t chunk_symbols, uint32_t *d_fine, uint64_t fine_cap, uint64_t *d_n_symbols,
                            void *d_ws, size_t ws_bytes, void *stream) {
    return dev_build_index(m, d_payload, nbits, prev0, d_index, index_cap, chunk_symbols, d_n_symbols, d_ws, ws_bytes, stream, d_fine, fine_cap);
}

int mh_dev_index_path(const void *d_ws, void *stream);
int mh_dev_encode_path(const void *d_ws, void *stream) { return mh_dev_index_path(d_ws, stream); }   // same word of the status block

int mh_dev_decode_path(const void *d_ws, void *stream) {
    if (!d_ws) return MH_ERR_ARG;
    uint32_t v = 0;
    HIP_TRY(hipMemcpyAsync(&v, static_cast<const unsigned char *>(d_ws) + 40, sizeof v, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return int(v);
}

int mh_dev_decode_variant(const void *d_ws, void *stream) {
    if (!d_ws) return MH_ERR_ARG;
    uint32_t v = 0;
    HIP_TRY(hipMemcpyAsync(&v, static_cast<const unsigned char *>(d_ws) + 44, sizeof v, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return int(v) - 1;                                           // (the launcher stores variant + 1: 0 = the chunk decoder did not run)
}

int mh_dev_index_path(const void *d_ws, void *stream) {
    if (!d_ws) return MH_ERR_ARG;
    uint32_t v = 0;
    HIP_TRY(hipMemcpyAsync(&v, static_cast<const unsigned char *>(d_ws) + 8, sizeof v, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return int(v);
}

int mh_dev_status(const void *d_ws, void *stream) {
    if (!d_ws) return MH_ERR_ARG;
    int s = 0;
    const double t0 = g_phase && g_phase->on ? PhaseClock::now() : 0;
    HIP_TRY(hipMemcpyAsync(&s, d_ws, sizeof s, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (g_phase && g_phase->on) g_phase->device += PhaseClock::now() - t0;
    return status_from_device(s);
}

/* ------------------------------------------------------- host-buffer calls */


}  // extern "C"
