// coding.cpp — implementation of the coding.h face over the C ABI (include/mh.h).
#include "coding.h"

#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <sstream>
#include <thread>

// MH_TIMING=1: one stderr line per stage with the time spent inside it (the first call that touches the
// device also pays the HIP runtime's start-up, reported separately by tools/cli_rate.py as wall - stages).
namespace {
struct StageTimer {
    const char* what; size_t bytes;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    StageTimer(const char* w, size_t b) : what(w), bytes(b) {}
    ~StageTimer() {
        if (!getenv("MH_TIMING")) return;
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        eprintf("[mh-timing] %s %zu bytes %.4f s %.2f GB/s\n", what, bytes, s, bytes / s / 1e9);
    }
};
}  // namespace

// ------------------------------------------------------------------------------------- utils

void mh_or_die(int status, const char* what) {
    if (status == MH_OK) return;
    // the two messages the reference prints on these paths (src/coding.cpp:104,108); others are ours
    if (status == MH_ERR_CORRUPT) eprintf("Error while decoding file: Input appears corrupt.\n");
    else if (status == MH_ERR_TYPE) eprintf("Error: File encoding method does not match provided encoding table.\n");
    else eprintf("Error in %s: %s.\n", what, mh_strerror(status));
    exit(1);
}

int read_buffer(void* ptr, size_t size, size_t count, FILE* stream) {        // src/utils.cpp:58-67
    size_t r = fread(ptr, size, count, stream);
    if (r != count && ferror(stream)) {
        eprintf("Error occurred while reading file.\n");
        exit(1);
    }
    return (int)r;
}

int write_buffer(void* ptr, size_t size, size_t count, FILE* stream) {       // src/utils.cpp:69-78
    size_t r = fwrite(ptr, size, count, stream);
    if (r != count && ferror(stream)) {
        eprintf("Error occurred while writing file.\n");
        exit(1);
    }
    return (int)r;
}

void check_access(const char* path, bool write) {                            // src/utils.cpp:44-56 (prints only)
    if (access(path, write ? W_OK : R_OK) == -1)
        eprintf("Error: Unable to open \"%s\" for %s; %s.", path, write ? "writing" : "reading", strerror(errno));
}

// Printable form of a byte for the -g dump; output format of src/utils.cpp:18-42.
std::string charv(unsigned char c) {
    static const struct { unsigned char c; const char* s; } named[] = {
        {' ', "\\\\sp"}, {'\t', "\\\\t"}, {'\r', "\\\\r"}, {'\n', "\\\\n"}, {'"', "\\\""}, {'\'', "\\'"}, {'\\', "\\\\"}};
    for (const auto& e : named)
        if (e.c == c) return e.s;
    if (c > 32 && c < 127) return std::string(1, (char)c);
    std::ostringstream s;
    s << "\\\\" << std::hex << (int)c;
    return s.str();
}

void bitbuffer::fwrite_checked(const unsigned char* p, size_t n) {
    if (n) write_buffer(const_cast<unsigned char*>(p), 1, n, file_);
}

void bitbuffer::fill() {
    if (loaded_) return;
    loaded_ = true;
    unsigned char tmp[32768];
    size_t r;
    while ((r = (size_t)read_buffer(tmp, 1, sizeof tmp, file_)) > 0) rbuf_.insert(rbuf_.end(), tmp, tmp + r);
}

static std::vector<unsigned char> slurp(FILE* f) {
    std::vector<unsigned char> all;
    unsigned char tmp[1 << 16];
    size_t r;
    while ((r = (size_t)read_buffer(tmp, 1, sizeof tmp, f)) > 0) all.insert(all.end(), tmp, tmp + r);
    return all;
}

// Whole input from the stream's current position to EOF.  A regular file is mapped, not copied: the
// library stages it through the card segment by segment, so a file larger than host memory still
// works and nothing is read twice into process memory.  Pipes are read into a vector.
struct InputView {
    const unsigned char* data = nullptr;
    size_t size = 0;
    void* map = nullptr;
    size_t map_len = 0;
    std::vector<unsigned char> own;
    explicit InputView(FILE* f) {
        if (!f) return;                                              // empty view (placeholder)
        struct stat st;
        long pos = ftell(f);
        if (pos >= 0 && fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && (size_t)st.st_size > (size_t)pos) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
            if (m != MAP_FAILED) {
                map = m; map_len = (size_t)st.st_size;
                (void)madvise(m, map_len, MADV_SEQUENTIAL);
                populate_read();
                data = (const unsigned char*)m + pos;
                size = map_len - (size_t)pos;
                fseek(f, 0, SEEK_END);
                return;
            }
        }
        own = slurp(f);
        data = own.data();
        size = own.size();
    }
    // The mapping's pages are in the page cache but not in this process's page table: the first pass over it (the
    // library's copy into its pinned upload ring) would take a minor fault per 4 KiB — a million of them for 4 GiB.  A few
    // helper threads ask the kernel to map them in bulk (MADV_POPULATE_READ) while the process is still starting the device.
    std::vector<std::thread> populate;
    void populate_read() {
#ifdef MADV_POPULATE_READ
        if (!map || map_len < (size_t(64) << 20) || getenv("MH_NO_POPULATE_READ")) return;
        const size_t threads = 4, step = size_t(64) << 20;
        const size_t part = ((map_len / threads) + step - 1) / step * step;
        for (size_t t = 0; t < threads; ++t) {
            const size_t lo = t * part, hi = std::min(map_len, lo + part);
            if (lo >= hi) break;
            unsigned char* base = (unsigned char*)map;
            populate.emplace_back([base, lo, hi, step] {
                for (size_t o = lo; o < hi; o += step)
                    if (madvise(base + o, std::min(step, hi - o), MADV_POPULATE_READ) != 0) return;   // old kernel: plain faults do it
            });
        }
#endif
    }
    ~InputView() { for (std::thread& t : populate) t.join(); if (map) munmap(map, map_len); }
    InputView(const InputView&) = delete;
    InputView& operator=(const InputView&) = delete;
};

// Output of a known size.  A regular file is grown to that size and mapped, so the library writes the
// result where it belongs; anything else (stdout, a pipe) gets a buffer that is written afterwards.
struct OutputView {
    unsigned char* data = nullptr;
    size_t size = 0;
    void* map = nullptr;
    FILE* file = nullptr;
    std::vector<unsigned char> own;
    // The mapping of a freshly grown file has no pages yet.  A few helper threads ask the kernel to
    // allocate them in bulk (MADV_POPULATE_WRITE) while the device is still computing, so that the
    // device-to-host copies find their destination resident instead of faulting it in page by page.
    std::vector<std::thread> populate;
    void start_populate() {
#ifdef MADV_POPULATE_WRITE
        if (!map || size < (size_t(64) << 20)) return;
        size_t threads = 4;                                   // MH_POPULATE_THREADS: measured 4 / 8 / 16 in profiles/r03/README.md
        if (const char* e = getenv("MH_POPULATE_THREADS")) { const long v = atol(e); if (v >= 1 && v <= 64) threads = size_t(v); }
        const size_t step = size_t(32) << 20;
        const size_t part = ((size / threads) + step - 1) / step * step;
        for (size_t t = 0; t < threads; ++t) {
            const size_t lo = t * part, hi = std::min(size, lo + part);
            if (lo >= hi) break;
            unsigned char* base = data;
            populate.emplace_back([base, lo, hi, step] {
                for (size_t o = lo; o < hi; o += step)
                    if (madvise(base + o, std::min(step, hi - o), MADV_POPULATE_WRITE) != 0) return;   // old kernel: plain faults do it
            });
        }
#endif
    }
    void join_populate() { for (std::thread& t : populate) t.join(); populate.clear(); }
    void open(FILE* f, size_t n) {
        file = f; size = n;
        struct stat st;
        fflush(f);
        if (n && fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && ftell(f) == 0 && ftruncate(fileno(f), (off_t)n) == 0) {
            void* m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fileno(f), 0);
            if (m != MAP_FAILED) {
                map = m; data = (unsigned char*)m;
                start_populate();
                // MH_PREFAULT_WAIT=1 (tools/cli_rate.py): wait for the pages here, under a stage line of its own, so that the
                // library call that follows shows the pipeline's rate and this line the file system's
                if (getenv("MH_PREFAULT_WAIT")) { StageTimer t("prefault", n); join_populate(); }
                return;
            }
        }
        own.assign(n ? n : 1, 0);
        data = own.data();
    }
    void finish() {
        join_populate();
        if (map) {
            if (munmap(map, size) != 0) { eprintf("Error occurred while writing file.\n"); exit(1); }
            map = nullptr;
            fseek(file, 0, SEEK_END);
        } else if (size) {
            write_buffer(data, 1, size, file);
        }
    }
};

// ------------------------------------------------------------------------ histogram (a1/a2)

// The histogram pass and the encode pass read the same file (src/main.cpp:173-183, then 204-212).  The
// mapping made for the first is kept for the second, so that the library sees the same buffer twice and
// its input-residency option (include/mh.h) can serve the second pass from HBM: the file then crosses PCIe once.
namespace {
struct KeptInput {
    InputView* view = nullptr;
    dev_t dev = 0; ino_t ino = 0; off_t size = 0;
    void release() { delete view; view = nullptr; }
} g_kept;
bool same_file(FILE* f, const KeptInput& k) {
    struct stat st;
    return k.view && fstat(fileno(f), &st) == 0 && st.st_dev == k.dev && st.st_ino == k.ino && st.st_size == k.size && ftell(f) == 0;
}
}  // namespace

void construct_table(FILE* input_fd, int order, uint64_t* counts64) {
    g_kept.release();
    struct stat st;
    const bool regular = fstat(fileno(input_fd), &st) == 0 && S_ISREG(st.st_mode) && ftell(input_fd) == 0;
    InputView* viewp = new InputView(input_fd);
    InputView& all = *viewp;
    if (regular && all.map) { g_kept.view = viewp; g_kept.dev = st.st_dev; g_kept.ino = st.st_ino; g_kept.size = st.st_size; (void)mh_set_input_residency(1); }
    struct Cleanup { InputView* v; ~Cleanup() { if (g_kept.view != v) delete v; } } cleanup{viewp};
    StageTimer timer("histogram", all.size);
    if (order == 2) mh_or_die(mh_histogram_o2(all.data, all.size, counts64), "histogram");
    else if (order) mh_or_die(mh_histogram_o1(all.data, all.size, MH_PREV0, counts64), "histogram");
    else mh_or_die(mh_histogram_o0(all.data, all.size, counts64), "histogram");
}

// ------------------------------------------------------------------------ provider base

i_coding_provider::~i_coding_provider() { mh_model_free(model_); }

void i_coding_provider::build_from_counts(const uint64_t* counts, int order) {
    counts_.assign(counts, counts + (order == 2 ? (size_t(1) << 24) : order ? 65536 : 256));     // kept: compress() sizes its output from them
    mh_model* m = nullptr;
    mh_or_die(mh_model_from_counts(counts, order, &m), "table build");
    adopt(m);
}

void i_coding_provider::build_from_buffer(bitbuffer& buffer, int expected_type) {
    std::vector<unsigned char> bytes = buffer.rest();
    mh_model* m = nullptr;
    int rc = mh_model_from_table_bits(bytes.data(), bytes.size(), &m);
    if (rc == MH_OK && mh_model_type(m) != expected_type) { mh_model_free(m); rc = MH_ERR_TYPE; }
    mh_or_die(rc, "encoding table load");
    adopt(m);
}

void i_coding_provider::write_coding_tree(bitbuffer& buffer) {
    size_t n = 0;
    mh_or_die(mh_model_write_table(model_, nullptr, 0, &n), "table size");
    std::vector<unsigned char> bytes(n ? n : 1);
    mh_or_die(mh_model_write_table(model_, bytes.data(), n, &n), "table write");
    buffer.push_bytes(bytes.data(), n);
}

encoding_descriptor& i_coding_provider::get_encoding(unsigned char prev, unsigned char c) {
    int len = 0;
    uint64_t code = 0;
    mh_model_get_code(model_, prev, c, &len, &code);
    scratch_desc_.length = len;
    scratch_desc_.encoding.assign((size_t)(len + 7) / 8, 0);
    for (int i = 0; i < len && i < 64; ++i)
        if ((code >> (len - 1 - i)) & 1) scratch_desc_.encoding[(size_t)i / 8] |= (unsigned char)(1u << (7 - i % 8));
    return scratch_desc_;
}

const tree_node* i_coding_provider::decoding_lookup(unsigned char prev, unsigned char w) {
    int present = 0, inner = 0, value = 0, depth = 0;
    mh_model_get_lut(model_, prev, w, &present, &inner, &value, &depth);
    if (!present) return nullptr;
    scratch_node_.is_internal = inner != 0;
    scratch_node_.value = (unsigned char)value;
    scratch_node_.depth = depth;
    return &scratch_node_;
}

bool i_coding_provider::context_empty(int prev) {
    int present = 0, a, b, c;
    mh_model_get_lut(model_, prev, 0, &present, &a, &b, &c);
    return !present;
}

static bool valid_chunk(uint64_t c) { return c >= MH_CHUNK_MIN && c <= MH_CHUNK_MAX && (c & (c - 1)) == 0; }

// src/coding.cpp:61-94: header placeholder, payload, header rewrite.  Here the payload comes out of the
// HIP encoder in one piece, so the header is known before anything is written and no seek is needed.
void i_coding_provider::compress(FILE* input_fd, FILE* output_fd) {
    // the mapping of the histogram pass, when this is the same file read from its start again
    InputView* kept = same_file(input_fd, g_kept) ? g_kept.view : nullptr;
    if (kept) fseek(input_fd, 0, SEEK_END);
    InputView fresh_or_empty(kept ? nullptr : input_fd);
    const InputView& in = kept ? *kept : fresh_or_empty;
    struct Release { ~Release() { g_kept.release(); (void)mh_set_input_residency(0); } } release_kept;
    StageTimer timer("compress", in.size);
    // The file size is known before anything is encoded: histogram of the input . code lengths.  (The
    // histogram that built the tables is reused when there is one; with -e it is taken here.)
    const int order = mh_model_type(model_);
    if (counts_.empty()) {
        (void)mh_set_input_residency(1);                             // this histogram's upload serves the encode below
        counts_.assign(order == 2 ? (size_t(1) << 24) : order ? 65536 : 256, 0);
        if (order == 2) mh_or_die(mh_histogram_o2(in.data, in.size, counts_.data()), "histogram");
        else if (order) mh_or_die(mh_histogram_o1(in.data, in.size, MH_PREV0, counts_.data()), "histogram");
        else mh_or_die(mh_histogram_o0(in.data, in.size, counts_.data()), "histogram");
    }
    uint64_t bits = 0;
    mh_or_die(mh_model_payload_bits(model_, counts_.data(), &bits), "compress");
    const size_t nbytes = (size_t)((bits + 7) / 8);
    OutputView out;
    out.open(output_fd, 1 + nbytes);
    out.data[0] = mh_stream_header(model_, bits);               // src/coding.cpp:88 — known up front, no seek back
    uint64_t nbits = 0;
    std::vector<uint64_t> index;
    if (!index_path_.empty() && !valid_chunk(chunk_)) mh_or_die(MH_ERR_ARG, "compress (--chunk)");
    if (!index_path_.empty()) index.resize((size_t)mh_index_entries(in.size, chunk_) + 1);
    mh_or_die(mh_encode(model_, in.data, in.size, MH_PREV0, out.data + 1, nbytes, &nbits,
                        index.empty() ? nullptr : index.data(), chunk_), "compress");
    if (nbits != bits) mh_or_die(MH_ERR_CORRUPT, "compress");
    out.finish();
    if (!index_path_.empty()) {
        // sidecar: magic, chunk size, symbol count, entries (little-endian u64s)
        FILE* f = fopen(index_path_.c_str(), "wb");
        if (!f) { eprintf("Error while opening index output; %s.\n", strerror(errno)); exit(1); }
        uint64_t head[3] = {0x315844494D48ull /* "HMIDX1" */, chunk_, (uint64_t)in.size};
        write_buffer(head, 8, 3, f);
        size_t ne = (size_t)mh_index_entries(in.size, chunk_);
        if (ne) write_buffer(index.data(), 8, ne, f);
        fclose(f);
    }
    fclose(input_fd);                                   // src/coding.cpp:93
    if (output_fd != stdout) fclose(output_fd);         // src/bitbuffer.h:35-40
    else fflush(output_fd);
}

static uint8_t* open_output_cb(void* ctx, size_t n) {
    OutputView* o = (OutputView*)ctx;
    o->open(o->file, n);
    return o->data;
}

// src/coding.cpp:96-160
void i_coding_provider::decompress(FILE* input_fd, FILE* output_fd) {
    InputView in(input_fd);
    if (in.size == 0) mh_or_die(MH_ERR_CORRUPT, "decompress");
    uint64_t nbits = 0;
    mh_or_die(mh_stream_parse_header(model_, in.data[0], in.size, &nbits), "decompress");
    std::vector<uint64_t> index;
    uint64_t n_symbols = 0;
    uint32_t chunk = 0;
    if (!index_path_.empty()) {
        FILE* f = fopen(index_path_.c_str(), "rb");
        if (f) {
            uint64_t head[3];
            // The sidecar is untrusted input: a chunk size that is not a power of two in [MH_CHUNK_MIN,
            // MH_CHUNK_MAX] or a symbol count above the payload's bit count (every code has >= 1 bit) cannot
            // come from this encoder; such a file is ignored and the stream is decoded without an index.
            if (fread(head, 8, 3, f) == 3 && head[0] == 0x315844494D48ull && valid_chunk(head[1]) && head[2] <= nbits) {
                chunk = (uint32_t)head[1];
                n_symbols = head[2];
                size_t ne = (size_t)mh_index_entries(n_symbols, chunk);
                index.resize(ne + 1);
                if (fread(index.data(), 8, ne, f) != ne) index.clear();
            } else {
                eprintf("Warning: index sidecar %s is not usable; extracting without it.\n", index_path_.c_str());
            }
            fclose(f);
        }
    }
    const bool have_index = !index.empty();
    // the output size is known once the symbols are counted (without an index: after the device has
    // rebuilt it); the library then asks for the buffer, which is the mapped output file
    size_t n = 0;
    OutputView out;
    out.file = output_fd;
    StageTimer timer("decompress", in.size);
    mh_or_die(mh_decode_to(model_, in.data + 1, nbits, MH_PREV0, open_output_cb, &out, &n,
                           have_index ? index.data() : nullptr, have_index ? chunk : 0, have_index ? n_symbols : 0), "decompress");
    if (out.data) out.finish();
    fclose(input_fd);
    if (output_fd != stdout) fclose(output_fd);
    else fflush(output_fd);
}

// ------------------------------------------------------------------------ -g dumps (N3)

void encoding_descriptor::print() {                                          // src/coding.cpp:29-33
    for (int i = 0; i < length; i++) printf("%d", (encoding[(size_t)i / 8] >> (8 - i % 8 - 1)) & 1);
}

void i_coding_provider::print_table_for(int prev) {                           // src/huffman.cpp:52-62
    printf("Table:\n");
    for (int i = 0; i < 256; i++) {
        encoding_descriptor& e = get_encoding((unsigned char)prev, (unsigned char)i);
        if (e.length) {
            printf("%s %d ", charv((unsigned char)i).c_str(), e.length);
            e.print();
            printf("\n");
        }
    }
}

// Graphviz dump in the reference's format (src/tree.cpp:7-53).  The tree is re-read from the model's
// own pre-order serialisation (src/huffman.cpp:174-188), so no tree structure crosses the C ABI.
namespace {
struct dot_reader {
    const std::vector<unsigned char>& b;
    size_t pos;
    int bit() { int v = (b[pos >> 3] >> (7 - (pos & 7))) & 1; ++pos; return v; }
    int byte() { int v = 0; for (int i = 0; i < 8; ++i) v = (v << 1) | bit(); return v; }
};
// prints the subtree that starts at the reader's cursor; n = this node's number; returns last number used
int dot_nodes(dot_reader& r, int n) {
    printf("\tn%d;\n", n);
    if (r.bit()) {
        printf("\tn%d [label=\"%s\"];\n", n, charv((unsigned char)r.byte()).c_str());
        return n;
    }
    printf("\tn%d [label=\"\"];\n", n);
    int next = n + 1;
    int last = dot_nodes(r, next);
    printf("\tn%d -- n%d;\n", n, next);
    next = last + 1;
    last = dot_nodes(r, next);
    printf("\tn%d -- n%d;\n", n, next);
    return last;
}
}  // namespace

int i_coding_provider::print_tree_for(int prev, bool subgraph, int n, const std::string& label) {
    size_t nb = 0;
    mh_or_die(mh_model_write_table(model_, nullptr, 0, &nb), "table size");
    std::vector<unsigned char> bytes(nb + 1);
    mh_or_die(mh_model_write_table(model_, bytes.data(), nb, &nb), "table write");
    dot_reader r{bytes, 0};
    if (get_type() == 1) {
        // skip to context `prev`: marker bit, then per context a presence bit (+ tree)
        r.bit();
        for (int p = 0; p < prev; ++p) {
            if (!r.bit()) continue;
            int open = 1;                      // subtrees still to read
            while (open) { if (r.bit()) { r.byte(); --open; } else { ++open; } }
        }
        r.bit();
    }
    if (subgraph) {
        printf("subgraph clusterG%d {\n", n);
        printf("\tlabel=\"%s\";\n", label.c_str());
        printf("\tcolor=invis;\n");
    } else {
        printf("graph G {\n");
    }
    printf("\tnodesep=0.3;\n");
    printf("\tranksep=0.2;\n");
    printf("\tnode [shape=circle, fixedsize=true];\n");
    printf("\tedge [arrowsize=0.8];\n");
    n = dot_nodes(r, n) + 1;
    printf("}\n");
    return n;
}

// ------------------------------------------------------------------------ concrete providers

static std::vector<uint64_t> widen_counts(const int* counts, size_t n) {
    std::vector<uint64_t> w(n);
    for (size_t i = 0; i < n; ++i) w[i] = (uint64_t)(counts[i] < 0 ? 0 : counts[i]);
    return w;
}

huffman_table::huffman_table(int* counts) { build_from_counts(widen_counts(counts, 256).data(), 0); }
huffman_table::huffman_table(const uint64_t* counts) { build_from_counts(counts, 0); }
huffman_table::huffman_table(bitbuffer& buffer) { build_from_buffer(buffer, 0); }
bool huffman_table::empty() { return context_empty(0); }
void huffman_table::print_table() { print_table_for(0); }
void huffman_table::print_tree() { print_tree_for(0, false, 0, ""); }

markov_huffman_table::markov_huffman_table(int* counts) { build_from_counts(widen_counts(counts, 65536).data(), 1); }
markov_huffman_table::markov_huffman_table(const uint64_t* counts) { build_from_counts(counts, 1); }
markov_huffman_table::markov_huffman_table(bitbuffer& buffer) { build_from_buffer(buffer, 1); }

// order-2 extension (parity unpinned; include/mh.h): 65536 two-byte contexts
markov2_huffman_table::markov2_huffman_table(const uint64_t* counts) { build_from_counts(counts, 2); }
markov2_huffman_table::markov2_huffman_table(bitbuffer& buffer) { build_from_buffer(buffer, 2); }
void markov2_huffman_table::print_table() { eprintf("Error: -g is not available for order-2 tables.\n"); exit(1); }
void markov2_huffman_table::print_tree() { print_table(); }

void markov_huffman_table::print_table() {                                    // src/markov_huffman.cpp:31-38
    for (int i = 0; i < 256; i++) {
        if (!context_empty(i)) {
            printf("Prev '%s' table:\n", charv((unsigned char)i).c_str());
            print_table_for(i);
        }
    }
}

void markov_huffman_table::print_tree() {                                     // src/markov_huffman.cpp:40-50
    printf("graph G {\n");
    printf("\tpackmode=\"cluster\";\n");
    for (int i = 0, n = 0; i < 256; i++) {
        if (!context_empty(i)) {
            printf("/* Prev '%s' tree: */\n", charv((unsigned char)i).c_str());
            n = print_tree_for(i, true, n, "Prev: " + charv((unsigned char)i));
        }
    }
    printf("}\n");
}
