/*
 * mh.h — C ABI of the MI355X-native Markov-Huffman codec (libmhc.so).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (jeremy-rifkin/Markov-Huffman-Coding) has no FFI of its own: its seam is the
 * C++ class i_coding_provider (src/coding.h:18-35) built in main()
 * (src/main.cpp:136-184) from a histogram made by construct_table()
 * (src/main.cpp:29-39).  Each entry point below names the reference interface
 * it replaces; INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns an int status
 *     (MH_OK == 0, negative = error).  The reference's convention is
 *     eprintf + exit(1) (src/utils.cpp:62-65, src/coding.cpp:103-110); the
 *     C++ face in markov-huffman-coding_amd/host/ turns a non-zero status into
 *     exactly that.
 *   - the caller owns every buffer.  "mh_*" functions take HOST pointers and
 *     stage through HBM internally; "mh_dev_*" functions take DEVICE pointers
 *     plus a hipStream_t (as void*) and neither allocate nor synchronise,
 *     with the exceptions stated at their declarations: the model builders
 *     (mh_dev_model_from_counts allocates and synchronises once,
 *     mh_dev_model_from_counts_ws only synchronises once), mh_dev_build_index
 *     and mh_dev_status.
 *   - all compute runs in hand-written HIP kernels for gfx950.  There is no
 *     CPU fallback: without a usable GPU every compute call returns
 *     MH_ERR_NO_DEVICE.
 *   - an mh_model is immutable after construction and may be shared by threads.
 *   - bit order everywhere: MSB first inside a byte (src/bitbuffer.cpp:12).
 */
#ifndef MH_H
#define MH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_VERSION 100

enum {
    MH_OK = 0,
    MH_ERR_ARG = -1,           /* null/misaligned pointer, bad size or option            */
    MH_ERR_NO_DEVICE = -2,     /* no usable gfx950 device / HIP runtime                  */
    MH_ERR_HIP = -3,           /* a HIP call failed (mh_last_hip_error() has the code)   */
    MH_ERR_CORRUPT = -4,       /* stream not decodable (src/coding.cpp:103-106)          */
    MH_ERR_TYPE = -5,          /* stream/table type mismatch (src/coding.cpp:107-110)    */
    MH_ERR_BADTABLE = -6,      /* table file not parseable (src/huffman.cpp:166-172)     */
    MH_ERR_CODE_TOO_LONG = -7, /* a codeword exceeds 64 bits (not reachable < 2^44 B)    */
    MH_ERR_CAPACITY = -8,      /* output buffer or workspace too small                   */
    MH_ERR_TIMEOUT = -9,       /* bounded device-side wait expired (should not happen)   */
    MH_ERR_NOMEM = -10,
};

/* Initial context of every stream: the space character (src/main.cpp:32, src/coding.cpp:67,118). */
#define MH_PREV0 0x20

const char *mh_strerror(int status);
int mh_last_hip_error(void);
/* Diagnostic: how the calling thread's last mh_decode / mh_decode_to WITHOUT an index rebuilt it (codes of
 * mh_dev_index_path below). */
int mh_last_index_path(void);
/* Diagnostic: how many segments of the calling thread's last mh_encode* call the one-pass order-2 encoder gave up on
 * (MH_ERR_TIMEOUT: its bounded waits ran out, e.g. on a device shared with a long-running kernel) and that were encoded
 * again with the two-pass pair — the bytes are the same, the time is not, so the retry is counted, never silent.
 * 0 in normal operation.  mh_total_encode_retries: the same over all threads since the library was loaded. */
int mh_last_encode_retries(void);
uint64_t mh_total_encode_retries(void);
/* Number of usable devices; 0 when there is none (never an error). */
int mh_device_count(void);
/* Device used by the calling thread's subsequent mh_* / mh_dev_* calls (hipSetDevice). */
int mh_set_device(int ordinal);

/* Minimal device-memory helpers for hosts that have no HIP allocator of their own (the Python tests):
 * hipMalloc / hipFree / synchronous hipMemcpy. */
int mh_dev_malloc(void **d_ptr, size_t bytes);
int mh_dev_free(void *d_ptr);
int mh_dev_upload(void *d_dst, const void *h_src, size_t bytes);
int mh_dev_download(void *h_dst, const void *d_src, size_t bytes);

/* ------------------------------------------------------------------ model */

typedef struct mh_model mh_model;

/* Replaces huffman_table(int*) (src/huffman.h:14, src/huffman.cpp:18-20,131-164; order 0, 256 counts)
 * and markov_huffman_table(int*) (src/markov_huffman.h:12, src/markov_huffman.cpp:9-13; order 1,
 * counts[256*prev+sym], 65536 counts).  Tie-breaking reproduces min_pq (src/min_pq.tpp:4-52) and the
 * height swap (src/huffman.cpp:147-149) exactly; counts are 64-bit (reference: int).
 * order 2 (extension, described before the chunk-index section): 1 << 24 counts, needs a device.
 * Builds the code tables and decode LUTs (src/huffman.cpp:91-123) and uploads them to the current
 * device.  Host pointer in. */
int mh_model_from_counts(const uint64_t *counts, int order, mh_model **out);

/* Same, counts resident in HBM (e.g. straight out of mh_dev_histogram_o1 or an RCCL all-reduce).
 * Order 1: the per-context tree build (heap emulation), code derivation and table fill run in HIP
 * kernels on `stream` (mh_tree.hip); the counts never leave the device.  The call synchronises the
 * stream (16 KiB of table sizes come back so that the host can pick the decode-table layout); the host
 * copy of the trees that table files and the query calls below need is made lazily, on first use.
 * Order 0 (one tree) takes the host route. */
int mh_dev_model_from_counts(const uint64_t *d_counts, int order, void *stream, mh_model **out);
/* The same (order 1 only) with every device byte of the model placed in a caller workspace of at least
 * mh_dev_model_workspace(1) bytes (16-byte aligned): no allocation inside, and `stream` is synchronised
 * exactly once.  The model borrows the workspace: keep it alive, and do not rebuild into it, until the
 * model has been freed and the work that uses it has finished.  MH_ERR_CAPACITY when it is too small. */
size_t mh_dev_model_workspace(int order);
int mh_dev_model_from_counts_ws(const uint64_t *d_counts, int order, void *d_ws, size_t ws_bytes, void *stream, mh_model **out);

/* Replaces the table-file constructors huffman_table(bitbuffer&) / markov_huffman_table(bitbuffer&)
 * (src/huffman.cpp:22-25,166-172; src/markov_huffman.cpp:15-25) and main()'s type sniffing on the
 * first bit (src/main.cpp:147-161).  `bytes` = whole table file. */
int mh_model_from_table_bits(const uint8_t *bytes, size_t n, mh_model **out);

/* Replaces write_coding_tree (src/markov_huffman.cpp:80-88, src/huffman.cpp:83-85,174-188) plus the
 * bitbuffer flush that pads to a byte (src/bitbuffer.cpp:170-180).  *nbytes = size needed/written. */
int mh_model_write_table(const mh_model *m, uint8_t *out, size_t cap, size_t *nbytes);

/* get_type() (src/coding.h:29-32): 0 simple Huffman, 1 Markov-Huffman. */
int mh_model_type(const mh_model *m);
/* Longest codeword in bits (0 for an all-empty model). */
int mh_model_max_code_len(const mh_model *m);
/* The shortest code of any context (0: a model without codes).  A payload of nbits bits holds at most nbits / min symbols:
 * the bound for index and fine-index capacities of a stream whose symbol count is not known (mh_dev_build_index_fine). */
int mh_model_min_code_len(const mh_model *m);
/* get_encoding(prev, c) (src/markov_huffman.cpp:52-54 -> src/huffman.cpp:71-73): *len bits,
 * *code right-aligned (valid when *len <= 64).  *len == 0: symbol has no code in this context. */
int mh_model_get_code(const mh_model *m, int prev, int sym, int *len, uint64_t *code);
/* decoding_lookup(prev, w) (src/markov_huffman.cpp:56-58 -> src/huffman.cpp:87-89): the 8-bit-window
 * LUT entry.  *present == 0 for a null entry (empty context). */
int mh_model_get_lut(const mh_model *m, int prev, int w, int *present, int *is_internal, int *value, int *depth);
/* How the decode tables of this model are laid out on the device: *primary_bits = width P of the
 * first-level window (8 = the reference's own 8-bit LUT, src/huffman.cpp:97-123; narrower when that
 * is what makes both table levels fit LDS), *secondary_entries = second-level entries,
 * *in_lds = 1 when both levels are LDS-resident in the decode kernel. */
int mh_model_decode_layout(const mh_model *m, int *primary_bits, int *secondary_entries, int *in_lds);
/* The same for the tile decoder's tables (mh_dev_decode_fine; LSB-first indexed): *primary_bits = width of the
 * LDS-resident first level (0: this model has no tile tables, mh_dev_decode_fine then runs the chunk decoder),
 * *secondary_bits = height of the uniform second-level tables, *secondary_entries = their total entry count (L2). */
int mh_model_tile_layout(const mh_model *m, int *primary_bits, int *secondary_bits, int *secondary_entries);
/* Diagnostic: copies one of the model's device images to the host (tests compare the host-built and the
 * device-built tables bit for bit).  which: 0 enc16, 1 len8, 2 len_slot, 3 code64, 4 decode prim,
 * 5 decode sec, 6 sec_base, 7 walk tree, 8 / 9 the tile decoder's first- / second-level tables (LSB-first
 * indexed, see mh_dev_decode_fine; empty when the model has none).  *bytes = image size; copied when cap suffices. */
int mh_model_image(const mh_model *m, int which, void *out, size_t cap, size_t *bytes);
void mh_model_free(mh_model *m);

/*
 * ORDER 2 — EXTENSION, PARITY UNPINNED.  order == 2 selects 65536 contexts, ctx = (byte before previous)
 * << 8 | previous byte (both ' ' before the stream): counts[ctx * 256 + sym], 1 << 24 entries.  The
 * reference implements order 1 only and merely speculates about higher orders (README.md:158-166), so
 * there is nothing to pin this against; the spec is the generalised oracle (oracle/mh_oracle.h): the same
 * per-context algorithm (src/huffman.cpp:131-164) per two-byte context.  Differences at the boundary:
 *   - mh_model_type() == 2; mh_model_get_code() takes the 16-bit context as `prev`; mh_model_get_lut() is
 *     not available; the model always lives on the device (there is no host-only order-2 model);
 *   - stream header 0x40 | unused bits (mh_stream_header / mh_stream_parse_header): the reference's magic
 *     is 0x30 (src/coding.cpp:103-106), so its decompress reports an order-2 stream as corrupt;
 *   - table file: the 33 bytes of an EMPTY order-1 table (which is what the reference's type sniffing,
 *     src/main.cpp:147-161, and loader make of it), the magic "MH2\x01", then per context bit 0 | bit 1 +
 *     tree (src/huffman.cpp:174-188), zero padded;
 *   - index entries carry the two context bytes in bits 48..63 (MH_INDEX2_BIT_MASK);
 *   - `prev0` arguments stand for BOTH context bytes.
 * All tables live in HBM and are served from L2 / the Infinity Cache (16.7 M codewords do not fit LDS).
 */
/* The order-2 model build in two steps, for ranks that share it: after a reduce-scatter of the 1 << 24
 * counts every rank holds the summed counts of its 65536 / G contexts, builds THEIR trees into a workspace of
 * mh_dev_model2_workspace() bytes (mh_dev_model2_build_slice; d_counts_slice = first count of context ctx_first), the
 * ranks all-gather the per-context arrays in place — mh_dev_model2_array(which, &offset, &bytes_per_context), which = 0..6:
 * a rank's share of array `which` is the byte range [offset + ctx_first * bytes_per_context, offset + ctx_end *
 * bytes_per_context) — and mh_dev_model2_finish derives every table from them (one stream synchronisation).  The model
 * borrows the workspace.  mh_dev_model_from_counts(order 2) is the same two steps over all contexts, in a ~600 MiB block of
 * its own; that block is kept when the model is freed and reused by the next order-2 build on the same device (a codec
 * that rebuilds its model per stream would otherwise allocate and free it every time). */
size_t mh_dev_model2_workspace(void);
int mh_dev_model2_array(int which, size_t *offset, size_t *bytes_per_context);
int mh_dev_model2_build_slice(const uint64_t *d_counts_slice, uint32_t ctx_first, uint32_t ctx_end,
                              void *d_ws, size_t ws_bytes, void *stream);
int mh_dev_model2_finish(void *d_ws, size_t ws_bytes, void *stream, mh_model **out);
int mh_histogram_o2(const uint8_t *data, size_t n, uint64_t *counts /* 1 << 24 */);
int mh_dev_histogram_o2(const uint8_t *d_data, size_t n, uint16_t ctx0, uint64_t *d_counts /* 1 << 24 */, void *stream);
/* [r4] The same with a workspace (256-byte aligned, mh_dev_histogram_o2_workspace(n) bytes: 2 bytes per input byte, at
 * most 4 GiB, + 34 to 97 MiB of tables): sources with millions of live (context, symbol) keys — Zipf or uniform bytes — overflow the kernel's LDS tag
 * cache and would count at the rate of 64-bit global atomics (36 ms per GiB); with the workspace such a slab of the input is
 * partitioned by the context's high byte and every bucket counted in LDS like an order-1 histogram.  The choice is made on
 * the device per 2 GiB slab, from the cache misses of the slab's first 4 MiB; mh_dev_index_path(d_ws) afterwards says what
 * was chosen (bit 0: a slab stayed in the tag cache, bit 1: a slab was partitioned).  Without a workspace (or below 32 MiB) everything goes through the tag cache. */
size_t mh_dev_histogram_o2_workspace(size_t n);
int mh_dev_histogram_o2_ws(const uint8_t *d_data, size_t n, uint16_t ctx0, uint64_t *d_counts /* 1 << 24 */,
                           void *d_ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------- chunk index */
/*
 * The reference's stream has no index (src/coding.cpp:35-59), and a decoder's state is
 * (bit position, previous byte), so parallel decode needs an out-of-band index: one uint64 per chunk
 * of `chunk_symbols` input bytes,
 *        entry = (context byte at the chunk start << 56) | payload bit offset of the chunk start.
 * mh_encode / mh_dev_encode produce it; it never changes the payload bytes.  chunk_symbols must be a
 * power of two in [MH_CHUNK_MIN, MH_CHUNK_MAX].
 */
#define MH_CHUNK_MIN 256u
#define MH_CHUNK_MAX 8192u
#define MH_CHUNK_DEFAULT 1024u
#define MH_INDEX_BIT_MASK 0x00FFFFFFFFFFFFFFull
/* order-2 models (extension below): entry = (two context bytes << 48) | bit offset */
#define MH_INDEX2_BIT_MASK 0x0000FFFFFFFFFFFFull
static inline uint64_t mh_index_entries(uint64_t n_symbols, uint32_t chunk_symbols) {
    return (n_symbols + chunk_symbols - 1) / chunk_symbols;
}

/* ------------------------------------------------------- host-buffer calls */

/* Replaces construct_table + the order-1 lambda (src/main.cpp:29-39,173-181): counts[256*prev+c]++,
 * prev starting at prev0.  counts: 65536 entries, overwritten. */
int mh_histogram_o1(const uint8_t *data, size_t n, uint8_t prev0, uint64_t *counts);
/* Replaces construct_table + the order-0 lambda (src/main.cpp:164-171).  counts: 256 entries. */
int mh_histogram_o0(const uint8_t *data, size_t n, uint64_t *counts);

/* Option for two-pass callers (histogram, then encode, of the SAME host buffer — what the CLI does,
 * src/main.cpp:173-183 + 204-212): when on, mh_histogram_o0/o1/o2 leave their upload of the input in HBM
 * (if it fits beside everything else) and the next mh_encode of that buffer — same pointer, size and
 * sampled content — reads it there, so the data crosses PCIe once.  The caller promises not to modify the
 * buffer between the two calls.  Off by default; turning it off frees what is held. */
int mh_set_input_residency(int on);

/* Replaces the body of i_coding_provider::compress (src/coding.cpp:61-94) between the header
 * placeholder and the header rewrite: payload bits only.  *nbits = payload length in bits; payload
 * bytes written = ceil(*nbits / 8), zero padded (src/bitbuffer.cpp:175).  cap must be >=
 * mh_encode_bound(n).  index/chunk_symbols optional (index == NULL: none). */
int mh_encode(const mh_model *m, const uint8_t *data, size_t n, uint8_t prev0,
              uint8_t *out_payload, size_t cap, uint64_t *nbits,
              uint64_t *index, uint32_t chunk_symbols);
/* Worst-case payload bytes for n input bytes under model m (n * max_code_len bits, rounded up, + slack). */
size_t mh_encode_bound(const mh_model *m, size_t n);
/* The header byte of src/coding.cpp:88: 0x30 | (~type & 1) << 3 | (8 - nbits % 8) % 8. */
uint8_t mh_stream_header(const mh_model *m, uint64_t nbits);
/* Validates a header byte as src/coding.cpp:100-116 does and returns the payload length in bits for a
 * file of file_bytes bytes: MH_ERR_CORRUPT on bad magic, MH_ERR_TYPE on a table/stream mismatch. */
int mh_stream_parse_header(const mh_model *m, uint8_t header, uint64_t file_bytes, uint64_t *nbits);

/* Replaces the loop of i_coding_provider::decompress (src/coding.cpp:118-157): decode exactly `nbits`
 * payload bits.  With an index (from mh_encode) chunks decode in parallel and n_symbols must be the
 * original length; with index == NULL (a stream produced by the reference) a device-side
 * index-building pass runs first and n_symbols is ignored.  *nbytes = decoded size (written if cap
 * suffices, else MH_ERR_CAPACITY with *nbytes set). */
int mh_decode(const mh_model *m, const uint8_t *payload, uint64_t nbits, uint8_t prev0,
              uint8_t *out, size_t cap, size_t *nbytes,
              const uint64_t *index, uint32_t chunk_symbols, uint64_t n_symbols);

/* Device footprint of the host-buffer calls: bounded by the segment size (256 MiB, MH_SEGMENT_BYTES) for
 * mh_histogram_*, mh_encode and mh_decode WITH an index.  mh_decode / mh_decode_to WITHOUT an index are the
 * exception: the whole payload is uploaded, an index of nbits / chunk_symbols + 2 entries is rebuilt beside it,
 * and the index builder's workspace takes about 28 bytes per 512 bytes of payload — roughly 1.1 x the payload
 * in total, plus one output segment. */
/* mh_decode for callers that cannot know the output size beforehand (a stream without an index):
 * get_out(ctx, n) is called exactly once, when the symbol count n is known, and returns where the n
 * bytes go (NULL -> MH_ERR_CAPACITY).  The CLI maps its output file there. */
typedef uint8_t *(*mh_output_fn)(void *ctx, size_t n);
int mh_decode_to(const mh_model *m, const uint8_t *payload, uint64_t nbits, uint8_t prev0,
                 mh_output_fn get_out, void *ctx, size_t *nbytes,
                 const uint64_t *index, uint32_t chunk_symbols, uint64_t n_symbols);

/* Payload bits this model produces for data with the given histogram (host counts: 65536 entries for a
 * Markov model, 256 for a Huffman model): the exact size of the compressed file before encoding. */
int mh_model_payload_bits(const mh_model *m, const uint64_t *counts, uint64_t *nbits);

/* ------------------------------------------------------------ device calls */
/* Device pointers, stream-ordered, no allocation, no synchronisation.  d_data / d_payload / d_out must
 * be 16-byte aligned.  Workspaces: query the size, allocate once, reuse. */

/* Optional workspace of mh_dev_histogram_o1 (pass NULL, 0 to do without): with it the workgroups'
 * counters leave as plain stores and are summed by a second kernel instead of 16.7 M device-scope
 * atomics, ~0.5 ms less per call.  A workspace of the full size (16-byte aligned) also keeps what
 * mh_dev_encode_hist needs: every workgroup counts one contiguous region of the input and leaves that
 * region's own pair counts behind. */
size_t mh_dev_histogram_workspace(size_t n);
/* d_counts (65536 or 256 x uint64) is overwritten.  With a workspace (>= 256 bytes) the order-1 call also checks
 * on the device that the counts add up to n (src/main.cpp:176-178: the reference's counts sum to the file size; a
 * spilled LDS counter field or a lost fix-up cannot hide): mh_dev_status(d_ws) then reports MH_ERR_CORRUPT. */
int mh_dev_histogram_o1(const uint8_t *d_data, size_t n, uint8_t prev0, uint64_t *d_counts,
                        void *d_ws, size_t ws_bytes, void *stream);
int mh_dev_histogram_o0(const uint8_t *d_data, size_t n, uint64_t *d_counts,
                        void *d_ws, size_t ws_bytes, void *stream);

/* (large enough for the encoder to take a region-mode histogram of the input by itself when the call comes without
 * one: mh_dev_encode, mh_dev_encode_at and mh_encode then run the same fast encoder as mh_dev_encode_hist) */
size_t mh_dev_encode_workspace(size_t n);
/* d_nbits: one uint64 (payload bits).  d_index: mh_index_entries(n, chunk_symbols) entries or NULL.
 * Errors found on the device (capacity overrun) go to the int32 at the start of d_ws: mh_dev_status(). */
int mh_dev_encode(const mh_model *m, const uint8_t *d_data, size_t n, uint8_t prev0,
                  uint8_t *d_payload, size_t cap, uint64_t *d_nbits,
                  uint64_t *d_index, uint32_t chunk_symbols,
                  void *d_ws, size_t ws_bytes, void *stream);

/* mh_dev_encode_at for the compress path, where a histogram of the SAME device buffer was taken just before
 * (src/main.cpp:173-183, then 204-212): d_hist_ws / hist_ws_bytes is the workspace mh_dev_histogram_o1
 * filled for (d_data, n, prev0), untouched since.  The encoder then prices each of the histogram's regions from
 * its pair counts and the code lengths and needs no pass of its own over the input to find where everything
 * goes: the input is read once.  Same payload, index and *d_nbits as mh_dev_encode_at.  Falls back to
 * mh_dev_encode_at by itself for order-2 models or a workspace that is too small (codes over 12 bits are handled
 * inside: src/bitbuffer.cpp:45-73 appends descriptors of any length).  A workspace that holds some other buffer's
 * histogram is reported as MH_ERR_CORRUPT by mh_dev_status(d_ws) and nothing is written.  d_data itself must be
 * unchanged too: if the buffer was refilled between the two calls, the header still matches, the regions are priced
 * from the old contents, and what is written (never beyond `cap`) is not a valid stream — every region compares
 * the bits it emitted with its price and mh_dev_status(d_ws) reports MH_ERR_CORRUPT. */
int mh_dev_encode_hist(const mh_model *m, const uint8_t *d_data, size_t n, uint8_t prev0,
                       const uint64_t *d_start_bit,
                       uint8_t *d_payload, size_t cap, uint64_t *d_nbits,
                       uint64_t *d_index, uint32_t chunk_symbols,
                       const void *d_hist_ws, size_t hist_ws_bytes,
                       void *d_ws, size_t ws_bytes, void *stream);

/* Sharded encode (contiguous byte ranges, one rank per shard).  A shard's payload length
 * is known before it is encoded: it is the dot product of the shard's LOCAL histogram with the code
 * lengths of the (global) model.  d_counts: 65536 (order 1) or 256 (order 0) uint64 counts on the
 * device; *d_nbits receives the bits. */
int mh_dev_payload_bits(const mh_model *m, const uint64_t *d_counts, uint64_t *d_nbits, void *stream);
/* mh_dev_encode with the payload emitted pre-shifted: *d_start_bit (device memory, may be NULL = 0) is
 * the global bit position at which this shard starts; its first code is written at bit
 * (*d_start_bit & 7) of d_payload[0], the bits before it are zero, so consecutive shards concatenate at
 * byte offset start_bit / 8 with ONE OR-merged seam byte.  *d_nbits = (*d_start_bit & 7) + payload bits,
 * i.e. the end position inside d_payload; index entries are positions inside d_payload as well, so the
 * shard decodes from its own buffer with mh_dev_decode(nbits = *d_nbits). */
int mh_dev_encode_at(const mh_model *m, const uint8_t *d_data, size_t n, uint8_t prev0,
                     const uint64_t *d_start_bit,
                     uint8_t *d_payload, size_t cap, uint64_t *d_nbits,
                     uint64_t *d_index, uint32_t chunk_symbols,
                     void *d_ws, size_t ws_bytes, void *stream);

/* mh_dev_encode_at with the full start context of the shard: the previous byte (order 0/1 models, ctx0 < 256) or,
 * for an order-2 model, (byte before previous) << 8 | previous byte — a shard of an order-2 stream starts in
 * the context of the last TWO bytes of the shard before it. */
int mh_dev_encode_ctx(const mh_model *m, const uint8_t *d_data, size_t n, uint32_t ctx0,
                      const uint64_t *d_start_bit,
                      uint8_t *d_payload, size_t cap, uint64_t *d_nbits,
                      uint64_t *d_index, uint32_t chunk_symbols,
                      void *d_ws, size_t ws_bytes, void *stream);

/* 64-byte status block + one uint32 per chunk (list of the chunks whose codes exceed the decode
 * tables and are decoded by a second launch); 0 for an invalid chunk size. */
size_t mh_dev_decode_workspace(uint64_t nbits, uint64_t n_symbols, uint32_t chunk_symbols);
/* Parallel decode with an index.  ws_bytes must be at least mh_dev_decode_workspace(...) (MH_ERR_ARG).
 * d_payload and d_out are 16-byte aligned; the kernel reads the payload in aligned 32- or 64-byte pieces,
 * so d_payload must be readable up to the next 64-byte boundary after its last byte (any hipMalloc'ed
 * buffer is).  Errors found on the device (null LUT entry, walk past the end)
 * are reported through the int32 at the start of the workspace: mh_dev_status() reads it. */
int mh_dev_decode(const mh_model *m, const uint8_t *d_payload, uint64_t nbits,
                  uint8_t *d_out, uint64_t n_symbols,
                  const uint64_t *d_index, uint32_t chunk_symbols,
                  void *d_ws, size_t ws_bytes, void *stream);
/* mh_dev_decode for a payload whose length is still on the device (e.g. straight after mh_dev_encode, with
 * no host round trip in between): the kernels read *d_nbits.  nbits_hint (0 = unknown) only steers the
 * choice between kernel variants; results never depend on it. */
int mh_dev_decode_dn(const mh_model *m, const uint8_t *d_payload, const uint64_t *d_nbits, uint64_t nbits_hint,
                     uint8_t *d_out, uint64_t n_symbols,
                     const uint64_t *d_index, uint32_t chunk_symbols,
                     void *d_ws, size_t ws_bytes, void *stream);
/*
 * FINE INDEX — a device-only acceleration structure of the decoder, never part of the stream or of the sidecar
 * index: one uint32 per MH_FINE_SYMBOLS = 64 input bytes,
 *        entry = context byte at that byte << 24 | (payload bit offset of its code & 0xFFFFFF);
 * the chunk index entry in front of it supplies the offset's high bits.  With it one wave decodes 64 adjacent
 * 64-symbol pieces: its compressed input and its output are each one contiguous run of memory (mh_tile.hip)
 * instead of 64 scattered cache lines per access.  It costs 1/16 of the input size in HBM, is written by
 * mh_dev_encode_fine alongside the payload (or by mh_dev_build_index_fine for a stream that came without any
 * index) and is consumed by mh_dev_decode_fine; it lives and dies in device memory.  chunk_symbols <= 4096 for the
 * decoder to use it.  Order-2 models (extension): entry = two context bytes << 16 | bits from the chunk's index entry
 * to the piece (0xFFFF: does not fit), chunk_symbols <= 1024, and only models whose live contexts all have a slot in
 * the decoder's LDS tables (text-like sources: a few hundred contexts) use it; mh_dev_build_index_fine writes none.
 */
#ifndef MH_T_SUB_SHIFT            /* (an experimental build may halve the piece: csrc/Makefile, MH_T_SUB_SHIFT=5) */
#define MH_T_SUB_SHIFT 6
#endif
#define MH_FINE_SYMBOLS (1u << MH_T_SUB_SHIFT)
static inline uint64_t mh_fine_entries(uint64_t n_symbols) { return (n_symbols + MH_FINE_SYMBOLS - 1) / MH_FINE_SYMBOLS; }
/* mh_dev_encode_hist (d_hist_ws may be NULL: then mh_dev_encode_at) that also fills d_fine[mh_fine_entries(n)]
 * (d_fine may be NULL).  Payload, index and *d_nbits do not depend on d_fine. */
int mh_dev_encode_fine(const mh_model *m, const uint8_t *d_data, size_t n, uint8_t prev0,
                       const uint64_t *d_start_bit,
                       uint8_t *d_payload, size_t cap, uint64_t *d_nbits,
                       uint64_t *d_index, uint32_t chunk_symbols, uint32_t *d_fine,
                       const void *d_hist_ws, size_t hist_ws_bytes,
                       void *d_ws, size_t ws_bytes, void *stream);
/* mh_dev_encode_ctx (full start context: two bytes for an order-2 model) that also fills d_fine. */
int mh_dev_encode_ctx_fine(const mh_model *m, const uint8_t *d_data, size_t n, uint32_t ctx0,
                           const uint64_t *d_start_bit,
                           uint8_t *d_payload, size_t cap, uint64_t *d_nbits,
                           uint64_t *d_index, uint32_t chunk_symbols, uint32_t *d_fine,
                           void *d_ws, size_t ws_bytes, void *stream);
/* mh_dev_decode / mh_dev_decode_dn (d_nbits != NULL: the payload length is read there, nbits is a hint) with the
 * fine index of the same stream.  d_fine == NULL, a model without tile tables, chunk_symbols > 4096 or a small
 * stream: exactly mh_dev_decode.  Same output either way; same workspace size. */
int mh_dev_decode_fine(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, const uint64_t *d_nbits,
                       uint8_t *d_out, uint64_t n_symbols,
                       const uint64_t *d_index, uint32_t chunk_symbols, const uint32_t *d_fine,
                       void *d_ws, size_t ws_bytes, void *stream);

/* Index building for a stream without one (what the reference writes: src/coding.cpp:35-59 has no
 * index): parallel fixed-point iteration over 512-byte bit segments — each segment is decoded from a
 * guessed state and re-decoded while its predecessor's end state changes; Huffman streams
 * re-synchronise, so a few passes converge, and a sequential pass is the fallback.  Fills d_index
 * (capacity index_cap entries) and *d_n_symbols.  Unlike the other device calls this one synchronises
 * `stream` between batches of passes (the pass count depends on the data). */
size_t mh_dev_build_index_workspace(uint64_t nbits);
int mh_dev_build_index(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, uint8_t prev0,
                       uint64_t *d_index, uint64_t index_cap, uint32_t chunk_symbols,
                       uint64_t *d_n_symbols, void *d_ws, size_t ws_bytes, void *stream);
/* mh_dev_build_index that also fills the fine index (see above) of the stream: d_fine[fine_cap], one entry per 64
 * symbols (nbits / 64 + 2 entries always suffice: a code has at least one bit). */
int mh_dev_build_index_fine(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, uint8_t prev0,
                            uint64_t *d_index, uint64_t index_cap, uint32_t chunk_symbols,
                            uint32_t *d_fine, uint64_t fine_cap,
                            uint64_t *d_n_symbols, void *d_ws, size_t ws_bytes, void *stream);
/*
 * STREAMS WITHOUT AN INDEX IN TWO PASSES OVER THE PAYLOAD (round 5) — what i_coding_provider::decompress is handed
 * (src/coding.cpp:96-160: a `.cm` carries nothing but the header byte and the bits).  Building both indices and then decoding
 * reads the payload three times; these two calls read it twice and write no index at all:
 *   mh_dev_decode_stream_states  pass 1: the payload is cut into 352-bit segments, every segment is decoded from a guessed
 *       state after a short warm-up, the segments whose guess was wrong are decoded again from their predecessor's end state
 *       until none is left (the fixed point of mh_dev_build_index), a prefix sum of the segments' symbol counts gives every
 *       segment its output offset; *d_n_symbols = the decoded size.  Synchronises `stream` between its passes.
 *       mh_dev_index_path(d_ws) afterwards: 6 = the workspace holds the states, mh_dev_decode_stream_emit may follow;
 *       0 = this model / stream does not take this path (an order-2 model, no tile tables, a code-length lattice, a code
 *       longer than the tile tables resolve, under a megabit, or segments that do not synchronise): build an index
 *       (mh_dev_build_index_fine) and decode from it instead.
 *   mh_dev_decode_stream_emit    pass 2: every segment is decoded once more from its true state and its bytes are written
 *       to d_out[offset of its first symbol ...); nothing is written at or beyond out_cap (MH_ERR_CAPACITY via mh_dev_status).
 *       End state and count of every segment must come out as converged and the stream must end exactly at nbits
 *       (src/coding.cpp:124,158): MH_ERR_CORRUPT otherwise.  No allocation; synchronises `stream` once, before its launch (it
 *       reads the workspace's path word: MH_ERR_ARG when the workspace does not hold the states of this stream).
 * Workspace for both: mh_dev_build_index_workspace(nbits), the same buffer, untouched in between.
 */
int mh_dev_decode_stream_states(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, uint8_t prev0,
                                uint64_t *d_n_symbols, void *d_ws, size_t ws_bytes, void *stream);
int mh_dev_decode_stream_emit(const mh_model *m, const uint8_t *d_payload, uint64_t nbits, uint8_t prev0,
                              uint8_t *d_out, uint64_t out_cap, void *d_ws, size_t ws_bytes, void *stream);
/* Diagnostic: how the last mh_dev_build_index on this workspace arrived at the index — 1 the segment iteration
 * converged, 2 per-group context maps (fixed-length codes), 3 per-group state maps (mixed lengths), 4 the one-lane
 * walk, 0 nothing ran.  Synchronises `stream`. */
int mh_dev_index_path(const void *d_ws, void *stream);
/* Diagnostic: which encoder the last mh_dev_encode* call on this workspace ran — 1 the region encoder (priced from a
 * histogram of the input, one read), 3 the same with its escape variant launched too (model with codes over 12
 * bits), 2 the length pass + emit pair (inputs under 4 MiB without a histogram; order-2 models whose live contexts
 * do not fit the LDS image, or MH_ENCODE2_PATH=two_pass), 4 the one-pass order-2 encoder (enc_chain_kernel: every
 * symbol looked up once, start bits by a chained scan over groups of wave-tiles that are handed out by a ticket counter,
 * so a wait is only ever for a workgroup that is running).  Synchronises.
 * Every wait of the one-pass encoder is bounded; should one run out all the same, the workspace's status is
 * MH_ERR_TIMEOUT (nothing valid was written): mh_encode* retry with the pair by themselves, a caller of mh_dev_encode*
 * runs again with MH_ENCODE2_PATH=two_pass. */
int mh_dev_encode_path(const void *d_ws, void *stream);
/* Diagnostic: which decoder the last mh_dev_decode* call on this workspace ran — 1 the tile decoder (fine index given, 8 MiB
 * or more, average code shorter than the tile tables' first level), 2 the chunk decoder.  Synchronises. */
int mh_dev_decode_path(const void *d_ws, void *stream);
/* Diagnostic: which instantiation of the chunk decoder that call launched (csrc/mh_decode.hip, dec_cfg) — chosen from the
 * model's table layout and the stream's ratio, never from the environment:
 *   0 LDS_WIDE          both table levels in LDS, no code over 8 bits, ratio over 0.6: two streams, 64-byte granules and bursts
 *   1 LDS_SHORT         the same model on a better-compressed stream: four light streams
 *   2 LDS_TWO_LEVEL     both levels in LDS, first level narrower than 8 bits;   3 LDS_TWO_LEVEL_P8   ... of 8 bits
 *   4 L2_DIRECT         first level in LDS, uniform second-level tables of 2^H entries in L2, H read from the model;
 *   5 / 6 / 7 / 8       ... with H = 2 / 3 / 4 / 8 as a compile-time constant (longest code 10 / 11 / 12 / 16 or more bits)
 * -1: the chunk decoder did not run on this workspace (the tile decoder did).  The redo pass behind it (codes longer than
 * both levels, ragged ends: one lane per chunk, tree walk) uses variant 9 REDO_LDS or 10 REDO_L2_DIRECT of the same layout.
 * Synchronises. */
int mh_dev_decode_variant(const void *d_ws, void *stream);
/* Synchronises `stream` and returns the device-side status word of a workspace (MH_OK, MH_ERR_CORRUPT,
 * MH_ERR_TIMEOUT, MH_ERR_CAPACITY). */
int mh_dev_status(const void *d_ws, void *stream);

#ifdef __cplusplus
}
#endif
#endif
