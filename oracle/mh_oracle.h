/*
 * mh_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C restatement of the reference's Markov-Huffman hot path
 * (jeremy-rifkin/Markov-Huffman-Coding, src/).  It exists so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can check the HIP
 * path bit for bit.  Nothing under markov-huffman-coding_amd/ may include,
 * link or call this file: the product path is the HIP library and fails
 * loudly without it.
 *
 * Parity status: PINNED for orders 0 and 1 (the order-2 section at the end is a generalisation the
 * reference does not implement: PARITY UNPINNED, see there).  tests/test_oracle_golden.py checks this code
 * against (a) the golden vectors of SURVEY.md §8(c) — produced in this
 * container by the genuine reference compiled into oracle/_ref/ — and
 * (b) the reference binary itself whenever oracle/_ref/markovhuffman exists.
 *
 * Every function cites the reference file:line it restates.  Arithmetic
 * differs from the reference in one documented way: counts and weights are
 * 64-bit here (reference: int, src/main.cpp:166,174; src/tree.h:14), so
 * results agree wherever the reference does not overflow (SURVEY §8c
 * "parity domain").
 */
#ifndef MH_ORACLE_H
#define MH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mho_model mho_model;

enum {
    MHO_OK = 0,
    MHO_ERR_CORRUPT = -1,  /* src/coding.cpp:103-106 "Input appears corrupt" */
    MHO_ERR_TYPE = -2,     /* src/coding.cpp:107-110 table/file type mismatch */
    MHO_ERR_CAPACITY = -3,
    MHO_ERR_BADTABLE = -4,
};

/* src/main.cpp:29-39 + 176-178: counts[256*prev + c]++, prev starts at prev0 (' '). */
void mho_histogram_o1(const uint8_t *data, size_t n, uint8_t prev0, uint64_t *counts /*65536*/);
/* src/main.cpp:29-39 + 168-170: counts[c]++. */
void mho_histogram_o0(const uint8_t *data, size_t n, uint64_t *counts /*256*/);

/* order 0: src/huffman.cpp:18-20 (256 counts); order 1: src/markov_huffman.cpp:9-13 (65536 counts). */
mho_model *mho_model_from_counts(const uint64_t *counts, int order);
/* src/main.cpp:137-161 + src/markov_huffman.cpp:15-25 + src/huffman.cpp:22-25,166-172. */
mho_model *mho_model_from_table(const uint8_t *bytes, size_t n, int *err);
void mho_model_free(mho_model *m);
/* src/coding.h:29-32: 0 simple Huffman, 1 Markov-Huffman. */
int mho_model_type(const mho_model *m);

/* src/markov_huffman.cpp:80-88 / src/huffman.cpp:83-85,174-188; flush pads to a byte (src/bitbuffer.cpp:170-180).
 * Returns the number of bytes needed; writes them if cap suffices. */
size_t mho_model_write_table(const mho_model *m, uint8_t *out, size_t cap);

/* src/markov_huffman.cpp:52-54 → src/huffman.cpp:71-73.  len = code length in bits (0: no code);
 * bits32 receives the codeword MSB-first (src/coding.cpp:9-16), 32 bytes. */
void mho_get_code(const mho_model *m, int prev, int sym, int *len, uint8_t *bits32);
/* Bulk form for table comparisons: len8[prev*256+sym], code64 right-aligned (valid when len<=64). */
void mho_export_codes(const mho_model *m, uint8_t *len8 /*65536*/, uint64_t *code64 /*65536*/);
/* src/markov_huffman.cpp:56-58 → src/huffman.cpp:87-89.  present=0 → null entry.
 * For a leaf: value/depth; for the internal node stored at depth 8: is_internal=1. */
void mho_get_lut(const mho_model *m, int prev, int w, int *present, int *is_internal, int *value, int *depth);

/* src/coding.cpp:61-94: header byte + MSB-first payload.  Returns total bytes (1 + ceil(bits/8));
 * writes them if cap suffices.  *nbits (optional) receives the payload length in bits. */
size_t mho_compress(const mho_model *m, const uint8_t *in, size_t n, uint8_t *out, size_t cap, uint64_t *nbits);
/* src/coding.cpp:96-160.  in = whole compressed file (header + payload).  Returns decoded byte count
 * (writes min(count, cap) bytes) or a negative MHO_ERR_*. */
int64_t mho_decompress(const mho_model *m, const uint8_t *in, size_t n, uint8_t *out, size_t cap);

/* ---------------------------------------------------------------------------------------------
 * ORDER 2 (context = the previous TWO bytes) — GENERALISATION, PARITY UNPINNED.
 *
 * The reference implements order 1 only and merely speculates about higher orders
 * (README.md:158-166), so nothing here can be checked against it.  This is the SPEC the HIP path is
 * compared with: the same per-context algorithm (src/huffman.cpp:131-164, the same heap, the same
 * tie-breaks, the same one-symbol hack) applied to 65536 contexts ctx = (byte before previous) << 8 |
 * previous byte, both starting as ' ' (the reference's initial context, src/main.cpp:32, doubled).
 *   counts      counts[ctx * 256 + c], 1 << 24 entries
 *   stream      header byte 0x40 | unused bits of the last byte (the reference's magic is 0x30,
 *               src/coding.cpp:103-106: its decompress reports such a file as corrupt instead of
 *               decoding it with the wrong tables), then the MSB-first payload as for order 1
 *   table file  the 33 bytes of an EMPTY order-1 table (bit 1 + 256 zero bits, zero padded: what the
 *               reference's loader accepts and what main()'s first-bit sniffing, src/main.cpp:147-161,
 *               takes for a Markov table with no contexts), then the magic "MH2\x01", then for each of
 *               the 65536 contexts bit 0 (empty) or bit 1 + the pre-order tree of
 *               src/huffman.cpp:174-188, zero padded to a byte.  A genuine empty order-1 table is exactly
 *               33 bytes long, so the two cannot be confused.
 */
#define MHO_O2_CONTEXTS 65536
#define MHO_O2_COUNTS (1u << 24)
void mho_histogram_o2(const uint8_t *data, size_t n, uint64_t *counts /* 1 << 24 */);
/* order == 2 in mho_model_from_counts takes 1 << 24 counts; mho_model_from_table recognises the
 * order-2 file; mho_model_type returns 2; mho_compress / mho_decompress / mho_model_write_table work
 * on such a model as described above. */
/* Bulk export for order 2: len8[ctx*256+sym], code64 right-aligned; 1 << 24 entries each. */
void mho_export_codes_o2(const mho_model *m, uint8_t *len8, uint64_t *code64);

#ifdef __cplusplus
}
#endif
#endif
