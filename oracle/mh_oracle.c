/*
 * mh_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see mh_oracle.h).
 *
 * Plain-C restatement of the reference algorithm, one thread, in memory.
 * Citations are into /root/reference (jeremy-rifkin/Markov-Huffman-Coding).
 * Parity: pinned by tests/test_oracle_golden.py (golden vectors made with the
 * compiled reference, oracle/_ref) — see the header.
 */
#include "mh_oracle.h"

#include <stdlib.h>
#include <string.h>

#define MAXN 511 /* <=256 leaves -> <=255 internal nodes (+2 for the single-symbol hack) */

/* src/tree.h:9-27 — node fields as flat arrays. child = -1 on a leaf. */
typedef struct {
    int nnodes;
    int root; /* -1: empty table (src/huffman.cpp:44-46) */
    int left[MAXN + 2], right[MAXN + 2];
    unsigned char is_internal[MAXN + 2], value[MAXN + 2];
    int64_t weight[MAXN + 2];
    int height[MAXN + 2], depth[MAXN + 2];
    /* src/huffman.h:10-11 — encoding_table[256], decoding_lookup_table[256] */
    int code_len[256];
    unsigned char code_bits[256][32];
    int lut[256]; /* node index or -1 (null) */
} table_t;

struct mho_model {
    int type;       /* src/coding.h:29-32; 2 = the order-2 generalisation (parity unpinned, see mh_oracle.h) */
    table_t *t;     /* 1 table (Huffman) or 256 (Markov, src/markov_huffman.h:10) */
    table_t **t2;   /* order 2: 65536 pointers, NULL = empty context */
};



/* ---------------------------------------------------------------- histogram */

void mho_histogram_o1(const uint8_t *data, size_t n, uint8_t prev0, uint64_t *counts) {
    /* src/main.cpp:32-37: prev carried over the whole stream; 176-178: counts[256*prev+c]++ */
    unsigned prev = prev0;
    memset(counts, 0, 65536 * sizeof(uint64_t));
    for (size_t i = 0; i < n; i++) {
        counts[256u * prev + data[i]]++;
        prev = data[i];
    }
}

void mho_histogram_o0(const uint8_t *data, size_t n, uint64_t *counts) {
    /* src/main.cpp:168-170 */
    memset(counts, 0, 256 * sizeof(uint64_t));
    for (size_t i = 0; i < n; i++) counts[data[i]]++;
}

/* order-2 generalisation (parity unpinned): counts[ctx * 256 + c]++, ctx = (byte before previous) << 8 |
 * previous byte, both ' ' at the start (src/main.cpp:32 doubled) */
void mho_histogram_o2(const uint8_t *data, size_t n, uint64_t *counts) {
    unsigned ctx = 0x2020u;
    memset(counts, 0, (size_t)MHO_O2_COUNTS * sizeof(uint64_t));
    for (size_t i = 0; i < n; i++) {
        counts[((size_t)ctx << 8) | data[i]]++;
        ctx = ((ctx << 8) | data[i]) & 0xFFFFu;
    }
}

/* ------------------------------------------------------- min_pq (exact heap) */

typedef struct {
    int64_t key[257];
    int item[257];
    int size;
} heap_t;

static void heap_swap(heap_t *h, int a, int b) {
    int64_t k = h->key[a]; h->key[a] = h->key[b]; h->key[b] = k;
    int it = h->item[a]; h->item[a] = h->item[b]; h->item[b] = it;
}

/* src/min_pq.tpp:4-7 + 29-36: append, swim while parent key is STRICTLY greater */
static void heap_insert(heap_t *h, int64_t key, int item) {
    int i = h->size++;
    h->key[i] = key; h->item[i] = item;
    while (i != 0 && h->key[(i - 1) / 2] > h->key[i]) {
        heap_swap(h, i, (i - 1) / 2);
        i = (i - 1) / 2;
    }
}

/* src/min_pq.tpp:9-15 + 38-52: last to root, sink; right child chosen only if STRICTLY smaller */
static int heap_pop_min(heap_t *h) {
    int top = h->item[0];
    h->size--;
    h->key[0] = h->key[h->size]; h->item[0] = h->item[h->size];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = 2 * i + 2;
        int t = (r < h->size && h->key[r] < h->key[l]) ? r : l;
        if (t < h->size && h->key[t] < h->key[i]) { heap_swap(h, i, t); i = t; }
        else break;
    }
    return top;
}

/* ------------------------------------------------------------- tree nodes */

static int new_leaf(table_t *t, unsigned char v, int64_t w) { /* src/tree.h:22-23 */
    int n = t->nnodes++;
    t->left[n] = t->right[n] = -1; t->is_internal[n] = 0; t->value[n] = v;
    t->weight[n] = w; t->height[n] = 0; t->depth[n] = -1;
    return n;
}

static int new_internal(table_t *t, int l, int r) { /* src/tree.h:19-21 */
    int n = t->nnodes++;
    t->left[n] = l; t->right[n] = r; t->is_internal[n] = 1; t->value[n] = 0;
    t->weight[n] = t->weight[l] + t->weight[r];
    t->height[n] = (t->height[l] > t->height[r] ? t->height[l] : t->height[r]) + 1;
    t->depth[n] = -1;
    return n;
}

static void table_init(table_t *t) { /* src/huffman.cpp:12-16 */
    t->nnodes = 0; t->root = -1;
    memset(t->code_len, 0, sizeof t->code_len);
    memset(t->code_bits, 0, sizeof t->code_bits);
    for (int i = 0; i < 256; i++) t->lut[i] = -1;
}

/* src/huffman.cpp:97-123 — DFS, left = bit 0 first, then right = bit 1.
 * desc/len model encoding_descriptor (src/coding.cpp:9-27): MSB-first, unused low bits zero. */
static void walk(table_t *t, int node, unsigned char *desc, int depth) {
    if (node < 0) return;
    t->depth[node] = depth; /* :103 */
    if (t->is_internal[node]) {
        if (depth >= 255) return; /* cannot happen for <=256 leaves; guards desc[32] */
        int byte = depth / 8, bit = 7 - depth % 8;
        desc[byte] &= (unsigned char)~(1u << bit);          /* push_bit(0) :105 */
        walk(t, t->left[node], desc, depth + 1);
        desc[byte] |= (unsigned char)(1u << bit);           /* pop_bit, push_bit(1) :107-108 */
        walk(t, t->right[node], desc, depth + 1);
        desc[byte] &= (unsigned char)~(1u << bit);          /* pop_bit resets the bit :110 */
        if (depth == 8) t->lut[desc[0]] = node;             /* :111-113 */
    } else {
        unsigned char v = t->value[node];
        t->code_len[v] = depth;                             /* :115 — later visit overwrites */
        memcpy(t->code_bits[v], desc, 32);
        if (depth <= 8 && depth >= 1) {                     /* :116-121 */
            unsigned cw = desc[0];
            for (int i = 0; i < (1 << (8 - depth)); i++) t->lut[cw + i] = node;
        }
    }
}

static void build_tables(table_t *t) { /* src/huffman.cpp:91-95 */
    unsigned char desc[32];
    memset(desc, 0, sizeof desc);
    walk(t, t->root, desc, 0);
}

/* src/huffman.cpp:131-164 */
static void table_build(table_t *t, const uint64_t *counts) {
    heap_t h; h.size = 0;
    table_init(t);
    for (int i = 0; i < 256; i++)                       /* :134-138 ascending symbol order */
        if (counts[i]) heap_insert(&h, (int64_t)counts[i], new_leaf(t, (unsigned char)i, (int64_t)counts[i]));
    if (h.size == 0) return;                            /* :140-142 */
    while (h.size > 1) {                                /* :143-151 */
        int a = heap_pop_min(&h), b = heap_pop_min(&h);
        if (t->height[a] > t->height[b]) { int x = a; a = b; b = x; }   /* :147-149 */
        int n = new_internal(t, a, b);
        heap_insert(&h, t->weight[n], n);
    }
    t->root = heap_pop_min(&h);                         /* :152 */
    if (!t->is_internal[t->root]) {                     /* :154-162 single-symbol hack */
        int r = t->root;
        t->left[r] = new_leaf(t, t->value[r], t->weight[r]);
        t->right[r] = new_leaf(t, t->value[r], t->weight[r]);
        t->height[r] = 1; t->is_internal[r] = 1;
    }
    build_tables(t);                                    /* :163 */
}

/* ------------------------------------------------------------- bit streams */

typedef struct { const uint8_t *p; size_t nbits, pos; int fail; } bitrd;

static int rd_bit(bitrd *b) { /* src/bitbuffer.cpp:75-90, MSB first */
    if (b->pos >= b->nbits) { b->fail = 1; return 0; }
    int v = (b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
    b->pos++;
    return v;
}
static int rd_byte(bitrd *b) { /* src/bitbuffer.cpp:92-114 */
    int v = 0;
    for (int i = 0; i < 8; i++) v = (v << 1) | rd_bit(b);
    return v;
}

typedef struct { uint8_t *p; size_t cap; uint64_t nbits; } bitwr;

static void wr_bit(bitwr *b, int v) { /* src/bitbuffer.cpp:9-19; buffer pre-zeroed (bitbuffer.h:32-33) */
    size_t byte = (size_t)(b->nbits >> 3);
    if (byte < b->cap) {
        if ((b->nbits & 7) == 0) b->p[byte] = 0;
        if (v) b->p[byte] |= (uint8_t)(1u << (7 - (b->nbits & 7)));
    }
    b->nbits++;
}
static void wr_byte(bitwr *b, int v) { /* src/bitbuffer.cpp:21-43 */
    for (int i = 7; i >= 0; i--) wr_bit(b, (v >> i) & 1);
}

/* ---------------------------------------------------- table (de)serialise */

/* src/huffman.cpp:166-172.  The FIRST subtree read is the left child (the reference relies on
 * left-to-right argument evaluation; SURVEY §8c portability caveat). */
static int load_tree(table_t *t, bitrd *b, int depth) {
    if (b->fail || t->nnodes >= MAXN || depth > 256) { b->fail = 1; return -1; }
    if (rd_bit(b)) {
        int v = rd_byte(b);
        return new_leaf(t, (unsigned char)v, 0);
    }
    int l = load_tree(t, b, depth + 1);
    int r = load_tree(t, b, depth + 1);
    if (b->fail || l < 0 || r < 0) { b->fail = 1; return -1; }
    return new_internal(t, l, r);
}

/* src/huffman.cpp:174-188: pre-order; internal -> 0, leaf -> 1 + 8-bit value */
static void save_tree(const table_t *t, int node, bitwr *b) {
    if (node < 0) return;
    if (t->is_internal[node]) wr_bit(b, 0);
    else { wr_bit(b, 1); wr_byte(b, t->value[node]); }
    save_tree(t, t->left[node], b);
    save_tree(t, t->right[node], b);
}

mho_model *mho_model_from_counts(const uint64_t *counts, int order) {
    mho_model *m = (mho_model *)calloc(1, sizeof *m);
    if (order == 2) {                                   /* generalisation: one table per two-byte context */
        m->type = 2;
        m->t2 = (table_t **)calloc(MHO_O2_CONTEXTS, sizeof(table_t *));
        for (unsigned c = 0; c < MHO_O2_CONTEXTS; c++) {
            const uint64_t *row = counts + 256u * (size_t)c;
            int any = 0;
            for (int s = 0; s < 256 && !any; s++) any = row[s] != 0;
            if (!any) continue;
            m->t2[c] = (table_t *)malloc(sizeof(table_t));
            table_build(m->t2[c], row);
        }
        return m;
    }
    int nt = order ? 256 : 1;
    m->type = order ? 1 : 0;
    m->t = (table_t *)calloc((size_t)nt, sizeof(table_t));
    for (int i = 0; i < nt; i++) table_build(&m->t[i], counts + 256 * i); /* src/markov_huffman.cpp:10-12 */
    return m;
}

static const unsigned char O2_MAGIC[4] = { 'M', 'H', '2', 1 };
static int is_o2_table(const uint8_t *bytes, size_t n) {
    if (n < 37 || bytes[0] != 0x80) return 0;
    for (int i = 1; i < 33; i++) if (bytes[i]) return 0;
    return memcmp(bytes + 33, O2_MAGIC, 4) == 0;
}

mho_model *mho_model_from_table(const uint8_t *bytes, size_t n, int *err) {
    bitrd b = { bytes, n * 8, 0, 0 };
    mho_model *m = (mho_model *)calloc(1, sizeof *m);
    if (err) *err = MHO_OK;
    /* src/main.cpp:147-161: first bit 0 -> Huffman tree, 1 -> Markov file */
    int first = (n > 0) ? ((bytes[0] >> 7) & 1) : 0;
    if (is_o2_table(bytes, n)) {                        /* order-2 generalisation (see mh_oracle.h) */
        bitrd b2 = { bytes + 37, (n - 37) * 8, 0, 0 };
        m->type = 2;
        m->t2 = (table_t **)calloc(MHO_O2_CONTEXTS, sizeof(table_t *));
        for (unsigned c = 0; c < MHO_O2_CONTEXTS; c++) {
            if (rd_bit(&b2)) {
                table_t *t = m->t2[c] = (table_t *)malloc(sizeof(table_t));
                table_init(t);
                t->root = load_tree(t, &b2, 0);
                if (b2.fail || t->root < 0 || !t->is_internal[t->root]) goto bad;
                build_tables(t);
            }
            if (b2.fail) goto bad;
        }
        return m;
    }
    if (first == 0) {
        m->type = 0;
        m->t = (table_t *)calloc(1, sizeof(table_t));
        table_init(&m->t[0]);
        m->t[0].root = load_tree(&m->t[0], &b, 0);      /* src/huffman.cpp:22-25 */
        if (b.fail || m->t[0].root < 0 || !m->t[0].is_internal[m->t[0].root]) goto bad;
        build_tables(&m->t[0]);
    } else {
        m->type = 1;
        m->t = (table_t *)calloc(256, sizeof(table_t));
        rd_bit(&b);                                     /* src/markov_huffman.cpp:17 */
        for (int i = 0; i < 256; i++) {                 /* :19-24 */
            table_init(&m->t[i]);
            if (rd_bit(&b)) {
                m->t[i].root = load_tree(&m->t[i], &b, 0);
                if (b.fail || m->t[i].root < 0 || !m->t[i].is_internal[m->t[i].root]) goto bad;
                build_tables(&m->t[i]);
            }
            if (b.fail) goto bad;
        }
    }
    return m;
bad:
    if (err) *err = MHO_ERR_BADTABLE;
    mho_model_free(m);
    return NULL;
}

void mho_model_free(mho_model *m) {
    if (!m) return;
    if (m->t2) { for (unsigned c = 0; c < MHO_O2_CONTEXTS; c++) free(m->t2[c]); free(m->t2); }
    free(m->t);
    free(m);
}

int mho_model_type(const mho_model *m) { return m->type; }

size_t mho_model_write_table(const mho_model *m, uint8_t *out, size_t cap) {
    bitwr b = { out, cap, 0 };
    if (m->type == 2) {                                 /* empty order-1 table + magic + 65536 x (0 | 1 + tree) */
        wr_bit(&b, 1);
        for (int i = 0; i < 256; i++) wr_bit(&b, 0);
        for (int i = 0; i < 7; i++) wr_bit(&b, 0);
        for (int i = 0; i < 4; i++) wr_byte(&b, O2_MAGIC[i]);
        for (unsigned c = 0; c < MHO_O2_CONTEXTS; c++) {
            const table_t *t = m->t2[c];
            wr_bit(&b, t != NULL);
            if (t) save_tree(t, t->root, &b);
        }
        return (size_t)((b.nbits + 7) / 8);
    }
    if (m->type == 0) {
        save_tree(&m->t[0], m->t[0].root, &b);          /* src/huffman.cpp:83-85 */
    } else {
        wr_bit(&b, 1);                                  /* src/markov_huffman.cpp:81 */
        for (int i = 0; i < 256; i++) {                 /* :82-87 */
            int nonempty = m->t[i].root >= 0;
            wr_bit(&b, nonempty);
            if (nonempty) save_tree(&m->t[i], m->t[i].root, &b);
        }
    }
    return (size_t)((b.nbits + 7) / 8);                 /* src/bitbuffer.cpp:175 */
}

/* -------------------------------------------------------------- lookups */

static table_t g_empty_table;      /* an empty context of an order-2 model: no codes, null LUT */
static int g_empty_ready = 0;
static const table_t *empty_table(void) {
    if (!g_empty_ready) { table_init(&g_empty_table); g_empty_ready = 1; }
    return &g_empty_table;
}

/* prev: the previous byte (orders 0/1) or (byte before previous) << 8 | previous byte (order 2) */
static const table_t *ctx_table(const mho_model *m, int prev) {
    if (m->type == 2) { const table_t *t = m->t2[prev & 0xFFFF]; return t ? t : empty_table(); }
    return m->type ? &m->t[prev & 255] : &m->t[0];      /* src/markov_huffman.cpp:52-58 / src/huffman.cpp:71-73 */
}
/* context after symbol c */
static unsigned next_ctx(const mho_model *m, unsigned ctx, unsigned c) {
    return m->type == 2 ? (((ctx << 8) | c) & 0xFFFFu) : c;
}
static unsigned first_ctx(const mho_model *m) { return m->type == 2 ? 0x2020u : (unsigned)' '; }

void mho_get_code(const mho_model *m, int prev, int sym, int *len, uint8_t *bits32) {
    const table_t *t = ctx_table(m, prev);
    *len = t->code_len[sym & 255];
    memcpy(bits32, t->code_bits[sym & 255], 32);
}

void mho_export_codes(const mho_model *m, uint8_t *len8, uint64_t *code64) {
    for (int p = 0; p < 256; p++) {
        const table_t *t = ctx_table(m, p);
        for (int s = 0; s < 256; s++) {
            int l = t->code_len[s];
            uint64_t c = 0;
            for (int i = 0; i < l && i < 64; i++)
                c = (c << 1) | ((t->code_bits[s][i >> 3] >> (7 - (i & 7))) & 1u);
            len8[p * 256 + s] = (uint8_t)l;
            code64[p * 256 + s] = c;
        }
    }
}

void mho_export_codes_o2(const mho_model *m, uint8_t *len8, uint64_t *code64) {
    for (unsigned p = 0; p < MHO_O2_CONTEXTS; p++) {
        const table_t *t = ctx_table(m, (int)p);
        for (int s = 0; s < 256; s++) {
            int l = t->code_len[s];
            uint64_t c = 0;
            for (int i = 0; i < l && i < 64; i++)
                c = (c << 1) | ((t->code_bits[s][i >> 3] >> (7 - (i & 7))) & 1u);
            len8[(size_t)p * 256 + s] = (uint8_t)l;
            code64[(size_t)p * 256 + s] = c;
        }
    }
}

void mho_get_lut(const mho_model *m, int prev, int w, int *present, int *is_internal, int *value, int *depth) {
    const table_t *t = ctx_table(m, prev);
    int n = t->lut[w & 255];
    *present = n >= 0;
    *is_internal = *value = *depth = 0;
    if (n >= 0) { *is_internal = t->is_internal[n]; *value = t->value[n]; *depth = t->depth[n]; }
}

/* ------------------------------------------------------------- compress */

size_t mho_compress(const mho_model *m, const uint8_t *in, size_t n, uint8_t *out, size_t cap, uint64_t *nbits) {
    /* src/coding.cpp:61-94.  out[0] is the header; payload starts at out[1]. */
    bitwr b = { cap ? out + 1 : out, cap ? cap - 1 : 0, 0 };
    unsigned prev = first_ctx(m);                       /* :67 (order 2: both context bytes start as ' ') */
    uint64_t acc = 0; int accn = 0;                     /* MSB-first accumulator of pending bits */
    for (size_t i = 0; i < n; i++) {
        const table_t *t = ctx_table(m, (int)prev);     /* :71 get_encoding(prev, c) */
        unsigned c = in[i];
        int l = t->code_len[c];                         /* length 0: symbol silently skipped under NDEBUG (:72) */
        prev = next_ctx(m, prev, c);                    /* :74 */
        if (l <= 32) {                                  /* src/bitbuffer.cpp:45-73, batched */
            const unsigned char *cb = t->code_bits[c];
            uint64_t v = ((uint64_t)cb[0] << 24) | ((uint64_t)cb[1] << 16) | ((uint64_t)cb[2] << 8) | cb[3];
            if (l) { acc = (acc << l) | (v >> (32 - l)); accn += l; }
            while (accn >= 8) {
                size_t byte = (size_t)(b.nbits >> 3);
                if (byte < b.cap) b.p[byte] = (uint8_t)(acc >> (accn - 8));
                b.nbits += 8; accn -= 8;
            }
        } else {
            for (int k = 0; k < l; k++) {
                acc = (acc << 1) | ((t->code_bits[c][k >> 3] >> (7 - (k & 7))) & 1u); accn++;
                if (accn == 8) {
                    size_t byte = (size_t)(b.nbits >> 3);
                    if (byte < b.cap) b.p[byte] = (uint8_t)acc;
                    b.nbits += 8; accn = 0;
                }
            }
        }
    }
    uint64_t total = b.nbits + (uint64_t)accn;
    if (accn) {                                         /* src/bitbuffer.cpp:175: round up, zero padded */
        size_t byte = (size_t)(b.nbits >> 3);
        if (byte < b.cap) b.p[byte] = (uint8_t)(acc << (8 - accn));
    }
    int bi = (int)(total & 7);                          /* src/coding.cpp:85 get_bi() */
    if (cap) out[0] = m->type == 2 ? (uint8_t)(0x40 | ((8 - bi) % 8))         /* order 2: own magic nibble */
                                   : (uint8_t)(0x30 | ((~m->type & 1) << 3) | ((8 - bi) % 8));  /* :88 */
    if (nbits) *nbits = total;
    return 1 + (size_t)((total + 7) / 8);
}

/* ----------------------------------------------------------- decompress */

int64_t mho_decompress(const mho_model *m, const uint8_t *in, size_t n, uint8_t *out, size_t cap) {
    /* src/coding.cpp:96-160 */
    if (n < 1) return MHO_ERR_CORRUPT;
    unsigned header = in[0];                            /* :100 */
    if (m->type == 2) {
        if ((header & 0xF8) != 0x40) return (header & 0xF0) == 0x30 ? MHO_ERR_TYPE : MHO_ERR_CORRUPT;
    } else {
        if ((header & 0xF0) != 0x30) return MHO_ERR_CORRUPT;                    /* :103-106 */
        if (((~(header & (1 << 3)) >> 3) & 1) != (unsigned)m->type) return MHO_ERR_TYPE;  /* :107-110 */
    }
    int remainder = header & 7;                         /* :111 */
    int64_t length = (int64_t)(n - 1) * 8 - remainder;  /* :115 (reference: int) */
    const uint8_t *p = in + 1;
    int64_t avail = (int64_t)(n - 1) * 8;               /* bits physically present */
    int64_t pos = 0;                                    /* bitbuffer read cursor */
    unsigned prev = first_ctx(m);                       /* :118 */
    int64_t bi = 0;                                     /* :120 */
    unsigned w = 0; int wi = 0;                         /* :122-123 */
    int64_t nout = 0;
    while (bi < length) {                               /* :124 */
        /* pop_rest (src/bitbuffer.cpp:129-140): refill to 8 bits, zero bits past EOF (116-127) */
        for (int k = wi; k < 8; k++) {
            unsigned bit = 0;
            if (pos < avail) { bit = (p[pos >> 3] >> (7 - (pos & 7))) & 1u; pos++; }
            w |= bit << (7 - k);
        }
        const table_t *t = ctx_table(m, (int)prev);
        int node = t->lut[w & 255];                     /* :126 decoding_lookup(prev, w) */
        if (node < 0) return MHO_ERR_CORRUPT;           /* reference: null deref under NDEBUG */
        if (t->is_internal[node]) {                     /* :129-149 */
            bi += 8;
            for (;;) {
                if (pos >= avail) return MHO_ERR_CORRUPT;
                unsigned bit = (p[pos >> 3] >> (7 - (pos & 7))) & 1u; pos++;
                bi++;
                node = bit ? t->right[node] : t->left[node];
                if (node < 0) return MHO_ERR_CORRUPT;
                if (!t->is_internal[node]) {
                    if ((size_t)nout < cap) out[nout] = t->value[node];
                    nout++;
                    prev = next_ctx(m, prev, t->value[node]);
                    break;
                }
            }
            w = 0; wi = 0;                              /* :147-148 */
        } else {                                        /* :150-156 */
            if ((size_t)nout < cap) out[nout] = t->value[node];
            nout++;
            prev = next_ctx(m, prev, t->value[node]);
            w = (w << t->depth[node]) & 0xFF;
            wi = 8 - t->depth[node];
            bi += t->depth[node];
        }
    }
    return nout;
}
