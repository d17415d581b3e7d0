// mh_model.hpp — host-side model of the Markov-Huffman codec: per-context Huffman trees with the
// reference's exact tie-breaking, code tables, decode LUTs, table-file (de)serialisation, and the
// packed images the HIP kernels consume.  Product code (no dependency on oracle/).
//
// Reference behaviour restated here (file:line into jeremy-rifkin/Markov-Huffman-Coding):
//   tree build        src/huffman.cpp:131-164, src/min_pq.tpp:4-52, src/tree.h:19-23
//   codes + LUT       src/huffman.cpp:91-123
//   table file        src/huffman.cpp:166-188, src/markov_huffman.cpp:15-25,80-88
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mh {

// ---- device table formats (shared with the kernel files: mh_encode.hip, mh_decode.hip, mh_tile.hip, mh_tree.hip) ------------------------------------------

// Encode table, LDS-resident: 65536 x u16, entry = len(4) << 12 | code(12) for len <= 12,
// 0 = "no code" (symbol skipped, src/coding.cpp:72 under NDEBUG), ENC16_ESCAPE = look the
// codeword up in the full (len8, code64) tables in HBM/L2.
// Index = enc_slot(window) where window = sym << 8 | prev is the raw little-endian 16-bit field
// read straight out of the byte stream.  The low byte is mixed, prev ^ rotl8(sym, 3), so that the few
// very frequent (prev, sym) pairs of a skewed source land in different LDS banks and words (with a
// plain or plainly XOR-ed low byte, Zipf-like data puts ~80 % of all accesses into four banks: small
// prev ^ small sym is small; the rotation moves the symbol's low bits up to bits 3..7).  It is a
// bijection of prev for every sym, and for four packed symbols it costs five instructions on the
// device: bfi(0xF8F8F8F8, x << 3, x >> 5) ^ (x << 8 | previous byte).
constexpr uint16_t ENC16_ESCAPE = 0xFFFF;
constexpr int ENC16_MAX_LEN = 12;
constexpr uint32_t slot_mix(uint32_t sym) { return ((sym << 3) | (sym >> 5)) & 0xFFu; }
constexpr uint32_t enc_slot(uint32_t window) {
    return (window & 0xFF00u) | ((window & 0xFFu) ^ slot_mix(window >> 8));
}
constexpr uint32_t enc_slot_prev(uint32_t slot) { return (slot & 0xFFu) ^ slot_mix(slot >> 8); }

// Decode tables: two levels, BOTH LDS-resident whenever they fit (a code that needs an L2 gather per
// symbol costs the whole wave ~10x an LDS lookup, and with 64 lanes some lane always needs it).
//   prim[ctx << P | first P stream bits]   u16, P = dec_bits, model-wide, 4..8
//       0      null (empty context)
//       leaf   0x8000 | len(1..P) << 8 | symbol          (0x8000 alone = null: no code has this prefix)
//       inner  (h - 1) << 12 | off : the internal node at depth P; its 2^h-entry table starts
//              at sec[sec_base[ctx] + off] and is indexed by the next h stream bits (1 <= h <= 8)
//   sec[...]   u16
//       leaf   0x8000 | total_len(P+1..P+h) << 8 | symbol  (5-bit length field, bits 8..12)
//       inner  tree node id : code longer than P + h, walked bit by bit in the L2 tree (rare)
// Leaves carry the flag so that (a) max(first-level entry, second-level entry) IS the resolving entry
// when lanes that need no second level read 0 there, and (b) a leaf used as a table id indexes past
// the end of `sec`, which a range-checked buffer load turns into that 0 without touching memory.
// P is the largest width for which prim + sec fit the LDS budget; if none does P = 8 and sec stays in
// HBM/L2.  In that case the tables are made uniform (2^H entries each, H = min(max_len - 8, 8)) and an
// inner entry is just the global table id — no per-context base and no per-node height to decode
// on the device (falls back to the general form above 32767 tables).  P = 8 is the reference's own 8-bit LUT (src/huffman.cpp:97-123).
constexpr uint16_t DEC16_LEAF = 0x8000;
constexpr uint16_t DEC16_NULL = DEC16_LEAF;           // leaf of length 0: consumes nothing, the chunk then ends at the wrong bit
constexpr int DEC_LDS_ENTRIES = (163840 - 1024) / 2;   // u16 entries beside the 1 KiB sec_base array
constexpr int DEC_SEC_MAX_PER_CTX = 4096;              // 12-bit offsets
// Last-resort tree in HBM/L2: per context 256 x u32 = right << 16 | left; a child is
// 0x8000 | symbol for a leaf, else the internal-node id (0 = root).
constexpr uint32_t TREE_LEAF = 0x8000;
constexpr int TREE_STRIDE = 256;

constexpr int MAX_CODE_BITS = 64;  // device path limit (MH_ERR_CODE_TOO_LONG beyond)

struct Code {
    int len = 0;                       // bits; 0 = no code
    std::array<uint64_t, 4> bits{};    // MSB-first: bit i of the code is bit (63 - i % 64) of bits[i / 64]
    uint64_t right_aligned() const {   // valid for len <= 64
        return len == 0 ? 0 : (len >= 64 ? bits[0] : bits[0] >> (64 - len));
    }
};

struct Node {
    int16_t child[2] = {-1, -1};  // -1 on a leaf
    uint8_t sym = 0;
    bool leaf = true;
    int64_t weight = 0;
    int height = 0;
    int depth = -1;
};

class BitReader;
class BitWriter;

// One context = one Huffman table (huffman_table in the reference).
class ContextCoder {
public:
    ContextCoder() { lut_.fill(-1); }  // (a context nobody built has no nodes: its LUT must say so — found by the UBSan fuzz)
    void clear();
    bool empty() const { return root_ < 0; }
    void build_from_counts(const uint64_t *counts256);
    bool load(BitReader &in);          // reads one serialized tree; false on a malformed stream
    // takes over a tree built elsewhere (the device build): nodes in creation order, child = 0xFFFF on a leaf
    void adopt(int nnodes, int root, const uint16_t *left, const uint16_t *right, const uint8_t *sym);
    void save(BitWriter &out) const;
    const Code &code(int sym) const { return codes_[sym & 255]; }
    int lut(int w) const { return lut_[w & 255]; }   // node index or -1
    const Node &node(int i) const { return nodes_[i]; }
    int root() const { return root_; }
    int max_len() const { return max_len_; }
    // second-level entries this context needs for primary width P and table-height cap hcap
    size_t sec_entries(int P, int hcap) const;
    // the same for every P in 0..8 at once (one tree walk): out[P]
    void sec_profile(int hcap, size_t (&out)[9]) const;
    // packed device images for this context: prim (1 << P entries), this context's second-level
    // tables appended to `sec` (offsets relative to sec_start), walk tree (TREE_STRIDE entries)
    // uniform_h > 0: every table has exactly 2^uniform_h entries and the inner entry holds the table's rank
    // within this context instead of (height, offset)
    void pack_decode(int P, int hcap, int uniform_h, uint16_t *prim, std::vector<uint16_t> &sec, size_t sec_start, uint32_t *tree256) const;

private:
    void derive_tables();
    int add_leaf(uint8_t sym, int64_t w);
    int add_inner(int l, int r);
    std::vector<Node> nodes_;
    int root_ = -1;
    std::array<Code, 256> codes_{};
    std::array<int, 256> lut_{};
    int max_len_ = 0;
};

class BitReader {
public:
    BitReader(const uint8_t *p, size_t nbytes) : p_(p), nbits_(nbytes * 8) {}
    int bit() {
        if (pos_ >= nbits_) { fail_ = true; return 0; }
        int v = (p_[pos_ >> 3] >> (7 - (pos_ & 7))) & 1;
        ++pos_;
        return v;
    }
    int byte() { int v = 0; for (int i = 0; i < 8; ++i) v = (v << 1) | bit(); return v; }
    bool failed() const { return fail_; }
private:
    const uint8_t *p_;
    size_t nbits_, pos_ = 0;
    bool fail_ = false;
};

class BitWriter {
public:
    void bit(int v) {
        if ((nbits_ & 7) == 0) buf_.push_back(0);
        if (v) buf_.back() |= uint8_t(1u << (7 - (nbits_ & 7)));
        ++nbits_;
    }
    void byte(int v) { for (int i = 7; i >= 0; --i) bit((v >> i) & 1); }
    const std::vector<uint8_t> &bytes() const { return buf_; }  // zero padded to a byte
private:
    std::vector<uint8_t> buf_;
    size_t nbits_ = 0;
};

// Whole model: 1 context (simple Huffman, type 0) or 256 (Markov-Huffman, type 1).
class Model {
public:
    int type = 1;
    std::vector<ContextCoder> ctx;      // size 1 or 256
    std::array<uint64_t, 256> ctx_weight{};   // symbols seen per context (0 for a loaded table)
    const ContextCoder &context(int prev) const { return type ? ctx[prev & 255] : ctx[0]; }
    int max_code_len() const;
    void build_from_counts(const uint64_t *counts, int order);
    bool load_table(const uint8_t *bytes, size_t n);
    std::vector<uint8_t> save_table() const;

    // Packed images (always 256 contexts; a type-0 model replicates its single table).
    struct Packed {
        std::vector<uint16_t> enc16;     // 65536, slot order (enc_slot)
        std::vector<uint8_t> len8;       // 65536, prev*256+sym
        std::vector<uint8_t> len_slot;   // 65536, slot order: code length (0..64) for the length pass
        std::vector<uint64_t> code64;    // 65536, prev*256+sym, right aligned
        std::vector<uint16_t> dec_prim;  // 256 << dec_bits
        std::vector<uint16_t> dec_sec;   // second-level tables
        std::vector<uint32_t> sec_base;  // 256: index of each context's first second-level entry
        std::vector<uint32_t> tree;      // 256*TREE_STRIDE
        int dec_bits = 8;                // P
        bool dec_lds = true;             // prim + sec fit the LDS budget
        bool dec_direct = false;         // L2 mode with uniform tables: inner entry = 0x8000 | global table id,
        int dec_h = 0;                   //   table t = sec[t << dec_h .. (t + 1) << dec_h)
        uint32_t sec_lds_entries = 0;    // leading sec entries kept in LDS (all of them when dec_lds; else
                                         // the tables of the most frequent contexts, which are laid out first)
        int max_len = 0;
        bool any_escape = false;
    };
    Packed pack() const;

    // Tables of the tile decoder (mh_tile.hip): first level of P bits (5..8) per context, uniform second-level
    // tables of 2^H entries (H = min(max(max_len - P, 1), 8)), BOTH indexed by the window's bits LSB-first (first
    // stream bit = bit 0).  An inner first-level entry is the global table id; context c's tables start at id
    // c << P (P <= 7; sparse: most ids stay unused) or at the running count of depth-8 inner nodes (P = 8; then
    // empty when there are more than 32767 such nodes).  Same images as tree_pack_kernel with lsb = 1.
    struct TilePacked { int P = 0, H = 0; std::vector<uint16_t> prim, sec; };
    TilePacked pack_tile(int P) const;
};

}  // namespace mh
