// mh_model.cpp — see mh_model.hpp.  Host-side, integer-only, deterministic.
#include "mh_model.hpp"

#include <algorithm>
#include <utility>

namespace mh {

namespace {

// Array binary min-heap keyed on weight ONLY, with the reference's comparison directions
// (src/min_pq.tpp:29-52): swim while parent > child (strict), sink towards the right child only when
// it is strictly smaller than the left, and stop when the chosen child is not strictly smaller.
// These rules fix every tie and therefore every codeword.
class TieExactHeap {
public:
    // Same comparisons, in the same order, as swap-based swim/sink; a "hole" is moved instead of
    // swapping pairs (the sequence of element positions is identical).
    void push(int64_t key, int item) {
        int i = n_++;
        while (i != 0) {
            int parent = (i - 1) / 2;
            if (!(key_[parent] > key)) break;
            key_[i] = key_[parent]; item_[i] = item_[parent];
            i = parent;
        }
        key_[i] = key; item_[i] = item;
    }
    int pop() {
        const int top = item_[0];
        --n_;
        const int64_t key = key_[n_];
        const int item = item_[n_];
        int i = 0;
        for (;;) {
            int l = 2 * i + 1, r = l + 1;
            int pick = (r < n_ && key_[r] < key_[l]) ? r : l;
            if (pick < n_ && key_[pick] < key) {
                key_[i] = key_[pick]; item_[i] = item_[pick];
                i = pick;
            } else {
                break;
            }
        }
        if (n_ > 0 || i == 0) { key_[i] = key; item_[i] = item; }
        return top;
    }
    int size() const { return n_; }

private:
    int64_t key_[257];
    int item_[257];
    int n_ = 0;
};

template <typename F>
void parallel_for(int n, F body) {   // contexts are independent; kept serial (host threads did not pay off)
    for (int i = 0; i < n; ++i) body(i);
}

inline void code_set_bit(Code &c, int pos, int v) {
    uint64_t mask = 1ull << (63 - (pos & 63));
    if (v) c.bits[pos >> 6] |= mask; else c.bits[pos >> 6] &= ~mask;
}

}  // namespace

void ContextCoder::clear() {
    nodes_.clear();
    nodes_.reserve(520);
    root_ = -1;
    codes_.fill(Code{});
    lut_.fill(-1);
    max_len_ = 0;
}

int ContextCoder::add_leaf(uint8_t sym, int64_t w) {
    Node n;
    n.sym = sym; n.leaf = true; n.weight = w; n.height = 0;
    nodes_.push_back(n);
    return int(nodes_.size()) - 1;
}

int ContextCoder::add_inner(int l, int r) {
    Node n;
    n.child[0] = int16_t(l); n.child[1] = int16_t(r); n.leaf = false;
    n.weight = nodes_[l].weight + nodes_[r].weight;                    // src/tree.h:20
    n.height = std::max(nodes_[l].height, nodes_[r].height) + 1;       // src/tree.h:21
    nodes_.push_back(n);
    return int(nodes_.size()) - 1;
}

void ContextCoder::build_from_counts(const uint64_t *counts) {
    clear();
    TieExactHeap heap;
    for (int s = 0; s < 256; ++s)                                      // src/huffman.cpp:134-138
        if (counts[s]) heap.push(int64_t(counts[s]), add_leaf(uint8_t(s), int64_t(counts[s])));
    if (heap.size() == 0) return;                                      // :140-142 empty context
    while (heap.size() > 1) {                                          // :143-151
        int a = heap.pop();
        int b = heap.pop();
        if (nodes_[a].height > nodes_[b].height) std::swap(a, b);      // :147-149
        int merged = add_inner(a, b);
        heap.push(nodes_[merged].weight, merged);
    }
    root_ = heap.pop();
    if (nodes_[root_].leaf) {                                          // :154-162 one-symbol context
        uint8_t s = nodes_[root_].sym;
        int64_t w = nodes_[root_].weight;
        int l = add_leaf(s, w), r = add_leaf(s, w);
        Node &rt = nodes_[root_];
        rt.child[0] = int16_t(l); rt.child[1] = int16_t(r);
        rt.leaf = false; rt.height = 1;
    }
    derive_tables();
}

void ContextCoder::adopt(int nnodes, int root, const uint16_t *left, const uint16_t *right, const uint8_t *sym) {
    clear();
    if (root < 0 || nnodes <= 0) return;
    nodes_.resize(size_t(nnodes));
    for (int i = 0; i < nnodes; ++i) {
        Node &n = nodes_[i];
        n.leaf = left[i] == 0xFFFF;
        n.child[0] = n.leaf ? int16_t(-1) : int16_t(left[i]);
        n.child[1] = n.leaf ? int16_t(-1) : int16_t(right[i]);
        n.sym = sym[i];
    }
    // subtree heights (children were created before their parent, except under a one-symbol root)
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < nnodes; ++i) {
            Node &n = nodes_[i];
            n.height = n.leaf ? 0 : std::max(nodes_[n.child[0]].height, nodes_[n.child[1]].height) + 1;
        }
    root_ = root;
    derive_tables();
}

// Iterative DFS, left (bit 0) before right (bit 1) — src/huffman.cpp:97-123.  A later leaf with the
// same symbol overwrites the earlier code (:115), which is what makes the one-symbol code "1".
void ContextCoder::derive_tables() {
    codes_.fill(Code{});
    lut_.fill(-1);
    max_len_ = 0;
    if (root_ < 0) return;
    // frames carry the first 64 path bits; deeper paths (pathological tables only) rebuild the full
    // 256-bit code from parent links
    struct Frame { int node; int depth; uint64_t path; };
    std::vector<int> parent(nodes_.size(), -1);
    std::vector<Frame> stack;
    stack.reserve(64);
    stack.push_back({root_, 0, 0});
    while (!stack.empty()) {
        Frame f = stack.back();
        stack.pop_back();
        Node &n = nodes_[f.node];
        n.depth = f.depth;
        if (!n.leaf) {
            if (f.depth == 8) lut_[int(f.path >> 56)] = f.node;                   // :111-113
            if (f.depth >= 255) continue;
            parent[n.child[0]] = parent[n.child[1]] = f.node;
            const uint64_t one = f.depth < 64 ? (1ull << (63 - f.depth)) : 0;
            stack.push_back({n.child[1], f.depth + 1, f.path | one});   // popped second
            stack.push_back({n.child[0], f.depth + 1, f.path});         // popped first: left before right
        } else {
            Code c;
            c.len = f.depth;
            c.bits[0] = f.path;
            if (f.depth > 64) {
                int child = f.node;
                for (int d = f.depth - 1; d >= 64; --d) {
                    int par = parent[child];
                    code_set_bit(c, d, nodes_[par].child[1] == child);
                    child = par;
                }
            }
            codes_[n.sym] = c;
            if (f.depth >= 1 && f.depth <= 8) {                                   // :116-121
                int base = int(f.path >> 56);
                for (int i = 0; i < (1 << (8 - f.depth)); ++i) lut_[base + i] = f.node;
            }
        }
    }
    // max_len_ reflects surviving codes only (a duplicate symbol may have been overwritten)
    for (const Code &c : codes_) max_len_ = std::max(max_len_, c.len);
}

// Pre-order: inner -> 0, leaf -> 1 + 8-bit symbol (src/huffman.cpp:174-188).
void ContextCoder::save(BitWriter &out) const {
    if (root_ < 0) return;
    std::vector<int> stack{root_};
    while (!stack.empty()) {
        int i = stack.back();
        stack.pop_back();
        const Node &n = nodes_[i];
        if (n.leaf) {
            out.bit(1);
            out.byte(n.sym);
        } else {
            out.bit(0);
            stack.push_back(n.child[1]);
            stack.push_back(n.child[0]);
        }
    }
}

// src/huffman.cpp:166-172.  The first subtree in the stream is the LEFT child (the reference leaves
// this to the compiler's argument evaluation order; g++ and clang agree — SURVEY §8c).
bool ContextCoder::load(BitReader &in) {
    clear();
    // Iterative reconstruction: `open` holds inner nodes still waiting for children.
    struct Open { int node; int filled; };
    std::vector<Open> open;
    int leaves = 0;
    for (;;) {
        if (in.failed()) { clear(); return false; }
        int made;
        if (in.bit()) {
            int s = in.byte();
            if (in.failed() || ++leaves > 257) { clear(); return false; }
            made = add_leaf(uint8_t(s), 0);
        } else {
            Node n;
            n.leaf = false;
            nodes_.push_back(n);
            made = int(nodes_.size()) - 1;
            if (open.size() > 256) { clear(); return false; }
        }
        if (root_ < 0) root_ = made;
        if (!open.empty()) {
            Open &o = open.back();
            nodes_[o.node].child[o.filled++] = int16_t(made);
        }
        if (!nodes_[made].leaf) open.push_back({made, 0});
        while (!open.empty() && open.back().filled == 2) {
            Node &n = nodes_[open.back().node];
            n.height = std::max(nodes_[n.child[0]].height, nodes_[n.child[1]].height) + 1;
            open.pop_back();
        }
        if (open.empty()) break;
    }
    if (in.failed() || nodes_[root_].leaf) { clear(); return false; }  // writer never emits a bare leaf
    int inner = 0;
    for (const Node &n : nodes_) inner += !n.leaf;
    if (inner > TREE_STRIDE) { clear(); return false; }
    derive_tables();
    return true;
}

size_t ContextCoder::sec_entries(int P, int hcap) const {
    if (root_ < 0) return 0;
    size_t total = 0;
    std::vector<std::pair<int, int>> stack{{root_, 0}};
    while (!stack.empty()) {
        auto [n, d] = stack.back();
        stack.pop_back();
        const Node &nd = nodes_[n];
        if (nd.leaf) continue;
        if (d == P) { total += size_t(1) << std::min(nd.height, hcap); continue; }
        stack.push_back({nd.child[0], d + 1});
        stack.push_back({nd.child[1], d + 1});
    }
    return total;
}

void ContextCoder::sec_profile(int hcap, size_t (&out)[9]) const {
    for (size_t &v : out) v = 0;
    if (root_ < 0) return;
    std::vector<std::pair<int, int>> stack{{root_, 0}};
    while (!stack.empty()) {
        auto [n, d] = stack.back();
        stack.pop_back();
        const Node &nd = nodes_[n];
        if (nd.leaf) continue;
        out[d] += size_t(1) << std::min(nd.height, hcap);
        if (d == 8) continue;
        stack.push_back({nd.child[0], d + 1});
        stack.push_back({nd.child[1], d + 1});
    }
}

void ContextCoder::pack_decode(int P, int hcap, int uniform_h, uint16_t *prim, std::vector<uint16_t> &sec, size_t sec_start,
                               uint32_t *tree) const {
    for (int i = 0; i < (1 << P); ++i) prim[i] = DEC16_NULL;
    for (int i = 0; i < TREE_STRIDE; ++i) tree[i] = 0;
    if (root_ < 0) return;
    // inner-node ids for the last-resort walk, root = 0
    std::vector<int> id(nodes_.size(), -1);
    int next = 1;
    for (size_t i = 0; i < nodes_.size(); ++i)
        if (!nodes_[i].leaf) id[i] = (int(i) == root_) ? 0 : next++;
    auto enc_child = [&](int c) -> uint32_t {
        return nodes_[c].leaf ? (TREE_LEAF | nodes_[c].sym) : uint32_t(id[c]);
    };
    for (size_t i = 0; i < nodes_.size(); ++i)
        if (!nodes_[i].leaf)
            tree[id[i]] = (enc_child(nodes_[i].child[1]) << 16) | enc_child(nodes_[i].child[0]);

    struct Item { int node; int depth; uint32_t path; };
    // fills a 2^width table (dst) with the subtree under `top` (itself at depth `above`): leaves ->
    // LEAF | total len << 8 | sym for every completion of their path, internal nodes at depth `width`
    // -> inner_entry(node)
    auto fill = [&](int top, int width, int above, uint16_t *dst, auto inner_entry) {
        std::vector<Item> st{{top, 0, 0}};
        while (!st.empty()) {
            Item it = st.back();
            st.pop_back();
            const Node &nd = nodes_[it.node];
            if (nd.leaf) {
                uint32_t lo = it.path << (width - it.depth);
                for (uint32_t k = 0; k < (1u << (width - it.depth)); ++k)
                    dst[lo + k] = uint16_t(DEC16_LEAF | ((above + it.depth) << 8) | nd.sym);
            } else if (it.depth == width) {
                dst[it.path] = inner_entry(it.node);
            } else {
                st.push_back({nd.child[1], it.depth + 1, (it.path << 1) | 1u});
                st.push_back({nd.child[0], it.depth + 1, it.path << 1});
            }
        }
    };
    fill(root_, P, 0, prim, [&](int node) -> uint16_t {
        const int h = uniform_h > 0 ? uniform_h : std::min(nodes_[node].height, hcap);   // >= 1: the node is internal
        const size_t off = sec.size() - sec_start;
        sec.resize(sec.size() + (size_t(1) << h), DEC16_NULL);
        fill(node, h, P, sec.data() + sec_start + off, [&](int deep) -> uint16_t { return uint16_t(id[deep]); });
        if (uniform_h > 0) return uint16_t(off >> h);                               // rank within this context
        return uint16_t(((h - 1) << 12) | uint32_t(off));
    });
}

int Model::max_code_len() const {
    int m = 0;
    for (const ContextCoder &c : ctx) m = std::max(m, c.max_len());
    return m;
}

void Model::build_from_counts(const uint64_t *counts, int order) {
    type = order ? 1 : 0;
    ctx.assign(order ? 256 : 1, ContextCoder{});
    parallel_for(int(ctx.size()), [&](int i) { ctx[i].build_from_counts(counts + 256 * i); });  // src/markov_huffman.cpp:10-12
    ctx_weight.fill(0);
    for (size_t i = 0; i < ctx.size(); ++i)
        for (int s = 0; s < 256; ++s) ctx_weight[i] += counts[256 * i + s];
}

bool Model::load_table(const uint8_t *bytes, size_t n) {
    BitReader in(bytes, n);
    // src/main.cpp:147-161: a Huffman tree file starts with 0 (its root), a Markov file with 1.
    int first = n ? ((bytes[0] >> 7) & 1) : 0;
    if (!first) {
        type = 0;
        ctx.assign(1, ContextCoder{});
        return n > 0 && ctx[0].load(in);
    }
    type = 1;
    ctx.assign(256, ContextCoder{});
    in.bit();                                                       // src/markov_huffman.cpp:17
    for (int p = 0; p < 256; ++p) {                                 // :19-24
        if (in.bit()) {
            if (!ctx[p].load(in)) return false;
        }
        if (in.failed()) return false;
    }
    return true;
}

std::vector<uint8_t> Model::save_table() const {
    BitWriter out;
    if (type == 0) {
        ctx[0].save(out);                                           // src/huffman.cpp:83-85
    } else {
        out.bit(1);                                                 // src/markov_huffman.cpp:81
        for (int p = 0; p < 256; ++p) {                             // :82-87
            out.bit(!ctx[p].empty());
            ctx[p].save(out);
        }
    }
    return out.bytes();
}

Model::Packed Model::pack() const {
    Packed pk;
    pk.enc16.assign(65536, 0);
    pk.len8.assign(65536, 0);
    pk.len_slot.assign(65536, 0);
    pk.code64.assign(65536, 0);
    pk.tree.assign(256 * TREE_STRIDE, 0);
    pk.max_len = max_code_len();
    // ---- decode tables: widest primary whose two levels fit the LDS budget
    const int nctx = type ? 256 : 1;                 // a type-0 model shares one set of tables
    int hcap = 8;
    auto profile = [&](int cap, size_t (&total)[9], size_t (&worst)[9]) {
        std::vector<std::array<size_t, 9>> per(nctx);
        parallel_for(nctx, [&](int i) { size_t o[9]; ctx[i].sec_profile(cap, o); for (int P = 0; P < 9; ++P) per[i][P] = o[P]; });
        for (int P = 0; P < 9; ++P) { total[P] = worst[P] = 0; for (int i = 0; i < nctx; ++i) { total[P] += per[i][P]; worst[P] = std::max(worst[P], per[i][P]); } }
    };
    size_t total[9], worst[9];
    profile(8, total, worst);
    pk.dec_bits = 0;
    for (int P = 8; P >= 4 && !pk.dec_bits; --P)
        if (worst[P] <= size_t(DEC_SEC_MAX_PER_CTX) && (size_t(256) << P) + total[P] <= size_t(DEC_LDS_ENTRIES)) pk.dec_bits = P;
    pk.dec_lds = pk.dec_bits != 0;
    int uniform_h = 0;
    if (!pk.dec_lds) {
        pk.dec_bits = 8;
        // uniform tables: one per depth-8 internal node (profile with hcap = 0 counts the nodes)
        size_t ntab[9], wtab[9];
        profile(0, ntab, wtab);
        // (always: a context has at most 127 inner nodes at depth 8 — 128 would hold all 256 leaves below depth 8, a Kraft sum of
        //  1/2 — so 256 contexts have at most 32 512; the general layout behind the else is kept for a table file that lies)
        if (ntab[8] <= 32767) uniform_h = std::min(std::max(pk.max_len - 8, 1), 8);
        else while (worst[8] > size_t(DEC_SEC_MAX_PER_CTX) && hcap > 1) { --hcap; profile(hcap, total, worst); }
    }
    pk.dec_direct = uniform_h > 0;
    pk.dec_h = uniform_h;
    const int P = pk.dec_bits;
    pk.dec_prim.assign(size_t(256) << P, 0);
    pk.sec_base.assign(256, 0);
    std::vector<std::vector<uint16_t>> sec_of(nctx);
    parallel_for(256, [&](int prev) {
        const ContextCoder &c = context(prev);
        if (type != 0 || prev == 0)
            c.pack_decode(P, hcap, uniform_h, &pk.dec_prim[size_t(prev) << P], sec_of[prev], 0, &pk.tree[prev * TREE_STRIDE]);
        for (int sym = 0; sym < 256; ++sym) {
            const Code &cd = c.code(sym);
            uint32_t window = uint32_t(sym) << 8 | uint32_t(prev);
            uint16_t e = 0;
            if (cd.len > ENC16_MAX_LEN) e = ENC16_ESCAPE;
            else if (cd.len > 0) e = uint16_t((cd.len << 12) | uint32_t(cd.right_aligned()));
            pk.enc16[enc_slot(window)] = e;
            pk.len_slot[enc_slot(window)] = uint8_t(std::min(cd.len, 255));
            pk.len8[prev * 256 + sym] = uint8_t(std::min(cd.len, 255));
            pk.code64[prev * 256 + sym] = cd.len <= 64 ? cd.right_aligned() : 0;
        }
    });
    pk.any_escape = pk.max_len > ENC16_MAX_LEN;
    // second-level tables of the most frequent contexts first: when not everything fits LDS, the
    // leading part that does is still served from there
    std::vector<int> order(256);
    for (int i = 0; i < 256; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ctx_weight[a] > ctx_weight[b]; });
    for (int prev : order) {
        if (type == 0 && prev > 0) continue;
        pk.sec_base[prev] = uint32_t(pk.dec_sec.size());
        pk.dec_sec.insert(pk.dec_sec.end(), sec_of[prev].begin(), sec_of[prev].end());
        if (pk.dec_direct)      // rank within the context -> global table id
            for (int w = 0; w < (1 << P); ++w) {
                uint16_t &e = pk.dec_prim[(size_t(prev) << P) + w];
                if (!(e & DEC16_LEAF)) e = uint16_t((pk.sec_base[prev] >> uniform_h) + e);
            }
    }
    if (type == 0)
        for (int prev = 1; prev < 256; ++prev) {
            std::copy(pk.dec_prim.begin(), pk.dec_prim.begin() + (1 << P), pk.dec_prim.begin() + (size_t(prev) << P));
            std::copy(pk.tree.begin(), pk.tree.begin() + TREE_STRIDE, pk.tree.begin() + size_t(prev) * TREE_STRIDE);
        }
    const size_t room = size_t(DEC_LDS_ENTRIES) - (size_t(256) << P);
    pk.sec_lds_entries = uint32_t(std::min(pk.dec_sec.size(), room) & ~size_t(7));
    if (pk.dec_lds) pk.sec_lds_entries = uint32_t(pk.dec_sec.size());
    return pk;
}

static uint32_t bit_reverse(uint32_t v, int width) {
    uint32_t r = 0;
    for (int i = 0; i < width; ++i) r |= ((v >> i) & 1u) << (width - 1 - i);
    return r;
}

Model::TilePacked Model::pack_tile(int P) const {
    TilePacked t;
    if (P < 5 || P > 8) return t;
    const int max_len = max_code_len();
    const int H = std::min(std::max(max_len - P, 1), 8);
    const int nctx = type ? 256 : 1;
    // the MSB-first images of every context (pack_decode with uniform tables: an inner entry is the table's rank
    // within its context in path order), then the LSB-first permutation
    std::vector<std::vector<uint16_t>> prim_of(nctx), sec_of(nctx);
    parallel_for(nctx, [&](int c) {
        prim_of[c].assign(size_t(1) << P, DEC16_NULL);
        uint32_t tree[TREE_STRIDE];
        ctx[c].pack_decode(P, 8, H, prim_of[c].data(), sec_of[c], 0, tree);
    });
    std::vector<size_t> first_id(256, 0);
    size_t ntab = size_t(256) << P;
    if (P == 8) {
        ntab = 0;
        for (int c = 0; c < 256; ++c) { first_id[c] = ntab; ntab += sec_of[type ? c : 0].size() >> H; }
        if (ntab > 32767) return t;
    } else {
        for (int c = 0; c < 256; ++c) first_id[c] = size_t(c) << P;
    }
    t.P = P; t.H = H;
    t.prim.assign(size_t(256) << P, DEC16_NULL);
    t.sec.assign(ntab << H, 0);
    parallel_for(256, [&](int c) {
        const std::vector<uint16_t> &pm = prim_of[type ? c : 0], &sm = sec_of[type ? c : 0];
        size_t rank = 0;
        for (uint32_t v = 0; v < (1u << P); ++v) {               // tables are laid out by increasing LSB-first window value
            const uint16_t e = pm[bit_reverse(v, P)];
            if (e & DEC16_LEAF) { t.prim[(size_t(c) << P) | v] = e; continue; }
            const size_t id = first_id[c] + rank++;
            t.prim[(size_t(c) << P) | v] = uint16_t(id);
            for (uint32_t x = 0; x < (1u << H); ++x) t.sec[(id << H) + x] = sm[(size_t(e) << H) + bit_reverse(x, H)];
        }
    });
    return t;
}

}  // namespace mh
