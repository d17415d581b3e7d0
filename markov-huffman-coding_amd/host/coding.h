// coding.h — the reference's coding.h face (src/coding.h:9-35, src/huffman.h:8-35,
// src/markov_huffman.h:9-24) on top of the MI355X C ABI (include/mh.h).
//
// Same class names, constructors, public methods, ownership and error conventions as the reference, so
// a main() written against the reference compiles against this header unchanged:
//   huffman_table(int* counts) / markov_huffman_table(int* counts)      histogram -> tables
//   huffman_table(bitbuffer&)  / markov_huffman_table(bitbuffer&)       table file -> tables
//   compress(FILE*, FILE*), decompress(FILE*, FILE*), write_coding_tree(bitbuffer&),
//   print_table(), print_tree()
// Errors: message on stderr + exit(1), like the reference (src/coding.cpp:103-110, src/utils.cpp:62-65).
// The per-byte work — histogram, codeword lookup + bit packing, LUT decode — runs in the HIP kernels
// behind libmhc.so; nothing here computes on the CPU.
#ifndef MHC_HOST_CODING_H
#define MHC_HOST_CODING_H

#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/mh.h"
#include "bitbuffer.h"

// src/utils.h:7-9
#define eprintf(...) fprintf(stderr, __VA_ARGS__)

// src/coding.h:9-16 — variable-length codeword, MSB-first bytes
struct encoding_descriptor {
    int length = 0;
    std::vector<unsigned char> encoding;
    void print();
};

// src/tree.h:9-27 — what decoding_lookup hands back (a view, not a linked tree)
struct tree_node {
    bool is_internal = false;
    unsigned char value = 0;
    int depth = -1;
};

// src/main.cpp:29-39: the histogram pass.  counts64: 65536 entries (order 1) or 256 (order 0),
// counts[256*prev+c] with prev starting at ' ' (order 2, the extension: 1 << 24 entries, counts[ctx*256+c]).
// Reads the stream from its current position to EOF.
void construct_table(FILE* input_fd, int order, uint64_t* counts64);

class i_coding_provider {
public:
    virtual ~i_coding_provider();
    virtual void print_table() = 0;
    virtual void print_tree() = 0;
    virtual void write_coding_tree(bitbuffer& buffer);
    // ownership of both handles is transferred in (src/coding.cpp:93, src/bitbuffer.h:35-40)
    void compress(FILE* input_fd, FILE* output_fd);
    void decompress(FILE* input_fd, FILE* output_fd);
    // extension: chunk index sidecar (not part of the reference's format; see include/mh.h)
    void set_index_path(const std::string& path, uint32_t chunk_symbols) { index_path_ = path; chunk_ = chunk_symbols; }
    const mh_model* model() const { return model_; }

protected:
    i_coding_provider() = default;
    void adopt(mh_model* m) { model_ = m; }
    void build_from_counts(const uint64_t* counts, int order);
    void build_from_buffer(bitbuffer& buffer, int expected_type);
    void print_table_for(int prev);
    int print_tree_for(int prev, bool subgraph, int n, const std::string& label);
    bool context_empty(int prev);

private:
    virtual int get_type() = 0;   // 0 simple Huffman, 1 Markov-Huffman (src/coding.h:29-32)
    virtual encoding_descriptor& get_encoding(unsigned char prev, unsigned char c);
    virtual const tree_node* decoding_lookup(unsigned char prev, unsigned char c);
    mh_model* model_ = nullptr;
    encoding_descriptor scratch_desc_;
    tree_node scratch_node_;
    std::string index_path_;
    std::vector<uint64_t> counts_;               // histogram the tables were built from (empty when loaded from a file)
    uint32_t chunk_ = MH_CHUNK_DEFAULT;
};

class huffman_table : public i_coding_provider {
public:
    explicit huffman_table(int* counts);          // src/huffman.h:14 (256 ints)
    explicit huffman_table(const uint64_t* counts);
    explicit huffman_table(bitbuffer& buffer);    // src/huffman.h:15
    bool empty();
    void print_table() override;
    void print_tree() override;

private:
    int get_type() override { return 0; }
};

class markov_huffman_table : public i_coding_provider {
public:
    explicit markov_huffman_table(int* counts);   // src/markov_huffman.h:12 (65536 ints)
    explicit markov_huffman_table(const uint64_t* counts);
    explicit markov_huffman_table(bitbuffer& buffer);   // src/markov_huffman.h:13
    void print_table() override;
    void print_tree() override;

private:
    int get_type() override { return 1; }
};

// Extension, not in the reference (README.md:158-166 only speculates about it): context = the previous TWO
// bytes.  Own stream magic and table-file header so that the reference's tools do not mistake either
// (include/mh.h, "ORDER 2").  Parity unpinned.
class markov2_huffman_table : public i_coding_provider {
public:
    explicit markov2_huffman_table(const uint64_t* counts);   // 1 << 24 counts, counts[ctx * 256 + c]
    explicit markov2_huffman_table(bitbuffer& buffer);
    void print_table() override;
    void print_tree() override;

private:
    int get_type() override { return 2; }
};

// src/utils.h:13-21
std::string charv(unsigned char c);
void check_access(const char* path, bool write);
int read_buffer(void* ptr, size_t size, size_t count, FILE* stream);
int write_buffer(void* ptr, size_t size, size_t count, FILE* stream);
// exits with the reference's convention when a C-ABI call fails
void mh_or_die(int status, const char* what);

#endif
