// fuzz_table — mutation fuzz of the table-file parser and of everything a loaded model hands out, for the
// AddressSanitizer / UndefinedBehaviorSanitizer build of the HOST code (make -C csrc asan; CPU only: the
// kernels are not instrumented and no device call is made without a GPU).
//
// The reference's loader recurses over the file's bits with no bounds checks (src/huffman.cpp:166-172,
// src/markov_huffman.cpp:15-25) and its asserts are compiled out (Makefile:20-21); a drop-in reads table files a
// user hands it (-e table), so mh_model_from_table_bits must take ANY bytes: MH_ERR_BADTABLE or a model whose
// accessors stay inside their arrays, never a crash.
//
// usage: fuzz_table <cases> <seed> <table file> [<table file> ...]
// Every seed file is loaded as it is, then mutated <cases> times in all: truncations, bit flips, byte
// overwrites, splices of two seeds, runs of 0x00 / 0xFF.  Prints a summary line; exit code 0 unless a seed file
// itself fails to load.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/mh.h"

static uint64_t rng_state = 1;
static uint64_t rnd() {              // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// what a caller does with a model: every accessor, over every index
static void exercise(const mh_model *m) {
    const int type = mh_model_type(m);
    (void)mh_model_max_code_len(m);
    size_t nb = 0;
    if (mh_model_write_table(m, nullptr, 0, &nb) == MH_OK && nb) {
        std::vector<uint8_t> t(nb);
        size_t nb2 = 0;
        (void)mh_model_write_table(m, t.data(), t.size(), &nb2);
        if (nb > 1) (void)mh_model_write_table(m, t.data(), nb - 1, &nb2);       // too small: MH_ERR_CAPACITY
        mh_model *again = nullptr;                                                // what was written loads again
        if (mh_model_from_table_bits(t.data(), nb2, &again) == MH_OK) mh_model_free(again);
    }
    if (type == 2) return;                                                        // order-2 accessors read device tables
    for (int prev = 0; prev < 256; prev += 1 + int(rnd() % 7)) {
        for (int s = 0; s < 256; ++s) {
            int len = 0; uint64_t code = 0;
            (void)mh_model_get_code(m, prev, s, &len, &code);
            int present = 0, internal = 0, value = 0, depth = 0;
            (void)mh_model_get_lut(m, prev, s, &present, &internal, &value, &depth);
        }
    }
    int a = 0, b = 0, c = 0;
    (void)mh_model_decode_layout(m, &a, &b, &c);
    (void)mh_model_tile_layout(m, &a, &b, &c);
    uint64_t nbits = 0;
    (void)mh_stream_header(m, rnd());
    (void)mh_stream_parse_header(m, uint8_t(rnd()), rnd() % 100000, &nbits);
    (void)mh_encode_bound(m, size_t(rnd() % 1000000));
    static std::vector<uint64_t> counts(65536);                                  // (256 x 256 pair counts; a Huffman model reads the first 256)
    for (int i = 0; i < 65536; i += 1 + int(rnd() % 5)) counts[size_t(i)] = rnd() % 1000;
    (void)mh_model_payload_bits(m, counts.data(), &nbits);
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <cases> <seed> <table file>...\n", argv[0]); return 2; }
    const long cases = atol(argv[1]);
    rng_state = strtoull(argv[2], nullptr, 10) * 2654435761ull + 1;
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 3; i < argc; ++i) {
        FILE *f = fopen(argv[i], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[i]); return 2; }
        std::vector<uint8_t> b;
        uint8_t buf[4096];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
        fclose(f);
        mh_model *m = nullptr;
        const int rc = mh_model_from_table_bits(b.data(), b.size(), &m);
        if (rc != MH_OK && !b.empty()) { fprintf(stderr, "seed %s does not load: %d\n", argv[i], rc); return 1; }
        if (rc == MH_OK) { exercise(m); mh_model_free(m); }
        seeds.push_back(std::move(b));
    }
    long ok = 0, bad = 0, other = 0;
    for (long c = 0; c < cases; ++c) {
        std::vector<uint8_t> t = seeds[rnd() % seeds.size()];
        const int kind = int(rnd() % 8);
        if (kind == 0 && !t.empty()) {
            t.resize(rnd() % t.size());                                            // truncation (also to 0 bytes)
        } else if (kind == 1 && !t.empty()) {
            const int flips = 1 + int(rnd() % 8);
            for (int i = 0; i < flips; ++i) t[rnd() % t.size()] ^= uint8_t(1u << (rnd() % 8));
        } else if (kind == 2 && !t.empty()) {
            const int n = 1 + int(rnd() % 16);
            for (int i = 0; i < n; ++i) t[rnd() % t.size()] = uint8_t(rnd());
        } else if (kind == 3) {
            const std::vector<uint8_t> &o = seeds[rnd() % seeds.size()];           // head of one, tail of another
            const size_t cut = t.empty() ? 0 : rnd() % t.size(), from = o.empty() ? 0 : rnd() % o.size();
            t.resize(cut);
            t.insert(t.end(), o.begin() + long(from), o.end());
        } else if (kind == 4 && !t.empty()) {
            const size_t at = rnd() % t.size(), len = 1 + rnd() % 64;              // a run of zeros or ones: deep one-sided trees
            const uint8_t v = (rnd() & 1) ? 0xFF : 0x00;
            for (size_t i = at; i < at + len && i < t.size(); ++i) t[i] = v;
        } else if (kind == 5) {
            t.insert(t.end(), size_t(rnd() % 256), uint8_t(rnd()));                // trailing garbage
        } else if (kind == 6 && !t.empty()) {
            t[0] ^= uint8_t(0x80 >> (rnd() % 3));                                  // the type bit and its neighbours
            if (t.size() > 1) t.resize(1 + rnd() % (t.size() - 1));
        } else {
            t.assign(size_t(rnd() % 512), 0);                                      // all-random bytes
            for (auto &b : t) b = uint8_t(rnd());
        }
        mh_model *m = nullptr;
        const int rc = mh_model_from_table_bits(t.data(), t.size(), &m);
        if (rc == MH_OK) { ++ok; exercise(m); mh_model_free(m); }
        else if (rc == MH_ERR_BADTABLE) ++bad;
        else ++other;
    }
    printf("fuzz_table: %ld cases, %ld loaded, %ld rejected (MH_ERR_BADTABLE), %ld other status\n", cases, ok, bad, other);
    return 0;
}
