// main.cpp — `markovhuffman`, command-line compatible with the reference's driver (src/main.cpp):
//   markov-huffman <input> [-o output] [-h] [-e encoding_file] [-d output_encoding_file] [-g] [-x]
// Same flag grammar (clustered short flags such as -xh; -o/-e/-d take the following arguments in the
// order the letters appear; a bare "-" is accepted; unknown letters only warn), same validation, same
// progress lines on stderr, same file formats.  All per-byte work runs on the GPU through
// coding.h -> libmhc.so.
//
// Extensions use long options the reference's parser never accepted:
//   --index FILE     write (compress) / read (extract) the chunk-index sidecar that lets decode run in
//                    parallel; without it extraction first rebuilds the index on the device
//   --chunk N        symbols per index entry (power of two, 256..8192; default 1024)
//   --device N       HIP device ordinal
#include <errno.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "coding.h"

static void print_help() {
    eprintf("markov-huffman <input> [-o output] [options]\n");
    eprintf("\t-o output_file\n");
    eprintf("\t-h use simple huffman coding\n");
    eprintf("\n");
    eprintf("\t-e encoding_file\n");
    eprintf("\t-d output_encoding_file\n");
    eprintf("\n");
    eprintf("\t-g print huffman trees and tables\n");
    eprintf("\t-x extract\n");
    eprintf("\n");
    eprintf("\t--index file   chunk-index sidecar for parallel extraction (MI355X extension)\n");
    eprintf("\t--chunk n      symbols per index entry (default 1024)\n");
    eprintf("\t--order2       contexts of two previous bytes (MI355X extension; own file formats)\n");
    eprintf("\t--device n     HIP device ordinal\n");
}

struct options {
    bool extract = false, debug = false, simple_huffman = false;
    const char* input = nullptr;
    const char* output = nullptr;
    const char* encoding_input = nullptr;
    const char* encoding_output = nullptr;
    std::string index_path;
    uint32_t chunk = MH_CHUNK_DEFAULT;
    bool order2 = false;
    int device = -1;
};

// Flag grammar of src/main.cpp:55-101.
static options parse(int argc, char* argv[]) {
    options o;
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        if (a[0] == '-' && a[1] == '-' && a[2] != 0) {           // long options: ours only
            auto need = [&](const char* name) -> const char* {
                if (i + 1 >= argc) { eprintf("Error: Expected a value following %s.\n", name); exit(1); }
                return argv[++i];
            };
            if (!strcmp(a, "--index")) o.index_path = need(a);
            else if (!strcmp(a, "--chunk")) {
                const unsigned long v = strtoul(need(a), nullptr, 10);
                if (v < MH_CHUNK_MIN || v > MH_CHUNK_MAX || (v & (v - 1))) {
                    eprintf("Error: --chunk must be a power of two between %u and %u.\n", MH_CHUNK_MIN, MH_CHUNK_MAX);
                    exit(1);
                }
                o.chunk = (uint32_t)v;
            }
            else if (!strcmp(a, "--device")) o.device = atoi(need(a));
            else if (!strcmp(a, "--order2")) o.order2 = true;
            else eprintf("Warning: Unknown option %s.\n", a);
            continue;
        }
        if (a[0] != '-') {
            if (!o.input) o.input = a;
            else eprintf("Warning: Unexpected positional argument %s.\n", a);
            continue;
        }
        int taken = 0;                                            // following argv entries consumed by this cluster
        for (const char* c = a + 1; *c; ++c) {
            const char** slot = nullptr;
            const char* what = nullptr;
            switch (*c) {
                case 'o': slot = &o.output; what = "output file following -o"; break;
                case 'e': slot = &o.encoding_input; what = "encoding file following -e"; break;
                case 'd': slot = &o.encoding_output; what = "encoding output file following -d"; break;
                case 'x': o.extract = true; break;
                case 'h': o.simple_huffman = true; break;
                case 'g': o.debug = true; break;
                default: eprintf("Warning: Unknown option %c.\n", *c);
            }
            if (slot) {
                if (i + 1 < argc) *slot = argv[i + 1 + taken++];
                else eprintf("Error: Expected %s.\n", what);
            }
        }
        i += taken;
    }
    return o;
}

static FILE* open_or_die(const char* path, const char* mode, const char* what) {
    FILE* f = fopen(path, mode);
    if (!f) {
        eprintf("Error while opening %s; %s.\n", what, strerror(errno));
        exit(1);
    }
    return f;
}

int main(int argc, char* argv[]) {
    if (argc < 2) {
        print_help();
        return 1;
    }
    options o = parse(argc, argv);

    // validation of src/main.cpp:103-115
    if (!o.input) {
        eprintf("Error: Must provide input file.\n");
        exit(1);
    }
    if (o.encoding_input && o.encoding_output) {
        eprintf("Error: Don't provide an encoding input and an encoding output. Just use cp.\n");
        exit(1);
    }
    if (o.extract && !o.encoding_input) {
        eprintf("Error: Must provide encoding file input while in decompress mode.\n");
        exit(1);
    }
    if (mh_device_count() < 1) {
        eprintf("Error: no usable HIP device; this build has no CPU path.\n");
        exit(1);
    }
    if (o.device >= 0) mh_or_die(mh_set_device(o.device), "--device");

    check_access(o.input, false);                                  // src/main.cpp:118-121 (prints only)
    if (o.output) check_access(o.output, true);
    if (o.encoding_input) check_access(o.encoding_input, false);
    if (o.encoding_output) check_access(o.encoding_output, false);

    FILE* input_fd = open_or_die(o.input, "rb", "input");
    FILE* output_fd = o.output ? open_or_die(o.output, "w+b", "output") : stdout;   // read-write: the result is written through a mapping

    i_coding_provider* coder = nullptr;
    if (o.encoding_input) {
        eprintf("Loading encoding table from file...\n");
        FILE* fd = open_or_die(o.encoding_input, "rb", "encoding input");
        bitbuffer buffer(fd, bitbuffer::read);
        if (o.order2) {
            coder = new markov2_huffman_table(buffer);             // its loader checks the order-2 header
        } else
        // first bit: 0 = Huffman tree, 1 = Markov-Huffman file (src/main.cpp:147-161)
        if (buffer.peek_bit() != !o.simple_huffman) {
            eprintf("Error: Incorrect encoding table provided for current operation; expected %s, found %s.\n",
                    o.simple_huffman ? "simple Huffman" : "Markov-Huffman",
                    buffer.peek_bit() ? "Markov-Huffman" : "simple Huffman");
            exit(1);
        }
        if (coder) {}
        else if (buffer.peek_bit() == 0) coder = new huffman_table(buffer);
        else coder = new markov_huffman_table(buffer);
    } else {
        std::vector<uint64_t> counts(o.order2 ? (size_t(1) << 24) : o.simple_huffman ? 256 : 65536);
        if (o.order2) {
            eprintf("Building order-2 Markov-Huffman encoding table from input...\n");
            construct_table(input_fd, 2, counts.data());
            coder = new markov2_huffman_table(counts.data());
        } else if (o.simple_huffman) {
            eprintf("Building simple Huffman encoding table from input...\n");
            construct_table(input_fd, 0, counts.data());
            coder = new huffman_table(counts.data());
        } else {
            eprintf("Building Markov-Huffman encoding table from input...\n");
            construct_table(input_fd, 1, counts.data());
            coder = new markov_huffman_table(counts.data());
        }
        fseek(input_fd, 0, SEEK_SET);                              // src/main.cpp:183
    }
    if (!o.index_path.empty()) coder->set_index_path(o.index_path, o.chunk);

    if (o.debug) {                                                 // src/main.cpp:186-190
        coder->print_table();
        coder->print_tree();
    }

    if (o.encoding_output) {                                       // src/main.cpp:192-202
        FILE* fd = fopen(o.encoding_output, "wb");
        eprintf("Writing encoding table to %s...\n", o.encoding_output);
        if (!fd) {
            eprintf("Error while opening encoding file output; %s.\n", strerror(errno));
            exit(1);
        }
        bitbuffer buffer(fd, bitbuffer::write);
        coder->write_coding_tree(buffer);
    }

    if (o.extract) {
        eprintf("Extracting %s ===> %s...\n", o.input, o.output);
        coder->decompress(input_fd, output_fd);
    } else {
        eprintf("Compressing %s ===> %s...\n", o.input, o.output);
        coder->compress(input_fd, output_fd);
    }
    delete coder;
    eprintf("Done.\n");
    return 0;
}
