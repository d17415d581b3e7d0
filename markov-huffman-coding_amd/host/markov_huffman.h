// markov_huffman.h — forwarding header: the reference's callers include "markov_huffman.h"
// (src/main.cpp:11) for class markov_huffman_table (src/markov_huffman.h:9-24); here that class is
// declared in coding.h.
#ifndef MHC_HOST_MARKOV_HUFFMAN_H
#define MHC_HOST_MARKOV_HUFFMAN_H
#include "coding.h"
#endif
