// huffman.h — forwarding header: the reference's callers include "huffman.h" (src/main.cpp:10) for
// class huffman_table (src/huffman.h:8-35); here that class is declared in coding.h.
#ifndef MHC_HOST_HUFFMAN_H
#define MHC_HOST_HUFFMAN_H
#include "coding.h"
#endif
