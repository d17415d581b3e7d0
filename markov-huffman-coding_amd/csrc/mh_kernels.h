// mh_kernels.h — launch interface between the C-ABI layer (mh_api.cpp) and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mhk {

// device-side status word values (first int32 of every workspace)
enum { MHK_STATUS_OK = 0, MHK_STATUS_TIMEOUT = 1, MHK_STATUS_CAPACITY = 2, MHK_STATUS_CORRUPT = 3 };

struct EncParams {
    const uint8_t *data;          // n input bytes, 16-byte aligned
    uint64_t n;
    uint32_t prev0;
    uint32_t chunk_shift;         // log2(chunk_symbols)
    uint8_t *out;                 // payload, 16-byte aligned
    uint64_t cap;                 // bytes available at out
    const uint16_t *enc16;        // 65536 entries, slot order
    const uint8_t *len8;          // 65536, prev*256+sym
    const uint64_t *code64;       // 65536, prev*256+sym
    unsigned long long *nbits;    // out: payload bits
    unsigned long long *index;    // out: chunk index or nullptr
    uint64_t seed;                // (tail7 << 55 | start bit) of the virtual tile -1; 0 for a fresh stream
    // filled by launch_encode from the workspace
    unsigned long long *desc;
    unsigned int *ticket;
    int *status;
    uint32_t ntiles;
};

struct DecParams {
    const uint8_t *payload;
    uint64_t payload_bytes;
    uint64_t nbits;
    uint8_t *out;
    uint64_t n;                   // symbols to produce
    const unsigned long long *index;
    uint64_t nchunks;
    uint32_t chunk_shift;
    const uint16_t *dec16;        // 65536, prev*256+window
    const uint32_t *tree;         // 256 * TREE_STRIDE
    int *status;
};

struct IdxParams {
    const uint8_t *payload;
    uint64_t payload_bytes;
    uint64_t nbits;
    uint32_t prev0;
    uint32_t chunk_shift;
    unsigned long long *index;
    uint64_t index_cap;
    unsigned long long *n_symbols;
    const uint16_t *dec16;
    const uint32_t *tree;
    int *status;
};

hipError_t launch_hist_o1(const uint8_t *d_data, uint64_t n, uint32_t prev0, unsigned long long *d_counts, hipStream_t st);
hipError_t launch_hist_o0(const uint8_t *d_data, uint64_t n, unsigned long long *d_counts, hipStream_t st);
uint64_t encode_tiles(uint64_t n);
size_t encode_workspace_bytes(uint64_t n);
hipError_t launch_encode(EncParams p, void *d_ws, hipStream_t st);
hipError_t launch_decode(DecParams p, void *d_ws, hipStream_t st);
hipError_t launch_build_index(IdxParams p, void *d_ws, hipStream_t st);

}  // namespace mhk
