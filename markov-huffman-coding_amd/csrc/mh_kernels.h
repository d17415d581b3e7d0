// mh_kernels.h — launch interface between the C-ABI layer (mh_api.cpp) and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mhk {

// device-side status word values (first int32 of every workspace)
enum { MHK_STATUS_OK = 0, MHK_STATUS_TIMEOUT = 1, MHK_STATUS_CAPACITY = 2, MHK_STATUS_CORRUPT = 3 };

// host-side description of one encode call (device pointers)
struct EncodeArgs {
    int order;                    // 0/1: the LDS-table path; 2: order-2 contexts, tables gathered from HBM/L2
    const uint8_t *data;          // n input bytes, 16-byte aligned
    uint64_t n;
    uint32_t prev0;               // context before the first byte (order 2: the 16-bit context)
    uint32_t chunk_shift;         // log2(chunk_symbols)
    uint8_t *out;                 // payload, 16-byte aligned
    uint64_t cap;                 // bytes available at out
    const uint16_t *enc16;        // 65536 entries, slot order
    const uint8_t *len_slot;      // 65536 code lengths, slot order (length pass)
    const uint8_t *len8;          // 65536, prev*256+sym
    const uint64_t *code64;       // 65536, prev*256+sym
    const uint64_t *enc64;        // order 2, optional: len << 56 | code per (ctx, sym), len 255 = longer than 56 bits (see len8 / code64)
    unsigned long long *nbits;    // out: payload bits
    unsigned long long *index;    // out: chunk index or nullptr
    const unsigned long long *start_bit;   // device: global bit position of this payload (low 3 bits used) or nullptr
    uint32_t *fine;               // out, optional: the device-only fine index (TileParams), one entry per 64 input bytes
    int max_len;                  // the model's longest code (the region encoder launches its escape variant only above 12)
    const uint8_t *o2hot;         // order 2, optional: the live contexts' tables for LDS (mh_encode.hip, o2hot_lookup16)
    uint32_t o2hot_bytes;
    bool no_chain;                // order 2: the length pass + emit pair even where the one-pass encoder would run
};

struct LenParams {
    const uint8_t *data;
    uint64_t n;
    uint32_t prev0;
    const uint8_t *len_slot;
    uint32_t *wt_bits;
    uint64_t nwt;
    const uint8_t *o2hot;         // order 2 with the hot image in LDS (else nullptr)
    uint32_t o2hot_bytes;
};

struct ScanParams {
    unsigned long long *wt_start;
    const unsigned long long *blk_sum;
    uint64_t nwt, nblk;
    uint8_t *out;
    uint64_t cap;
    unsigned long long *nbits;
    int *status;
};

struct EmitParams {
    const uint8_t *data;
    uint64_t n;
    uint32_t prev0;
    uint32_t chunk_shift;
    uint8_t *out;
    const uint16_t *enc16;
    const uint8_t *len8;
    const uint64_t *code64;
    const uint64_t *enc64;        // order 2 only (EncodeArgs)
    const unsigned long long *wt_start;
    uint64_t nwt;
    unsigned long long *index;
    const int *status;
    uint32_t *fine;               // optional: fine index (see TileParams; order 2: context << 16 | bits from the chunk's index entry)
    const uint8_t *o2hot;         // order 2 with the hot image in LDS (else nullptr)
    uint32_t o2hot_bytes;
};

struct DecParams {
    int order;                    // 2: order-2 tables (general form, P = 8, 65536 contexts)
    const uint8_t *payload;
    uint64_t payload_bytes;
    uint64_t nbits;               // payload bits (a hint for the variant choice when d_nbits is set)
    const unsigned long long *d_nbits;   // when not null the kernels read the payload length from here
    uint8_t *out;
    uint64_t n;                   // symbols to produce
    const unsigned long long *index;
    uint64_t nchunks;
    uint32_t chunk_shift;
    const uint16_t *prim;         // 256 << P entries (mh_model.hpp)
    const uint16_t *sec;          // nsec second-level entries, buffer padded to 16 bytes
    const uint32_t *sec_base;     // 256
    const uint32_t *tree;         // 256 * TREE_STRIDE
    uint32_t P;
    uint32_t nsec;
    uint32_t sec_lds;             // 1: prim + sec fit LDS
    uint32_t direct;              // 1: uniform tables of 2^H entries, inner entry = table id
    uint32_t H;
    int *status;
    uint32_t *redo;               // [0] = number of chunks handed to the redo pass, [1..] their numbers (nchunks capacity)
    uint32_t redo_list;           // order 2: decode only the chunks listed in `redo`
};

// ---- tile decoder (mh_tile.hip) ----------------------------------------------------------------------------
// The FINE INDEX is a device-only acceleration structure (never part of the sidecar file): one uint32 per
// T_SUB = 64 symbols, entry = context byte at the sub-chunk's first symbol << 24 | (payload bit offset of that
// symbol & 0xFFFFFF).  The full offset is recovered from the chunk index entry in front of it (a chunk holds
// far fewer than 2^24 bits).  With it a WAVE decodes 64 adjacent sub-chunks: its compressed bytes are one
// contiguous piece of the payload (staged through LDS with coalesced loads) and so is its output.
#ifndef MH_T_SUB_SHIFT
#define MH_T_SUB_SHIFT 6
#endif
constexpr int T_SUB_SHIFT = MH_T_SUB_SHIFT;
constexpr uint32_t FINE_POS_MASK = 0x00FFFFFFu;
struct TileParams {
    const uint8_t *payload;
    uint64_t payload_bytes;
    uint64_t nbits;
    const unsigned long long *d_nbits;   // when not null the kernels read the payload length from here
    uint8_t *out;
    uint64_t n;                          // symbols to produce
    const unsigned long long *index;     // chunk index (order 1: context << 56 | bit offset)
    uint64_t nchunks;
    uint32_t chunk_shift;
    const uint32_t *fine;                // (n + 63) / 64 entries
    const uint16_t *prim;                // 256 << P entries, indexed ctx << P | first P stream bits LSB-first
    const uint16_t *sec;                 // uniform tables of 2^H entries, indexed LSB-first; table id = inner prim entry
    uint32_t P, H, nsec;
    int *status;
    uint32_t *redo;                      // [0] = count, [1..] chunk numbers handed to the redo pass (legacy tables)
    uint64_t ntiles;                     // full wave pieces (K tiles of 4096 symbols each)
    uint32_t probe;                      // timing probes (MH_TILE_PROBE): results are wrong and nothing is reported
    // order 2 (extension, parity unpinned): the LIVE contexts' tables, one slot each.  prim / sec then hold uint32
    // entries (low half as above, high half = slot of the context that follows the symbol), the first level has
    // nslots << P entries, index entries carry two context bytes (bits 48..63) and a fine entry is
    // context << 16 | bits from the chunk's index entry (0xFFFF: does not fit)
    uint32_t o2;
    uint32_t nslots;
    const uint16_t *ctx2slot;            // 65536 entries: slot of a two-byte context, 0xFFFF = none
};
size_t decode_tile_workspace_extra();    // bytes the tile decoder needs in front of the redo list (status block included)
hipError_t launch_decode_tile(TileParams p, const DecParams &legacy, void *d_ws, hipStream_t st);
// the redo pass of launch_decode alone (chunks listed in p.redo; legacy tables)
hipError_t launch_decode_redo(DecParams p, hipStream_t st);

struct IdxParams {
    int order;
    const uint8_t *payload;
    uint64_t payload_bytes;
    uint64_t nbits;
    uint32_t prev0;
    uint32_t chunk_shift;
    unsigned long long *index;
    uint64_t index_cap;
    unsigned long long *n_symbols;
    const uint16_t *prim;
    const uint16_t *sec;
    const uint32_t *sec_base;
    const uint32_t *tree;
    uint32_t P;
    uint32_t direct, H;
    int *status;
    // filled by launch_build_index from the workspace
    unsigned int *changed;
    unsigned long long *seg_end_state, *seg_used, *seg_sym_start;
    uint32_t *seg_count;
    uint32_t len_gcd;             // gcd of the model's code lengths (0/1: none)
    uint32_t max_len;             // the model's longest code (bits)
    uint32_t seg_bits;
    uint64_t nseg;
    uint32_t *fine;               // optional: fine index entries written by the fill passes (see TileParams)
    uint64_t fine_cap;            // entries available at `fine`
    // the fast path (mh_tile.hip, index_tile_kernel): the tile decoder's tables and, from the workspace, one compact
    // record per IX_SEG_BITS-bit segment — state = context << 8 | bits past the segment boundary
    const uint16_t *tprim, *tsec; // null: no tile tables, the fast path is not taken
    uint32_t tP, tH, tnsec;
    uint16_t *e16, *s16, *c16;    // end state, the state the segment was entered with, symbols that start in it
    uint32_t *tile_cnt;           // symbols per tile of 64 segments
    unsigned long long *tile_base;
    uint64_t nseg5, ntile5;
    uint32_t warm_bits;           // mode 0: bits in front of its segment a lane starts at (<= IX_WARM_BITS_MAX)
    uint32_t *dirty_list;         // segments to repair (index_tile_dirty_kernel); dirty_cap entries
    uint32_t dirty_cap;
    uint32_t iter;                // the pass counter (changed[iter]) the listed segments are counted in
};
// segments of the fast index path / of the segment decoder: 128 of them (two per lane) make a wave's tile; a lane that does not
// know its start state warms up over the bits in front of its segment (IdxParams::warm_bits).  A segment is an ODD number of
// dwords: the lanes of a wave stand at about the same offset in their segments, and with a stride of eight dwords (256 bits)
// their window reads met on 4 of the 32 LDS banks (measured: 81 % of the LDS cycles were bank conflicts, the LDS 72 % busy).
// [r5] eleven dwords (352 bits, 5.5 KiB per tile: sixteen waves' tiles still fit beside the 64 KiB first level) instead of
// nine: a states pass decodes warm-up + segment, and the longer segment spreads the warm-up over more symbols.
#ifndef MH_IX_SEG_DWORDS
#define MH_IX_SEG_DWORDS 11
#endif
static_assert(MH_IX_SEG_DWORDS % 2 == 1 && MH_IX_SEG_DWORDS >= 5 && MH_IX_SEG_DWORDS <= 11, "an odd number of dwords; sixteen tiles must fit 96 KiB");
constexpr uint32_t IX_SEG_BITS = 32u * MH_IX_SEG_DWORDS, IX_TILE_SEGS = 128, IX_TILE_BITS = IX_TILE_SEGS * IX_SEG_BITS, IX_WARM_BITS_MAX = 512;
constexpr uint16_t IX_INVALID = 0xFFFF;
// c16 (symbols that start in a segment): bits 0..14 the count, bit 15 = the segment holds a code longer than the tile tables'
// two levels resolve (set by the repair kernel; the segment decoder leaves such a segment to a walk) [r5]
constexpr uint32_t IX_C16_COUNT = 0x7FFFu, IX_C16_WALK = 0x8000u;
constexpr uint32_t IDX_MAX_PASSES = 96;                  // pass counters in the index builder's workspace (`changed`)
hipError_t launch_index_tile(const IdxParams &p, int mode, hipStream_t st);   // mode 0: states and counts, 1: the index entries
// [r5] the segment decoder: the second pass over a stream without an index emits the bytes itself (e16 / c16 / tile_base set)
hipError_t launch_segment_decode(const IdxParams &p, uint8_t *d_out, uint64_t out_cap, hipStream_t st);
// streams without an index in two passes: states + counts (synchronises between its passes), then the bytes
hipError_t launch_stream_states(IdxParams p, void *d_ws, hipStream_t st);
hipError_t launch_stream_emit(IdxParams p, void *d_ws, uint8_t *d_out, uint64_t out_cap, hipStream_t st);
// which encoder a launch_encode* call used (its workspace's status block, bytes 8..11)
// which decoder ran (status block of the decode workspace, bytes 40..43; mh_dev_decode_path)
enum { DEC_PATH_NONE = 0, DEC_PATH_TILE = 1, DEC_PATH_CHUNK = 2 };
// the chunk decoder's variants (mh_decode.hip, dec_cfg): which one launch_decode chose is noted in the status block (bytes 44..47,
// value + 1; mh_dev_decode_variant).  The redo pass runs behind every one of them with the variant of the same table layout.
enum DecVariant { DV_LDS_WIDE = 0, DV_LDS_SHORT = 1, DV_LDS_TWO_LEVEL = 2, DV_LDS_TWO_LEVEL_P8 = 3, DV_L2_DIRECT = 4, DV_L2_DIRECT_H2 = 5,
                  DV_L2_DIRECT_H3 = 6, DV_L2_DIRECT_H4 = 7, DV_L2_DIRECT_H8 = 8, DV_REDO_LDS = 9, DV_REDO_L2_DIRECT = 10 };
hipError_t launch_set_word(uint32_t *d_word, uint32_t v, hipStream_t st);
enum { ENC_PATH_NONE = 0, ENC_PATH_REGIONS = 1, ENC_PATH_LENGTH_PASS = 2, ENC_PATH_REGIONS_ESCAPES = 3, ENC_PATH_CHAIN = 4 };
// how launch_build_index arrived at the index (status block bytes 8..11)
enum { IDX_PATH_NONE = 0, IDX_PATH_SEGMENTS = 1, IDX_PATH_GROUP_MAPS = 2, IDX_PATH_STATE_MAPS = 3, IDX_PATH_WALK = 4, IDX_PATH_TILES = 5,
       IDX_PATH_STATES = 6 };   // 6: launch_stream_states left the segments' states for launch_stream_emit (no index was built)

// ---- device tree build (mh_tree.hip)
constexpr int TB_NODE_STRIDE = 520;   // >= 513 nodes per context
constexpr int TB_META_STRIDE = 16;    // per context: nnodes, root, max_len, #inner nodes at depth 8,
                                      // second-level sizes for P = 0..8 (table heights capped at 8), weight lo/hi,
                                      // bit mask of the code lengths in use (0 for a one-symbol context)
struct TreeBuildOut {
    uint8_t *len8; unsigned long long *code64;
    uint16_t *enc16; uint8_t *len_slot;      // LDS-table images of the order-1 encoder; nullptr for order 2
    uint16_t *node_left, *node_right; uint8_t *node_sym, *node_height;
    uint32_t *ctx_meta;
    uint32_t hcap;                // cap on second-level table heights in the size profile (8; order 2: 4)
};
struct TreePackArgs {
    const uint16_t *node_left, *node_right; const uint8_t *node_sym, *node_height;
    const uint32_t *ctx_meta;
    uint32_t *sec_base;           // 256: device copy of sec_base_val, written by the kernel (the decoders read it)
    uint32_t sec_base_val[256];   // entry offsets chosen by the host (order 0/1: travels in the kernel arguments)
    const uint32_t *sec_base_in;  // order 2: 65536 offsets in device memory (then sec_base_val is unused)
    uint32_t P, direct, H, hcap;
    uint16_t *prim, *sec; uint32_t *tree;
    uint32_t lsb;                 // 1: tables indexed by the window's bits LSB-first (first stream bit = bit 0): the
                                  // tile decoder's layout (mh_tile.hip); 0: MSB-first (first stream bit = top bit)
};
// order 2: tables of the live contexts (one slot each) for the LDS-resident encoder and the tile decoder
struct O2HotArgs {
    const uint16_t *slot_ctx;     // nslots: the context of each slot
    const uint8_t *slot_id1;      // nslots: id of each slot's second context byte (column permutation of the encoder rows)
    uint32_t nslots;
    uint8_t id_sym[64];           // byte value of id 0..62 (unused ids: any value with id_used 0)
    uint8_t id_used[64];
    const uint8_t *len8; const unsigned long long *code64;
    uint16_t *hot;                // (nslots + 1) * 64 u16: the encoder's rows (mh_encode.hip, o2hot_lookup16)
    // tile decoder tables (nullptr: none)
    const uint16_t *node_left, *node_right; const uint8_t *node_sym; const uint32_t *ctx_meta;
    const uint16_t *ctx2slot;
    uint32_t P, H;
    uint32_t *tprim, *tsec;       // nslots << P entries; (nslots << P) << H entries
};
hipError_t launch_o2_hot_pack(const O2HotArgs &a, hipStream_t st);
hipError_t launch_tree_build(const unsigned long long *d_counts, int nctx, const TreeBuildOut &o, hipStream_t st);
hipError_t launch_tree_pack(const TreePackArgs &a, int nctx, hipStream_t st);
hipError_t launch_tree_pack2(const TreePackArgs &a, const TreePackArgs &b, int nctx, hipStream_t st);   // both in one launch

size_t hist_workspace_bytes(uint64_t n);
hipError_t launch_hist_o1(const uint8_t *d_data, uint64_t n, uint32_t prev0, unsigned long long *d_counts, void *d_ws, size_t ws_bytes,
                          hipStream_t st);
hipError_t launch_hist_o0(const uint8_t *d_data, uint64_t n, unsigned long long *d_counts, hipStream_t st);
// order 2: d_ws (256-byte aligned, hist2_workspace_bytes(n)) lets sources with millions of live keys take the partition path;
// without it everything goes through the tag cache.  The workspace's first words afterwards: [0] status, [2] the choices made
// (bit 0: a slab stayed in the tag cache, bit 1: a slab was partitioned).
size_t hist2_workspace_bytes(uint64_t n);
hipError_t launch_hist_o2(const uint8_t *d_data, uint64_t n, uint32_t ctx0, unsigned long long *d_counts, void *d_ws, size_t ws_bytes,
                          hipStream_t st);
size_t encode_workspace_bytes(uint64_t n);
hipError_t launch_encode(const EncodeArgs &a, void *d_ws, hipStream_t st);
hipError_t launch_encode_regions(const EncodeArgs &a, const void *d_hist_ws, size_t hist_ws_bytes, void *d_ws, hipStream_t st);
hipError_t launch_payload_bits(const unsigned long long *d_counts, const uint8_t *d_len8, uint32_t entries, unsigned long long *d_out,
                               hipStream_t st);
// enc64[i] = len8[i] << 56 | code64[i] (len <= 56), else 0xFF << 56: one gather per symbol in the order-2 encoder
hipError_t launch_enc64_pack(const uint8_t *len8, const uint64_t *code64, uint64_t *enc64, uint64_t n, hipStream_t st);
hipError_t launch_decode(DecParams p, void *d_ws, hipStream_t st);
size_t build_index_workspace_bytes(uint64_t nbits);
hipError_t launch_build_index(IdxParams p, void *d_ws, hipStream_t st);

}  // namespace mhk
