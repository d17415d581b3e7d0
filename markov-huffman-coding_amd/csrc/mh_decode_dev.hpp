// mh_decode_dev.hpp — the bit source, the two-level table lookup and the granule FIFO of the chunk decoder: shared by
// mh_decode.hip (decode_kernel, decode2_kernel) and mh_index.hip (the segment walks of the index builder).
#pragma once
#include "mh_dev.hpp"

namespace mhk {

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------
// Stream words are big-endian in stream order: word w holds stream bits [32w, 32w+32), first bit in
// bit 31.  Reads past the last payload byte return zero bits (src/bitbuffer.cpp:116-127).
struct BitSrc {
    const uint8_t *p;
    uint64_t full_words;   // payload_bytes / 4
    uint64_t bytes;
    __device__ __forceinline__ uint32_t word(uint64_t w) const {
        if (w < full_words) return __builtin_bswap32(reinterpret_cast<const uint32_t *>(p)[w]);
        uint32_t v = 0;
        for (uint32_t i = 0; i < 4; ++i) {
            uint64_t b = (w << 2) + i;
            if (b < bytes) v |= uint32_t(p[b]) << (24u - 8u * i);
        }
        return v;
    }
};

// 64-bit window + one prefetched word: the load for the NEXT refill is always in flight, so a refill
// never waits on memory.
struct BitCursor {
    uint64_t buf;     // next bits, first at bit 63
    uint32_t cnt;     // valid bits in buf
    uint32_t ahead;   // stream word `next - 1`, already loaded
    uint64_t next;    // next word index to fetch
    __device__ __forceinline__ void init(const BitSrc &src, uint64_t bitpos) {
        uint64_t w = bitpos >> 5;
        uint32_t sh = uint32_t(bitpos & 31u);
        buf = ((uint64_t(src.word(w)) << 32) | src.word(w + 1)) << sh;
        cnt = 64u - sh;
        ahead = src.word(w + 2);
        next = w + 3;
    }
    // afterwards cnt >= 33
    __device__ __forceinline__ void refill(const BitSrc &src) {
        if (cnt <= 32u) {
            buf |= uint64_t(ahead) << (32u - cnt);
            cnt += 32u;
            ahead = src.word(next++);
        }
    }
    __device__ __forceinline__ void drop(uint32_t n) { buf <<= n; cnt -= n; }
    __device__ __forceinline__ uint64_t window() const { return buf; }
};

struct DecTables {
    const uint16_t *sec;         // second-level tables: LDS copy (decode_kernel) or global (index builder)
    const uint32_t *tree;        // last-resort walk (HBM/L2)
    uint32_t P;                  // primary width in bits
    uint32_t direct, H;          // uniform L2 tables: inner entry = table id, 2^H entries each
    __amdgpu_buffer_rsrc_t sec_rsrc;   // L2-resident second level: buffer resource over `sec` (a gather then takes a 32-bit
                                       // byte offset instead of a 64-bit address)
};

// Decodes one symbol (sequential index builder; tables read from global memory).  Returns the symbol,
// or 0 with *bad set on a null table entry (corrupt stream / context missing from the table).
// *used accumulates the bits consumed.
template <typename CUR>
__device__ __forceinline__ uint32_t decode_one(const uint16_t *prim, const uint32_t *sec_base, const DecTables &t,
                                               const BitSrc &src, CUR &bc, uint32_t prev, uint32_t &used, bool &bad) {
    bc.refill(src);                                        // >= 33 bits: enough for P + 8
    uint32_t e = prim[(prev << t.P) | uint32_t(bc.window() >> (64u - t.P))];
    if (e & DEC16_LEAF) {                                  // code of <= P bits (src/coding.cpp:150-156)
        uint32_t len = (e >> 8) & 31u;
        bad |= (len == 0);
        bc.drop(len); used += len;
        return e & 255u;
    }
    // longer code (src/coding.cpp:129-149): the node's own table, indexed by the next h bits
    const uint32_t h = t.direct ? t.H : ((e >> 12) & 7u) + 1u;
    bc.drop(t.P);
    const uint32_t tbase = t.direct ? (e << t.H) : sec_base[prev] + (e & 0xFFFu);
    uint32_t e2 = t.sec[tbase + uint32_t(bc.window() >> (64u - h))];
    if (e2 & DEC16_LEAF) {
        uint32_t len = (e2 >> 8) & 31u;                     // total length, P included
        bad |= (len == 0);
        if (len) { bc.drop(len - t.P); used += len; }
        return e2 & 255u;
    }
    // longer than P + h: walk the context's tree bit by bit from that node
    bc.drop(h);
    uint32_t node = e2 & 0x1FFu, n = t.P + h;
    const uint32_t *tr = t.tree + prev * TREE_STRIDE;
    for (int guard = 0; guard < 256; ++guard) {
        bc.refill(src);
        uint32_t bit = uint32_t(bc.window() >> 63);
        bc.drop(1); ++n;
        uint32_t pair = tr[node];
        uint32_t c = bit ? (pair >> 16) : (pair & 0xFFFFu);
        if (c & TREE_LEAF) { used += n; return c & 255u; }
        node = c;
    }
    bad = true;
    used += n;
    return 0;
}

// ---- the hot decoder ---------------------------------------------------------------------------
// One workgroup per CU, one lane per chunk, consecutive lanes on consecutive chunks.  Two measured
// facts shape the input side:
//  (1) a lane's compressed bytes are ~0.7 KiB away from its neighbour's, so a per-lane dword read costs
//      a whole cache line, and with ~1000 streams per CU the lines do not survive in L2 between two
//      reads (16x read amplification, L2 hit rate 32 %).  Each lane therefore pulls its stream in
//      32-byte aligned granules (two 16-byte loads): every byte is fetched once.
//  (2) vmcnt retires in order.  A load issued by SOME lane in a round sits in front of that round's
//      table gather for the WHOLE wave, so per-lane "refill when empty" loads put an HBM latency into
//      every round.  Loads are therefore issued only at block boundaries (every 16 symbols, all lanes
//      together; in the L2 layout right BEHIND the first step's gathers) and consumed a block later:
//      `nxt` (and with DEPTH 2 `pre`) is in flight or resident, `cur` feeds the bit window; inside a
//      block a granule switch is register-to-register.
// A block of 16 symbols decoded through the tables consumes at most 16 * 16 = 256 bits = one granule,
// and a block starts with `nxt` full, so the hot path never runs dry; longer codes (the walk) and the
// set-up use the checked pop.
// GW = dwords per granule: 8 (32 bytes) or 16 (64 bytes: half the read amplification, twice the
// registers).  A block is 2 * GW symbols (<= 16 bits each through the tables = one granule).
// DEPTH = granules a stream keeps beside `cur`: 2 (`nxt` resident + `pre` in flight: a granule is consumed
// one block after it was asked for, so the block-boundary loads never make a pop wait) or 1 (`nxt` alone,
// loaded straight into: eight registers per stream fewer; the switch cur <- nxt may then wait for a load
// that was issued at the last block boundary).  Measured at 16 GiB Zipf (profiles/r02/decode_variants.md):
// DEPTH 1 with 4, 5 or 6 streams per lane and 3 or 4 waves per SIMD all land within 2 % of, or behind,
// DEPTH 2 with 4 streams and 2 waves — the kernel is not short of streams in flight.  The L2 layout runs
// DEPTH 1: the eight registers per stream hold a 64-byte store burst instead (25.3 vs 29.4 ms).
template <int GW, int DEPTH = 2>
struct LaneStream {
    static constexpr int NQ = GW / 4;      // uint4 loads per granule
    static constexpr uint32_t BEHIND = DEPTH + 1;   // `cur` holds granule gnext - BEHIND while nxt is full
    const uint4 *base;    // payload (wave-uniform: lives in SGPRs)
    uint32_t glast;       // last readable granule of the payload (wave-uniform)
    uint32_t gnext;       // granule (counted from the payload start) that is loaded next
    uint32_t cur[GW];     // granule feeding the window, cur[0] is next
    uint32_t ccnt;        // dwords left in cur
    uint32_t nxt[GW];     // following granule (DEPTH 1: possibly still in flight)
    bool nxt_full;
    uint32_t pre[DEPTH == 2 ? GW : 1];     // DEPTH 2: the one after, possibly still in flight

    uint64_t buf;         // next bits, first at bit 63
    uint32_t cnt;         // valid bits in buf

    __device__ __forceinline__ void issue_into(uint32_t (&dst)[GW]) {
        const uint32_t g = gnext < glast ? gnext : glast;
        const uint4 *src = base + uint64_t(NQ) * g;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint4 a = src[q];
            dst[4 * q] = a.x; dst[4 * q + 1] = a.y; dst[4 * q + 2] = a.z; dst[4 * q + 3] = a.w;
        }
        ++gnext;
    }
    __device__ __forceinline__ void issue_pre() {
        if constexpr (DEPTH == 2) issue_into(pre); else issue_into(nxt);
    }
    // wave-synchronous point (block boundary): the only place where loads are issued and awaited
    __device__ __forceinline__ void block_sync() {
        if (!nxt_full) {
            if constexpr (DEPTH == 2) {
#pragma unroll
                for (int i = 0; i < GW; ++i) nxt[i] = pre[i];
            }
            nxt_full = true;
            issue_pre();
        }
    }
    // CHECKED (set-up, walk of over-long codes, tail chunks): keeps `nxt` full around every pop, so any
    // amount may be consumed.  Unchecked (hot path): relies on the per-block budget above.
    // The granule is consumed a quad at a time: 3 moves per pop, GW - 4 more every fourth pop.
    template <bool CHECKED>
    __device__ __forceinline__ uint32_t pop_word() {
        if (CHECKED) block_sync();
        const uint32_t w = cur[0];
        if (GW == 8) {
            // short granule: shifting all of it costs less than a second level of bookkeeping
#pragma unroll
            for (int i = 0; i < GW - 1; ++i) cur[i] = cur[i + 1];
            if (--ccnt == 0) {
#pragma unroll
                for (int i = 0; i < GW; ++i) cur[i] = nxt[i];
                nxt_full = false;
                ccnt = GW;
                if (CHECKED) block_sync();
            }
            return w;
        }

        cur[0] = cur[1]; cur[1] = cur[2]; cur[2] = cur[3];
        --ccnt;
        if ((ccnt & 3u) == 0u) {
            if (ccnt == 0) {
#pragma unroll
                for (int i = 0; i < GW; ++i) cur[i] = nxt[i];
                ccnt = GW;
                nxt_full = false;
                if (CHECKED) block_sync();
            } else {
#pragma unroll
                for (int i = 0; i < GW - 4; ++i) cur[i] = cur[i + 4];
            }
        }
        return w;
    }
    // total_bytes > 0 and bitpos < 8 * total_bytes (checked by the caller)
    __device__ __forceinline__ void init(const uint8_t *payload, uint64_t total_bytes, uint64_t bitpos) {
        const uint64_t w = bitpos >> 5;                      // first stream dword
        base = reinterpret_cast<const uint4 *>(payload);
        glast = uint32_t((total_bytes - 1) / (GW * 4));       // payloads stay below 2^32 granules (128 GiB)
        gnext = uint32_t(w / GW);
        if constexpr (DEPTH == 2) {
            issue_into(pre);
#pragma unroll
            for (int i = 0; i < GW; ++i) cur[i] = pre[i];
            ccnt = GW;
            issue_into(pre);
#pragma unroll
            for (int i = 0; i < GW; ++i) nxt[i] = pre[i];
            nxt_full = true;
            issue_into(pre);
        } else {
            issue_into(nxt);
#pragma unroll
            for (int i = 0; i < GW; ++i) cur[i] = nxt[i];
            ccnt = GW;
            issue_into(nxt);
            nxt_full = true;
        }
        for (uint32_t skip = uint32_t(w % GW); skip; --skip) (void)pop_word<true>();
        const uint32_t hi = __builtin_bswap32(pop_word<true>());
        const uint32_t lo = __builtin_bswap32(pop_word<true>());
        const uint32_t sh = uint32_t(bitpos & 31u);
        buf = ((uint64_t(hi) << 32) | lo) << sh;
        cnt = 64u - sh;
    }
    template <bool CHECKED>
    __device__ __forceinline__ void refill() {               // afterwards cnt >= 33
        if (cnt <= 32u) {
            buf |= uint64_t(__builtin_bswap32(pop_word<CHECKED>())) << (32u - cnt);
            cnt += 32u;
        }
    }
    // (The same refill as selects instead of masked regions — some lane needs the pop at practically every
    //  refill point, so the regions run anyway — takes 21 % of the loop's instructions and all but 6 of
    //  its 52 branches away and measured SLOWER: 26.6 vs 25.7 ms.  The kernel is not bound by what it issues.)
    // position of the next unread bit in the payload, modulo 2^32: `cur` holds granule gnext - BEHIND
    // (one more while `nxt` is empty), GW - ccnt of its dwords have gone into the window, cnt bits of
    // the window are still unread
    __device__ __forceinline__ uint32_t position() const {
        const uint32_t gran = gnext - (nxt_full ? BEHIND : BEHIND - 1u);
        return (gran * GW + (GW - ccnt)) * 32u - cnt;
    }
};

// BitCursor's face over the 32-byte-granule FIFO of the order-1 decoder: a lane's payload arrives in aligned
// 32-byte pieces (a dword at a time costs a 128-byte line per load once the lanes' streams are a chunk apart)
struct GranuleCursor {
    LaneStream<8, 2> ls;
    __device__ __forceinline__ void init(const BitSrc &src, uint64_t bitpos) { ls.init(src.p, src.bytes, bitpos); }
    __device__ __forceinline__ void refill(const BitSrc &) { ls.template refill<true>(); }
    __device__ __forceinline__ void drop(uint32_t n) { ls.buf <<= n; ls.cnt -= n; }
    __device__ __forceinline__ uint64_t window() const { return ls.buf; }
};


}  // namespace mhk
