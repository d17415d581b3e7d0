// mh_tile_probes.hpp — DIAGNOSTIC LIBRARY ONLY (libmhc_diag.so, -DMH_EXP_PROBES): other ways for decode_tile_kernel's symbol
// step to reach the second table level.  The shipped kernel (G = 0) has none of this; mh_tile.hip includes the file behind
// MH_EXP_PROBES and selects a form with MH_TILE_G.  Forms marked "output wrong" only measure what a step would cost.
// Results: profiles/r05/decode_probes_4GiB.txt, DESIGN.md 3.3 [r5].
//   1  no second level at all (output wrong): the floor of the loop
//   2  the gather issued by lanes 0..15 only (output wrong; with lanes 0..31, the first use of number 3: 4.51 ms, the same): does
//      the texture addresser's time follow the ACTIVE lanes of an instruction, or the instruction?
//   3  real (output right): the shipped gathers issued at raised wave priority (s_setprio 3 around them): does the order in which
//      the CU's sixteen waves get their vector-memory instructions out matter?
//   4  VERDICT r04 1(b), cost model (output wrong): the gather replaced by the two dependent LDS lookups an LDS-complete layout
//      needs for a deep symbol (subtree shape -> rank, then the context's list of deep symbols), 59 KiB of the LDS set aside for
//      those tables; 5 the same with the lanes that do not need them all reading one address (a broadcast, no bank conflict)
//   6  VERDICT r04 1(a), real (output right): ONE gather per step for both streams of a lane pair — stream B's requests are
//      compacted (ds_permute), handed to the lanes whose stream A entry is a leaf (ds_bpermute), and the answers travel back
//      the same way; a step with more than 64 requests in all takes two gathers
//   7  real (output right): stream B's request rides in its own lane wherever stream A does not need the gather; the lanes
//      that need it for both (9 %) issue a second, exec-masked one
// (included by mh_tile.hip INSIDE namespace mhk, behind its LDS helpers)
#pragma once

constexpr uint32_t T_PROBE_DEEP_BYTES = 59u * 1024u;
__host__ __device__ constexpr uint32_t tile_probe_reserve(int g) { return g == 4 || g == 5 ? T_PROBE_DEEP_BYTES : 0u; }

__device__ __forceinline__ uint32_t probe_gather16(const __amdgpu_buffer_rsrc_t &rsrc, uint32_t byte_off) {
    return uint32_t(uint16_t(__builtin_amdgcn_raw_buffer_load_b16(rsrc, int(byte_off), 0, 0)));
}

template <int K, int G, int PC>
__device__ __forceinline__ void tile_second_probe(const uint32_t (&e)[K], const uint32_t (&win)[K], const uint32_t (&cf)[K], uint32_t H,
                                                  const __amdgpu_buffer_rsrc_t &sec_rsrc, uint32_t lane, uint32_t (&e2)[K]) {
    constexpr uint32_t P = PC;
    using mh::DEC16_LEAF;
    uint32_t idx2[K];
#pragma unroll
    for (int k = 0; k < K; ++k) idx2[k] = (e[k] << (H + 1)) | ((win[k] >> (P - 1)) & ((2u << H) - 2u));
    if (G == 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) e2[k] = idx2[k] & 0u;
    } else if (G == 2) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            e2[k] = 0u;
            if (lane < 16u) e2[k] = probe_gather16(sec_rsrc, idx2[k]);
        }
    } else if (G == 4 || G == 5) {
        constexpr uint32_t BASE = (256u << P) * 2u;              // behind the first level: 4 KiB of "shapes", 55 KiB of "deep lists"
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool needy = !(e[k] & DEC16_LEAF);
            uint32_t a1 = BASE + (((e[k] & 0x7Fu) << 5) | ((win[k] >> (P - 1)) & 30u));
            if (G == 5) a1 = needy ? a1 : BASE;
            const uint32_t s = *lds_ptr<uint16_t>(a1);
            uint32_t a2 = BASE + 4096u + (cf[k] & 255u) * 216u + ((e[k] >> 7) & 0x7Fu) + (s & 0x3Fu);
            if (G == 5) a2 = needy ? a2 : BASE + 4096u;
            uint32_t v = *lds_ptr<uint8_t>(a2);
            asm volatile("v_and_b32 %0, 0, %0" : "+v"(v));      // (the value is not a table's: keep the dependence, drop the bits)
            e2[k] = v;
        }
    } else if (G == 6 && K == 2) {
        const bool nA = !(e[0] & DEC16_LEAF), nB = !(e[1] & DEC16_LEAF);
        const unsigned long long mA = __ballot(nA), mB = __ballot(nB);
        const uint32_t cA = uint32_t(__popcll(mA)), cB = uint32_t(__popcll(mB));
        if (cA + cB <= 64u) {                                    // (wave-uniform)
            const uint32_t rA = __builtin_amdgcn_mbcnt_hi(uint32_t(mA >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mA), 0u));
            const uint32_t rB = __builtin_amdgcn_mbcnt_hi(uint32_t(mB >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mB), 0u));
            // B's requests in rank order in the low lanes; the other lanes' words (leaf indices: past the end of the table) behind them
            const uint32_t d1 = nB ? rB : cB + lane - rB;
            const uint32_t c1 = uint32_t(__builtin_amdgcn_ds_permute(int(d1 << 2), int(idx2[1])));
            // the r-th lane whose stream A entry is a leaf asks for B's r-th request
            const uint32_t rNA = lane - rA;
            const uint32_t t = uint32_t(__builtin_amdgcn_ds_bpermute(int(rNA << 2), int(c1)));
            const uint32_t g = probe_gather16(sec_rsrc, nA ? idx2[0] : t);
            // and back: answers in rank order, B's lanes take theirs
            const uint32_t d3 = nA ? 64u - cA + rA : rNA;
            const uint32_t c3 = uint32_t(__builtin_amdgcn_ds_permute(int(d3 << 2), int(g)));
            const uint32_t t2 = uint32_t(__builtin_amdgcn_ds_bpermute(int(rB << 2), int(c3)));
            e2[0] = nA ? g : 0u;
            e2[1] = nB ? t2 : 0u;
        } else {
            e2[0] = probe_gather16(sec_rsrc, idx2[0]);
            e2[1] = probe_gather16(sec_rsrc, idx2[1]);
        }
    } else if (G == 7 && K == 2) {
        const bool nA = !(e[0] & DEC16_LEAF), nB = !(e[1] & DEC16_LEAF);
        const uint32_t g1 = probe_gather16(sec_rsrc, nA ? idx2[0] : idx2[1]);
        uint32_t g2 = 0u;
        if (nA && nB) g2 = probe_gather16(sec_rsrc, idx2[1]);
        e2[0] = nA ? g1 : 0u;
        e2[1] = nA ? g2 : g1;
    } else if (G == 3) {
        // MH_PRIO_MODE (compile time, make exp): 0 = priority 3 from before the gathers until their results are used (the first
        // form measured: 40.8 ms against 4.45 — sixteen waves that all wait at top priority take turns); 1 = priority 3 for the
        // ISSUE of the gathers only; 2 = priority 0 for the issue of the gathers, 2 for the rest of the step (the phases on the
        // dependent chain: resolve, window, first level); 3 = the same with 1
#ifndef MH_PRIO_MODE
#define MH_PRIO_MODE 0
#endif
        uint32_t raw[K];
        if (MH_PRIO_MODE == 0 || MH_PRIO_MODE == 1) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int k = 0; k < K; ++k) raw[k] = uint32_t(__builtin_amdgcn_raw_buffer_load_b16(sec_rsrc, int(idx2[k]), 0, 0));
        if (MH_PRIO_MODE == 1) __builtin_amdgcn_s_setprio(0);
        if (MH_PRIO_MODE == 2) __builtin_amdgcn_s_setprio(2);
        if (MH_PRIO_MODE == 3) __builtin_amdgcn_s_setprio(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < K; ++k) e2[k] = raw[k] & 0xFFFFu;
        if (MH_PRIO_MODE == 0) __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) e2[k] = probe_gather16(sec_rsrc, idx2[k]);
    }
}

