// fcm_kernels_common.hpp — gfx950 (MI355X, CDNA4) device code for the edge-flip
// MCMC hot path of flag-complex-mcmc.  64-wide wavefronts throughout; integer
// bitset work only (no MFMA).
//
//   fcm_step_kernel   one persistent 64-lane workgroup per chain.  Runs
//                     `nprop` iterations of the reference loop
//                     MCMCSampler::next (src/lib.rs:182-192): propose
//                     (src/lib.rs:292-325), count the change
//                     (State::apply_transition, :61-79), integer bounds check
//                     (Bounds::check, :157-160), commit or drop (:187-191).
//   fcm_count_kernel  flagser_count (src/lib.rs:51,130; src/flagser.rs:9):
//                     one wave per directed edge, simplices that start with
//                     that edge.  (fcm_count.hip)
//
// How a proposal is counted.  The reference recounts the whole induced
// subgraph on N(a) cap N(b) + {a,b} before and after (src/lib.rs:63,71); only
// post - pre matters.  Every simplex that differs contains
// the changed directed edge, so the kernel counts exactly those:
//   E(G, u->v)[d] = #d-simplices of G that contain the edge u->v.
// Removing an edge subtracts E before removal, adding one adds E after.
// All vertices of such a simplex lie in L = N(u) cap N(v) + {u,v} (static,
// src/lib.rs:330).  The wave builds the induced out-adjacency of L as bit
// masks over local indices (lane j tests bit L[j] of row L[i]; the v_cmp
// result *is* the ballot), stages the masks in LDS, classifies each w in L by
// where it can sit relative to u->v (P: w->u,w->v  M: u->w,w->v  S: u->w,v->w)
// and walks the simplices by mask intersection with popcounts at the leaves.
//
// Two evaluators share that scheme:
//   fast  |L| <= 64: one 64-bit mask per local vertex; per-lane DFS in
//         registers (lane = first vertex), depth unrolled at compile time.
//   wide  |L| <= 256 (rare: a handful of edges in the BASELINE graphs):
//         masks of NW <= 4 words; one wave-uniform DFS with its stack in LDS.
//         Register-light on purpose, so that it costs the fast path nothing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "fcm_device.hpp"

typedef unsigned long long u64;
typedef unsigned int u32;

#define WAVE 64

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Counter = (step_lo, step_hi, chain,
// sub), key = seed.  Product copy; the oracle has its own.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1, u32 (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const u32 hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const u32 hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const u32 n0 = hi1 ^ c1 ^ k0;
        const u32 n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ u32 rdlane(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ u64 rdlane64(u64 v, int l)
{
    return (u64)rdlane((u32)v, l) | ((u64)rdlane((u32)(v >> 32), l) << 32);
}
__device__ __forceinline__ u64 ballot(bool p) { return __ballot(p); }
// v_writelane_b32: lane `l` of `old` := the wave-uniform value `src`
__device__ __forceinline__ u32 wrlane(u32 src, int l, u32 old)
{
#if __has_builtin(__builtin_amdgcn_writelane)
    return (u32)__builtin_amdgcn_writelane((int)src, l, (int)old);
#else
    // two SGPR sources would break the constant-bus limit: the lane select goes through m0
    // (the value is made opaque first: a constant folded into the "s" operand may come out as a literal the instruction does not take)
    asm volatile("" : "+s"(src));
    asm volatile("s_nop 0\n\ts_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(src), "s"(l) : "m0");
    return old;
#endif
}

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// Sum of a 32-bit value over the wave with DPP adds (no LDS): prefix sums
// inside each row of 16 lanes (row_shr 1,2,4,8), then row_bcast:15 / :31 carry
// the row totals upward.  The total lands in lane 63 and is returned uniform.
// Whether the per-lane counts and this sum stay below 2^31 is checked per evaluation where it is not a
// matter of course (fcm_count_guard below).
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}
// the same steps, keeping every lane's inclusive prefix sum
__device__ __forceinline__ int wave_scan_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// The workgroup is ONE wave.  Vector-memory and LDS instructions of a wave are
// issued and performed in program order, so data one lane wrote is visible to
// every lane's later loads without waiting on counters or a barrier (the
// AMDGPU memory model needs no cache action or wait at wavefront scope).  What
// is needed is that the compiler keeps the order: a wavefront-scope fence plus a
// scheduling barrier.  (A __syncthreads() here would add s_waitcnt vmcnt(0) +
// s_barrier, i.e. a full store round trip, a dozen times per proposal.)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------
// Diagnostic build only (-DFCM_STAMP): s_memtime stamps that drain every
// counter first, accumulated per phase into the debug words of the stats row.
// Never compiled into the product build; a stamped build's run time means
// nothing, only the shares do (cdna_hip_programming.md, In-kernel stamps).
// ---------------------------------------------------------------------------
#ifdef FCM_STAMP
__device__ __forceinline__ u64 fcm_stamp()
{
    u64 t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define FCM_STAMP_DECL u64 stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u64 stamp_t = fcm_stamp();
#define FCM_STAMP_AT(slot) do { const u64 _n = fcm_stamp(); stamp_acc[slot] += _n - stamp_t; stamp_t = _n; } while (0)
#define FCM_STAMP_PTR(slot) do { if (sacc) { const u64 _n = fcm_stamp(); sacc[slot] += _n - *stt; *stt = _n; } } while (0)
#else
#define FCM_STAMP_DECL
#define FCM_STAMP_AT(slot) do { } while (0)
#define FCM_STAMP_PTR(slot) do { } while (0)
#endif

// ===========================================================================
// Fast evaluator: local sets of <= 64 vertices, one u64 mask per vertex
// ===========================================================================

// Induced adjacency of the local vertex set, as IN-masks.  Lane j (< s) holds
// local vertex Lv = L[j] and gets the mask { i : L[i] -> L[j] } over local
// indices 0..s-1.  One dword per lane per row: the one with bit L[j] of row
// L[i]; the lane keeps its own bit, so a row costs v_readlane + buffer_load +
// v_bfe + v_lshl_or and no cross-lane traffic.  (Out-masks would need a ballot
// and two v_writelane per row.)  Rows are 128-B multiples, so one row-read is
// one or few cache lines, shared by the 64 lanes.
//
// The evaluator below is written for out-masks; it is run on these in-masks,
// i.e. on the transposed graph, where the simplices through u->v are the
// simplices through v->u (same sets, orders reversed, classes P and S
// swapped).  Callers therefore hand classify() the endpoints the other way
// round, and read "x -> y is present" as bit x of the mask of y.
#ifndef FCM_BUILD_GROUP4
#define FCM_BUILD_GROUP4 1
#endif
#ifndef FCM_HB
#define FCM_HB 24  // rows in flight per batch (16: -2%, 32: -4% on config 3)
#endif
// Buffer descriptor over one chain's bitmap (or the counter's graph): raw
// buffer, 32-bit offsets, out-of-range reads return 0.  `bytes` < 4 GiB is
// checked on the host.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rows_rsrc(const u32 *rows, u64 bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)rows, 0, (int)(u32)bytes, 0x00020000);
}

// rows I0 .. I0+N-1 of the local adjacency (those below s when GUARDED): issue the reads ...
template <int I0, int N, bool GUARDED>
__device__ __forceinline__ void build_issue(const rsrc_t rsrc, u32 voff, u32 roff, int s, u32 (&w)[N])
{
#if FCM_BUILD_GROUP4
    if constexpr (!GUARDED && N % 4 == 0) {
        // four row offsets first, then the four loads: a load whose soffset was written by v_readlane right before it waits five
        // cycles for the SGPR (s_nop 4 per row); behind three more v_readlane it does not (round 4: default mix + 1 %.  The same
        // for the guarded tiers, a test per group of four, spills vector registers and is 10 % slower: not done)
#pragma unroll
        for (int g = 0; g < N; g += 4) {
            u32 o0 = rdlane(roff, I0 + g), o1 = rdlane(roff, I0 + g + 1), o2 = rdlane(roff, I0 + g + 2), o3 = rdlane(roff, I0 + g + 3);
            asm volatile("" : "+s"(o0), "+s"(o1), "+s"(o2), "+s"(o3));
            w[g] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, o0, 0);
            w[g + 1] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, o1, 0);
            w[g + 2] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, o2, 0);
            w[g + 3] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, o3, 0);
        }
        return;
    }
#endif
#pragma unroll
    for (int q = 0; q < N; ++q) {
        w[q] = 0u;
        if (!GUARDED || I0 + q < s)  // buffer_load_dword v, voff, rsrc, soffset: no address arithmetic
            w[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, rdlane(roff, I0 + q), 0);
    }
}
// h |= (bit bpos of x) << POS: v_bfe, v_lshl_or (written out: hipcc would make it v_bfe, v_lshl and half a v_or3)
template <int POS>
__device__ __forceinline__ void build_put(u32 &h, u32 x, u32 bpos)
{
    const u32 b = __builtin_amdgcn_ubfe(x, bpos, 1u);
    asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(h) : "v"(b), "n"(POS));
}
template <int Q, int N>
__device__ __forceinline__ void build_put_all(u32 &h, const u32 (&w)[N], u32 bpos)
{
    if constexpr (Q < N) { build_put<Q>(h, w[Q], bpos); build_put_all<Q + 1, N>(h, w, bpos); }
}
// ... and keep this lane's bit of each: bit I0+q of the mask.  A slot that was not read holds 0.
template <int I0, int N, int Q = 0>
__device__ __forceinline__ void build_consume(const u32 (&w)[N], u32 bpos, u32 &hlo, u32 &hhi)
{
    if constexpr (Q < N) {
        constexpr int R = I0 + Q;
        if constexpr (R < 32) build_put<R>(hlo, w[Q], bpos); else build_put<R - 32>(hhi, w[Q], bpos);
        build_consume<I0, N, Q + 1>(w, bpos, hlo, hhi);
    }
}

// A guarded batch of N slots costs a compare and a branch per slot at issue, so it is
// instantiated in steps of four and the smallest size that holds `cnt` rows is used.
template <int I0, int N>
__device__ __forceinline__ void build_tail(const rsrc_t rsrc, u32 voff, u32 bpos, u32 roff, int s, u32 &hlo, u32 &hhi)
{
    u32 w[N];
    build_issue<I0, N, true>(rsrc, voff, roff, s, w);
    build_consume<I0, N>(w, bpos, hlo, hhi);
}
// the same with the unguarded batch `wa` (rows 0..FCM_HB-1) consumed between issue and consume
// of the tail, so that all reads are in flight together
template <int N>
__device__ __forceinline__ void build_second(const rsrc_t rsrc, u32 voff, u32 bpos, u32 roff, int s, const u32 (&wa)[FCM_HB],
                                             u32 &hlo, u32 &hhi)
{
    u32 wb[N];
    build_issue<FCM_HB, N, true>(rsrc, voff, roff, s, wb);
    build_consume<0, FCM_HB>(wa, bpos, hlo, hhi);
    build_consume<FCM_HB, N>(wb, bpos, hlo, hhi);
}

// Every row index below is a compile-time constant (a local set has at most 64 rows).
__device__ __forceinline__ u64 build_local(const rsrc_t rsrc, u32 stride32, u32 Lv, int s, int lane)
{
    static_assert(FCM_HB == 24, "the tiers below are written for 24-row batches");
    const bool act = lane < s;
    const u32 voff = act ? (Lv >> 5) * 4u : 0u;          // byte offset of this lane's dword inside a row
    const u32 bpos = Lv & 31u;                           // this lane's bit inside that dword
    const u32 roff = Lv * (stride32 * 4u);               // byte offset of row Lv; read back per row by v_readlane
    u32 hlo = 0u, hhi = 0u;
    if (s <= 24) {                                       // one guarded batch
        if (s <= 4) build_tail<0, 4>(rsrc, voff, bpos, roff, s, hlo, hhi);
        else if (s <= 8) build_tail<0, 8>(rsrc, voff, bpos, roff, s, hlo, hhi);
        else if (s <= 12) build_tail<0, 12>(rsrc, voff, bpos, roff, s, hlo, hhi);
        else if (s <= 16) build_tail<0, 16>(rsrc, voff, bpos, roff, s, hlo, hhi);
        else if (s <= 20) build_tail<0, 20>(rsrc, voff, bpos, roff, s, hlo, hhi);
        else build_tail<0, 24>(rsrc, voff, bpos, roff, s, hlo, hhi);
    } else {
        // 24 unguarded rows plus a guarded tail, and the reads of BOTH are issued before either is
        // consumed: a local set of up to 48 vertices costs one memory round trip, not two.
        u32 wa[FCM_HB];
        build_issue<0, FCM_HB, false>(rsrc, voff, roff, s, wa);
        if (s <= 28) build_second<4>(rsrc, voff, bpos, roff, s, wa, hlo, hhi);
        else if (s <= 32) build_second<8>(rsrc, voff, bpos, roff, s, wa, hlo, hhi);
        else if (s <= 36) build_second<12>(rsrc, voff, bpos, roff, s, wa, hlo, hhi);
        else if (s <= 40) build_second<16>(rsrc, voff, bpos, roff, s, wa, hlo, hhi);
        else if (s <= 44) build_second<20>(rsrc, voff, bpos, roff, s, wa, hlo, hhi);
        else {
            build_second<24>(rsrc, voff, bpos, roff, s, wa, hlo, hhi);
            if (s > 48) {                                // rows 48..63
                if (s <= 56) build_tail<48, 8>(rsrc, voff, bpos, roff, s, hlo, hhi);
                else build_tail<48, 16>(rsrc, voff, bpos, roff, s, hlo, hhi);
            }
        }
    }
    return act ? ((u64)hlo | ((u64)hhi << 32)) : 0ull;   // lanes >= s read dword 0 of every row: discard
}

// The same as a loop over batches of 16 rows (small code, 16 registers; one round trip per batch): the
// producer wave of the two-wave kernel on graphs whose rows are longer than a cache line.
__device__ __forceinline__ u64 build_local_loop16(const rsrc_t rsrc, u32 stride32, u32 Lv, int s, int lane)
{
    const bool act = lane < s;
    const u32 voff = act ? (Lv >> 5) * 4u : 0u;
    const u32 bpos = Lv & 31u;
    const u32 roff = Lv * (stride32 * 4u);
    u32 hlo = 0u, hhi = 0u;
#pragma nounroll
    for (int i0 = 0; i0 < s; i0 += 16) {
        u32 w[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            w[q] = 0u;
            if (i0 + q < s) w[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, rdlane(roff, i0 + q), 0);
        }
        u32 part = 0u;   // bits i0 .. i0+15 of the mask, at positions 0..15
        build_put_all<0, 16>(part, w, bpos);
        if (i0 < 32) hlo |= part << i0; else hhi |= part << (i0 - 32);
    }
    return act ? ((u64)hlo | ((u64)hhi << 32)) : 0ull;
}

// n <= 1024: a row is 32 dwords, one cache line.  One load instruction then fetches two WHOLE rows
// (lane l: dword l & 31 of row L[2q + (l >> 5)], the vertex fetched from lane 2q + (l >> 5) by
// ds_bpermute), so a local set needs half as many loads and registers in flight as rows; the lane's bit of row i sits in lane (L[j] >> 5) + 32 (i & 1) of
// that register and comes over with ds_bpermute (the LDS crossbar, no LDS memory).  Used by the
// producer wave of the two-wave kernel, which has 64 VGPRs.
// groups of four row pairs (eight rows) G0 .. G1-1, registers w[0 ..]: issue ...
// (`sel` is opaque to the compiler, so that sel + 8q stays an addition and folds into ds_bpermute's offset field)
template <int G, int G0, int G1, int NW>
__device__ __forceinline__ void build128_issue(const rsrc_t rsrc, u32 Lv, u32 sel, u32 dw, int s, u32 (&w)[NW])
{
    if constexpr (G < G1) {
        if (8 * G < s) {
#pragma unroll
            for (int q = 4 * G; q < 4 * G + 4; ++q) {
                // the vertex whose row this half-wave reads: L[2q] (lanes 0..31), L[2q+1] (lanes 32..63)
                const u32 v = (u32)__builtin_amdgcn_ds_bpermute((int)(sel + 8u * q), (int)Lv);
                w[q - 4 * G0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (v << 7) + dw, 0, 0);
            }
            build128_issue<G + 1, G0, G1, NW>(rsrc, Lv, sel, dw, s, w);
        }
    }
}
// ... and consume
template <int Q, int Q0, int NW>
__device__ __forceinline__ void build128_pair(const u32 (&w)[NW], u32 src, u32 bpos, u32 &hlo, u32 &hhi)
{
    const u32 x0 = (u32)__builtin_amdgcn_ds_bpermute((int)src, (int)w[Q - Q0]);
    const u32 x1 = (u32)__builtin_amdgcn_ds_bpermute((int)(src + 128u), (int)w[Q - Q0]);
    if constexpr (Q < 16) { build_put<2 * Q>(hlo, x0, bpos); build_put<2 * Q + 1>(hlo, x1, bpos); }
    else { build_put<2 * Q - 32>(hhi, x0, bpos); build_put<2 * Q + 1 - 32>(hhi, x1, bpos); }
}
template <int G, int G0, int G1, int NW>
__device__ __forceinline__ void build128_consume(const u32 (&w)[NW], u32 src, u32 bpos, int s, u32 &hlo, u32 &hhi)
{
    if constexpr (G < G1) {
        if (8 * G < s) {
            build128_pair<4 * G, 4 * G0, NW>(w, src, bpos, hlo, hhi);
            build128_pair<4 * G + 1, 4 * G0, NW>(w, src, bpos, hlo, hhi);
            build128_pair<4 * G + 2, 4 * G0, NW>(w, src, bpos, hlo, hhi);
            build128_pair<4 * G + 3, 4 * G0, NW>(w, src, bpos, hlo, hhi);
            build128_consume<G + 1, G0, G1, NW>(w, src, bpos, s, hlo, hhi);
        }
    }
}
// (the registers of groups that are not read are never looked at; declaring them written by an empty asm statement
// keeps hipcc from zeroing them on every path that skips their reads: 60 v_mov per build of 40 rows)
template <int N>
__device__ __forceinline__ void build_regs_any(u32 (&w)[N])
{
#pragma unroll
    for (int q = 0; q < N; ++q) asm volatile("" : "=v"(w[q]));
}
__device__ __forceinline__ u64 build_local_rows128(const rsrc_t rsrc, u32 Lv, int s, int lane)
{
    u32 sel = lane >= 32 ? 4u : 0u;
    asm volatile("" : "+v"(sel));
    const u32 dw = (u32)(lane & 31) * 4u, src = (Lv >> 5) * 4u, bpos = Lv & 31u;
    u32 hlo = 0u, hhi = 0u;
    {   // rows 0..47: 24 registers in flight, one round trip
        u32 w[24];   // (a group of four is consumed under the very condition it is loaded under)
        build_regs_any(w);
        build128_issue<0, 0, 6, 24>(rsrc, Lv, sel, dw, s, w);
        build128_consume<0, 0, 6, 24>(w, src, bpos, s, hlo, hhi);
    }
    if (s > 48) {   // rows 48..63 (one pair in twenty on config 3): a second trip, the registers are free again
        u32 w[8];
        build_regs_any(w);
        build128_issue<6, 6, 8, 8>(rsrc, Lv, sel, dw, s, w);
        build128_consume<6, 6, 8, 8>(w, src, bpos, s, hlo, hhi);
    }
    const u64 h = (u64)hlo | ((u64)hhi << 32);
    return lane < s ? (h & (s >= 64 ? ~0ull : ((1ull << s) - 1ull))) : 0ull;  // rows >= s were read in whole groups of 8
}

// ---------------------------------------------------------------------------
// Counting the simplices through an edge u->v on the local set.
//
// A vertex w of K = N(u) cap N(v) can sit before u (class P: w->u, w->v),
// between (M: u->w, w->v) or after v (S: u->w, v->w); with reciprocal pairs it
// can have several classes.  A simplex through u->v is an ordered clique of K
// whose class sequence is P*M*S*.  Instead of carrying the phase through the
// walk, the phase order is folded into the graph ("node splitting"): one node
// per (vertex, class), an arc (w,c)->(x,c') iff w->x and c' >= c.  Ordered
// cliques of that graph are exactly the simplices wanted, so the walk is a
// plain clique enumeration: cand & Hp[x], popcount at the leaves.
//
// Node indices: a vertex's first class lives at the vertex's own local index;
// its further classes ("extras", a handful per evaluation) take the two
// indices of u and v (never nodes themselves) and then the free indices from
// s upwards.  If they do not fit in 64 the caller falls back to the wide path.
// ---------------------------------------------------------------------------
#define FCM_NEEDS_WIDE (-2)
#ifndef FCM_SEAT_BRANCHY
#define FCM_SEAT_BRANCHY 0   // 1: pick 32-bit halves by scalar branches (fewer VALU, more SALU and branches)
#endif
#ifndef FCM_EVAL_V2
#define FCM_EVAL_V2 1        // round 4: branch-free class select; nodes with a single child sit the walk out
#endif

struct Cls { u64 P, M, S; };

// classes relative to u->v (local indices iu, iv) in the graph whose out-masks are myH (one per lane).
// With the in-masks build_local makes, pass (index of v, index of u) for an edge u->v of G.
// ballot of bit i (wave-uniform) of a 64-bit per-lane mask: one v_and on the half that holds it and one v_cmp
__device__ __forceinline__ u64 ballot_bit(u64 h, int i)
{
#if FCM_SEAT_BRANCHY
    const u32 m = 1u << (i & 31);
    return i < 32 ? ballot(((u32)h & m) != 0u) : ballot(((u32)(h >> 32) & m) != 0u);
#else
    return ballot(((h >> i) & 1ull) != 0ull);   // (a 64-bit shift rather than a scalar branch on the half: see seat_bit)
#endif
}
// per-lane bool from a wave-uniform mask: the SGPR pair is used as the select mask directly, no VALU
__device__ __forceinline__ bool lane_in(u64 mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

__device__ __forceinline__ Cls classify(u64 myH, int iu, int iv)
{
    const u64 outU = rdlane64(myH, iu), outV = rdlane64(myH, iv);  // v_readlane with an SGPR lane index: no LDS round trip
    const u64 inU = ballot_bit(myH, iu), inV = ballot_bit(myH, iv);
    const u64 nbm = ~((1ull << iu) | (1ull << iv));
    Cls c;
    c.P = inU & inV & nbm;    // w->u, w->v : before u
    c.M = outU & inV & nbm;   // u->w, w->v : between
    c.S = outU & outV & nbm;  // u->w, v->w : after v
    return c;
}
__device__ __forceinline__ bool extras_fit(const Cls &c, int s)
{
    const int ne = __popcll(c.P & c.M) + __popcll(c.S & (c.P | c.M));
    return ne <= 2 + (WAVE - s);
}

// Plain clique walk.  A node has T vertices chosen; `cand` = its children
// (non-empty).  delta[t] accumulates sign * (#cliques with t vertices).
template <int T, int MAXT, bool DETECT, class ACC = int>
__device__ __forceinline__ void visit(u64 cand, const u64 *Hp, int tmax, int sign, ACC (&delta)[MAXT + 1], u32 &overflow)
{
    if constexpr (T < MAXT) {
        if (T + 1 <= tmax) {
            const int pc = __popcll(cand);
            delta[T + 1] += sign * pc;
            // (a lone child has no children inside `cand` -- a row never holds its own node -- so a node with one child is a
            //  leaf: its lane sits the loop out.  Lockstep: the wave runs max-over-lanes trips, and most lanes have one child)
            if (DETECT || (T + 2 <= tmax && (!FCM_EVAL_V2 || pc > 1))) {
                u64 c = cand;
                while (c) {
                    const int x = __ffsll((long long)c) - 1;
                    c &= c - 1;
                    const u64 nc = cand & Hp[x];
                    if (nc) visit<T + 1, MAXT, DETECT, ACC>(nc, Hp, tmax, sign, delta, overflow);
                }
            }
        } else if (DETECT) {
            overflow = 1u;  // simplices deeper than the tracked dimensions exist
        }
    } else if (DETECT) {
        overflow = 1u;
    }
}

// 32-bit guard.  The walk keeps its counts in 32-bit integers per lane and sums them over the wave in 32 bits.  A
// split graph with tp arcs (<= 256, <= 4 per lane) and at most m <= 62 children per node holds at most tp * m^(t-2)
// ordered cliques of t nodes.  Up to 5 tracked levels that is below 6.7e7: nothing can wrap.  With 6 levels a lane
// stays below 4 * 62^4 = 5.9e7, but the wave's sum can pass 2^31 on a dense reciprocal local set (13 parts of 4
// vertices, all pairs reciprocal: 5e9 ordered 6-cliques through one edge): those kernels (MAXT == 6) check the lanes'
// magnitudes before summing (fcm_lane_guard: all below 2^24, so the sum is below 2^30).  Deeper kernels (MAXT >= 7,
// the generic ones) can wrap inside a lane: they bound every walk by tp * m^(t-2), m = the wave's largest child count
// (fcm_count_guard).  Either way a status bit is raised and the run fails loudly at the next read-out instead of
// returning wrapped counts.  FcmGuard::limit = 2^31 - 1 (FcmStepParams::guard_limit; a test hook lowers it).
struct FcmGuard { u64 limit; u32 tripped; };
// Per-lane accumulators of the walk: 32 bits in the kernels with up to 6 tracked levels (guards above), 64 bits in the generic
// deep ones (MAXT >= 7) -- since round 4: with 32-bit counts their bound tripped on every evaluation of a dense graph of a
// hundred vertices and each was then redone by the wave-uniform wide evaluator, 37 ms per proposal (tools/cliff_case.py,
// DESIGN.md 7).  With 64-bit counts the bound only guards 2^62.
template <int MAXT> using fcm_acc_t = std::conditional_t<(MAXT >= 7), long long, int>;
__device__ __forceinline__ int wave_sum_acc(int v) { return wave_sum_i32(v); }
__device__ __forceinline__ long long wave_sum_acc(long long v) { return wave_sum_i64(v); }
__device__ __forceinline__ void fcm_count_guard(int nch, int tp, int tmax, FcmGuard *g)
{
    if (!g) return;
    int m = nch;   // wave maximum of the child counts
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x111, 0xf, 0xf, false));
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x112, 0xf, 0xf, false));
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x114, 0xf, 0xf, false));
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x118, 0xf, 0xf, false));
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x142, 0xa, 0xf, false));
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x143, 0xc, 0xf, false));
    const u64 mm = (u64)(u32)__builtin_amdgcn_readlane(m, 63);
    const u64 lim = g->limit;
    u64 b = (u64)(u32)tp;
    // b <- b * mm per further level, stopped before the product can pass 64 bits (bits(b) + bits(mm) <= 63 keeps it below 2^63)
    for (int t = 3; t <= tmax && b <= lim; ++t) {
        if ((64 - __clzll((long long)b)) + (64 - __clzll((long long)(mm | 1ull))) > 63) { b = ~0ull; break; }
        b *= mm;
    }
    if (b > lim) g->tripped = 1u;
}

template <int MAXT>
__device__ __forceinline__ void fcm_lane_guard(const fcm_acc_t<MAXT> (&delta)[MAXT + 1], FcmGuard &g)
{
    if constexpr (MAXT == 6) {
        int m = 0;
#pragma unroll
        for (int t = 3; t <= MAXT; ++t) m = max(m, max(delta[t], -delta[t]));
        const u64 lim = g.limit < (1ull << 30) ? g.limit : (1ull << 30);
        if (ballot((u64)(u32)m * 64ull > lim)) g.tripped = 1u;
    }
}

// Levels 1 and 2 of an evaluation are wave-uniform numbers -- the nodes and the arcs of the split graph -- and are kept
// as scalars; only the deeper levels (t >= 3) are counted per lane and summed over the wave by the caller.
struct EvScal { int d1, d2; };

// The walk over the split graph in Hp (lane x = node x, `row` = its children, as two 32-bit halves).  One lane per
// node would leave the lanes in lockstep through nested child loops whose trip counts are the
// maxima over the wave: a handful of children per node, most lanes idle.  Instead the arcs
// (node, child) -- about as many as there are lanes -- are listed in LDS and dealt out one per
// lane, so that the first loop level disappears and only the short deeper loops remain per lane.
// The list sits behind Hp (FCM_PAIR_CAP 16-bit entries).
#define FCM_PAIR_CAP 256
template <int MAXT>
__device__ __forceinline__ void walk_nodes(u32 rlo, u32 rhi, const u64 *Hp, int tmax, int sign, int lane, fcm_acc_t<MAXT> (&delta)[MAXT + 1], EvScal &es,
                                           u64 *sacc = nullptr, u64 *stt = nullptr, FcmGuard *guard = nullptr)
{
    u32 dummy = 0;
    if (tmax < 2) return;
    const int nch = __popc(rlo) + __popc(rhi);
    if constexpr (MAXT < 3) {
        es.d2 += sign * wave_sum_i32(nch);
        return;
    } else {
        const int incl = wave_scan_i32(nch);
        const int tp = __builtin_amdgcn_readlane(incl, 63);
        if constexpr (MAXT >= 7) {   // bound first: an evaluation that passes it changes nothing (the caller counts it on the wide path)
            if (tmax >= 3 && tp != 0) fcm_count_guard(nch, tp, tmax, guard);
            if (guard && guard->tripped) return;
        }
        es.d2 += sign * tp;
        if (tmax < 3 || tp == 0) return;
        if (tp <= FCM_PAIR_CAP) {
            // scatter the arcs: 32-bit halves (one v_ffbl per arc instead of a 64-bit find-first-set)
            unsigned short *list = (unsigned short *)(Hp + WAVE) + (incl - nch);
            for (u32 c = rlo; c; c &= c - 1u) *list++ = (unsigned short)((u32)lane | ((u32)(__ffs((int)c) - 1) << 8));
            for (u32 c = rhi; c; c &= c - 1u) *list++ = (unsigned short)((u32)lane | ((u32)(__ffs((int)c) + 31) << 8));
            wave_sync();
            FCM_STAMP_PTR(4);                                      // (flips-only diagnostic) arc scan + scatter
            const unsigned short *rd = (const unsigned short *)(Hp + WAVE);
            for (int base = 0; base < tp; base += WAVE) {
                const int pi = base + lane;
                u64 nc = 0ull;
                if (pi < tp) {
                    const u32 e = rd[pi];
                    nc = Hp[e & 0xFFu] & Hp[e >> 8];
                }
                if (nc) visit<2, MAXT, false, fcm_acc_t<MAXT>>(nc, Hp, tmax, sign, delta, dummy);
            }
            FCM_STAMP_PTR(5);                                      // (flips-only diagnostic) arcs and deeper levels
        } else {
            // (more arcs than the list holds: one lane per node, children in a loop)
            const u64 row = (u64)rlo | ((u64)rhi << 32);
            for (u64 c = row; c; c &= c - 1) {
                const u64 nc = row & Hp[__ffsll((long long)c) - 1];
                if (nc) visit<2, MAXT, false, fcm_acc_t<MAXT>>(nc, Hp, tmax, sign, delta, dummy);
            }
        }
    }
}

// row bit `pos` := mask bit `orig` (both wave-uniform).  On the 64-bit values, five VALU and nothing else: picking the
// half words by scalar branches takes two VALU but a dozen SALU and four branches, and a scalar instruction costs this
// kernel twice what a vector one does (tools: MW_PROBE 7/8).
__device__ __forceinline__ void seat_bit(u32 blo, u32 bhi, u32 &rlo, u32 &rhi, int orig, int pos)
{
#if FCM_SEAT_BRANCHY
    u32 t;
    if (orig < 32) { t = __builtin_amdgcn_ubfe(blo, (u32)orig, 1u); asm volatile("" : "+v"(t)); }
    else { t = __builtin_amdgcn_ubfe(bhi, (u32)(orig - 32), 1u); asm volatile("" : "+v"(t)); }
    if (pos < 32) { rlo |= t << pos; asm volatile("" : "+v"(rlo)); }
    else { rhi |= t << (pos - 32); asm volatile("" : "+v"(rhi)); }
#else
    const u64 b = (u64)blo | ((u64)bhi << 32);
    const u64 t = ((b >> orig) & 1ull) << pos;
    rlo |= (u32)t; rhi |= (u32)(t >> 32);
#endif
}

// E(G, u->v): builds the split graph for classes `c` into Hp and counts.
// myH holds the raw local adjacency; local indices k, k+1 are the edge's
// endpoints.  Requires extras_fit(c, k+2).  Levels 1 and 2 go to `es`, deeper ones to `delta` (per lane).
template <int MAXT>
__device__ __forceinline__ void eval_nodes(u64 myH, u64 *Hp, const Cls &c, int k, int tmax, int sign, int lane,
                                           fcm_acc_t<MAXT> (&delta)[MAXT + 1], EvScal &es, u64 *sacc = nullptr, u64 *stt = nullptr, FcmGuard *guard = nullptr)
{
    const u64 uv = 3ull << k;
    const u64 prim1 = c.M & ~c.P, prim2 = c.S & ~(c.P | c.M);
    const u64 xm = c.P & c.M, xs = c.S & (c.P | c.M);  // vertices that also need an M node / an S node
    u64 N1 = prim1, N2 = prim2;
    // raw out-mask of the vertex this lane's node stands for, as two halves
    u32 blo = (u32)myH & ~(u32)uv, bhi = (u32)(myH >> 32) & ~(u32)(uv >> 32);
    const u64 gprim = c.P | prim1 | prim2;
    u32 rlo, rhi;
    // pass 1: seat the extra nodes (wave-uniform loops, a handful of trips): the seat's lane takes the vertex's mask
    // (extra r is node k + r: the two lanes of u and v first, then the free lanes from s on.  The vertex's mask goes to the
    //  seat's lane by v_readlane / v_writelane -- two vector instructions per half instead of a move and a select: the
    //  kernel's time goes with its vector instructions first, scalar ones are the cheaper currency)
    int r = 0;
    for (u64 m = xm; m; m &= m - 1, ++r) {
        const int orig = __ffsll((long long)m) - 1, pos = k + r;
        const u64 ho = rdlane64(myH, orig) & ~uv;
        blo = wrlane((u32)ho, pos, blo); bhi = wrlane((u32)(ho >> 32), pos, bhi);
        N1 |= 1ull << pos;
    }
    for (u64 m = xs; m; m &= m - 1, ++r) {
        const int orig = __ffsll((long long)m) - 1, pos = k + r;
        const u64 ho = rdlane64(myH, orig) & ~uv;
        blo = wrlane((u32)ho, pos, blo); bhi = wrlane((u32)(ho >> 32), pos, bhi);
        N2 |= 1ull << pos;
    }
    // pass 2: a child vertex shows up at its own index and at each of its extras (bit `orig` of the mask goes to bit `pos`;
    // both wave-uniform, so the half words involved are picked by scalar branches)
    rlo = blo & (u32)gprim; rhi = bhi & (u32)(gprim >> 32);
    r = 0;
    for (u64 m = xm; m; m &= m - 1, ++r) seat_bit(blo, bhi, rlo, rhi, __ffsll((long long)m) - 1, k + r);
    for (u64 m = xs; m; m &= m - 1, ++r) seat_bit(blo, bhi, rlo, rhi, __ffsll((long long)m) - 1, k + r);
    // children must not come earlier in the P*M*S* order: a P node may have any node as a child (G0), an M node the M and S
    // nodes (G1), an S node S nodes (G2), a lane that is no node nothing.  G0 > G1 > G2, so allowed = G2 | N1 & [P or M
    // node] | P & [P node], cut to the node lanes -- as lane masks turned into all-ones words and plain vector ANDs / ORs (no
    // branches: hipcc makes nested selects on SGPR-pair conditions into exec-mask regions, 25 scalar instructions and six branches)
    const u64 G1 = N1 | N2, G0 = c.P | G1;
#if FCM_EVAL_V2
    {
        const u32 selP = lane_in(c.P) ? 0xFFFFFFFFu : 0u, selPM = lane_in(c.P | N1) ? 0xFFFFFFFFu : 0u, selN = lane_in(G0) ? 0xFFFFFFFFu : 0u;
        u32 alo = selPM & (u32)N1, ahi = selPM & (u32)(N1 >> 32);
        alo |= selP & (u32)c.P; ahi |= selP & (u32)(c.P >> 32);
        alo |= (u32)N2; ahi |= (u32)(N2 >> 32);
        rlo &= alo & selN; rhi &= ahi & selN;
    }
#else
    {
        const u64 G2 = N2;
        const bool inP = lane_in(c.P), in1 = lane_in(N1), in2 = lane_in(N2);
        rlo &= inP ? (u32)G0 : (in1 ? (u32)G1 : (in2 ? (u32)G2 : 0u));
        rhi &= inP ? (u32)(G0 >> 32) : (in1 ? (u32)(G1 >> 32) : (in2 ? (u32)(G2 >> 32) : 0u));
    }
#endif
    Hp[lane] = (u64)rlo | ((u64)rhi << 32);
    wave_sync();
    FCM_STAMP_PTR(3);                                                  // (flips-only diagnostic) classes, seating, split rows
    if constexpr (MAXT >= 7) {
        walk_nodes<MAXT>(rlo, rhi, Hp, tmax, sign, lane, delta, es, sacc, stt, guard);
        if (tmax >= 1 && !(guard && guard->tripped)) es.d1 += sign * __popcll(G0);
    } else {
        if (tmax >= 1) es.d1 += sign * __popcll(G0);
        walk_nodes<MAXT>(rlo, rhi, Hp, tmax, sign, lane, delta, es, sacc, stt, guard);
    }
    wave_sync();
}

// ---------------------------------------------------------------------------
// A flip as ONE evaluation (round 4).  Reversing u->v subtracts the simplices through u->v and adds those through v->u.
// Around the pair the two share their P class (w->u, w->v) and their S class (u->w, v->w); only the middle differs: MA =
// {u->w, w->v} before, MB = {v->w, w->u} after.  A clique with no middle node is counted by both evaluations and cancels.
// So ONE split graph carries both middles -- no arc between an MA and an MB node -- and every clique is weighted by what it
// holds: -1 with an MA node, +1 with an MB node, 0 with neither.  A prefix of P nodes only is "undetermined" (sg = 0) until
// a middle node joins it; S children of an undetermined prefix lead nowhere (no middle can follow an S node) and are neither
// listed as arcs nor walked.  One seating, one scan, one arc list -- 70 arcs on the headline graph where the two evaluations
// list 118 -- and the P/S-only cliques, most of the count, are never enumerated.  Weights at a level: children in MB count
// +1, in MA -1, in S the prefix's sign (0 while undetermined), in P 0: three masked popcounts.
// cls[x]: the sign a node brings (-1 MA, +1 MB, 0 P; S nodes hold 2 and are never asked).
// ---------------------------------------------------------------------------
#define FCM_MERGED_CAP 224   // arcs the list takes here: the last 64 bytes of its region hold cls[]
template <int T, int MAXT>
__device__ __forceinline__ void visit_sg(u64 cand, int sg, const u64 *Hp, const signed char *cls, u64 NA, u64 NB, u64 NS, int (&delta)[MAXT + 1])
{
    if constexpr (T < MAXT) {
        const u32 clo = (u32)cand, chi = (u32)(cand >> 32);
        const int b = __popc(clo & (u32)NB) + __popc(chi & (u32)(NB >> 32));
        const int a = __popc(clo & (u32)NA) + __popc(chi & (u32)(NA >> 32));
        const int s = __popc(clo & (u32)NS) + __popc(chi & (u32)(NS >> 32));
        delta[T + 1] += b - a + sg * s;
        if constexpr (T + 2 <= MAXT) {
            if (__popcll(cand) > 1) {                      // (a lone child has no children inside `cand`)
                u64 c = sg == 0 ? (cand & ~NS) : cand;
                while (c) {
                    const int x = __ffsll((long long)c) - 1;
                    c &= c - 1;
                    const u64 nc = cand & Hp[x];
                    if (nc) visit_sg<T + 1, MAXT>(nc, sg != 0 ? sg : (int)cls[x], Hp, cls, NA, NB, NS, delta);
                }
            }
        }
    }
}
__device__ __forceinline__ bool flip_merged_fits(u64 P, u64 MA, u64 MB, u64 S, int k)
{
    return __popcll(MA & P) + __popcll(MB & (P | MA)) + __popcll(S & (P | MA | MB)) <= WAVE - k;
}
// myH: the raw local masks (in-masks of the build: the evaluator's transposed graph), local indices k, k + 1 = the pair;
// P, MA, MB, S as classify() gives them for the two directions.  Requires flip_merged_fits().  Returns false -- nothing
// counted -- if the arc list does not hold the arcs (the caller then runs the two evaluations).  MAXT <= 6, exact depth.
template <int MAXT>
__device__ __forceinline__ bool eval_flip_merged(u64 myH, u64 *Hp, u64 P, u64 MA, u64 MB, u64 S, int k, int lane, int (&delta)[MAXT + 1], EvScal &es)
{
    const u64 uv = 3ull << k;
    const u64 xa = MA & P, xb = MB & (P | MA), xs = S & (P | MA | MB);     // vertices that need a further node in MA / MB / S
    u64 NA = MA & ~P, NB = MB & ~(P | MA), NS = S & ~(P | MA | MB);
    u32 blo = (u32)myH & ~(u32)uv, bhi = (u32)(myH >> 32) & ~(u32)(uv >> 32);
    int r = 0;
    for (u64 m = xa; m; m &= m - 1, ++r) {
        const u64 ho = rdlane64(myH, __ffsll((long long)m) - 1) & ~uv;
        blo = wrlane((u32)ho, k + r, blo); bhi = wrlane((u32)(ho >> 32), k + r, bhi);
        NA |= 1ull << (k + r);
    }
    for (u64 m = xb; m; m &= m - 1, ++r) {
        const u64 ho = rdlane64(myH, __ffsll((long long)m) - 1) & ~uv;
        blo = wrlane((u32)ho, k + r, blo); bhi = wrlane((u32)(ho >> 32), k + r, bhi);
        NB |= 1ull << (k + r);
    }
    for (u64 m = xs; m; m &= m - 1, ++r) {
        const u64 ho = rdlane64(myH, __ffsll((long long)m) - 1) & ~uv;
        blo = wrlane((u32)ho, k + r, blo); bhi = wrlane((u32)(ho >> 32), k + r, bhi);
        NS |= 1ull << (k + r);
    }
    const u64 gprim = P | MA | MB | S;
    u32 rlo = blo & (u32)gprim, rhi = bhi & (u32)(gprim >> 32);
    r = 0;
    for (u64 m = xa; m; m &= m - 1, ++r) seat_bit(blo, bhi, rlo, rhi, __ffsll((long long)m) - 1, k + r);
    for (u64 m = xb; m; m &= m - 1, ++r) seat_bit(blo, bhi, rlo, rhi, __ffsll((long long)m) - 1, k + r);
    for (u64 m = xs; m; m &= m - 1, ++r) seat_bit(blo, bhi, rlo, rhi, __ffsll((long long)m) - 1, k + r);
    // children by the parent's class: P -> any node, MA -> MA and S, MB -> MB and S, S -> S; a lane that is no node: none.
    // The listed arcs (elo, ehi): not P -> S (nothing can follow but S nodes: no middle), none from an S node.  The sign a
    // node brings (cv) and the tag its list entries carry (bits 6, 7) are set in the same pass.  One exec mask per class, two
    // or three vector instructions under each (all 64 lanes are active here: the callers' branches are wave-uniform).
    const u64 GA = NA | NS, GB = NB | NS, G0 = P | GA | NB, GN = ~G0;
    u32 elo, ehi, cv, base;
    asm volatile("s_mov_b64 exec, %[mP]\n\t"
                 "v_and_b32 %[rlo], %[g0l], %[rlo]\n\tv_and_b32 %[rhi], %[g0h], %[rhi]\n\t"
                 "v_bitop3_b32 %[elo], %[rlo], %[nsl], %[rlo] bitop3:0x30\n\tv_bitop3_b32 %[ehi], %[rhi], %[nsh], %[rhi] bitop3:0x30\n\t"
                 "v_mov_b32 %[cv], 0\n\tv_mov_b32 %[base], %[lane]\n\t"
                 "s_mov_b64 exec, %[mA]\n\t"
                 "v_and_b32 %[rlo], %[gal], %[rlo]\n\tv_and_b32 %[rhi], %[gah], %[rhi]\n\t"
                 "v_mov_b32 %[elo], %[rlo]\n\tv_mov_b32 %[ehi], %[rhi]\n\tv_mov_b32 %[cv], -1\n\tv_or_b32 %[base], 64, %[lane]\n\t"
                 "s_mov_b64 exec, %[mB]\n\t"
                 "v_and_b32 %[rlo], %[gbl], %[rlo]\n\tv_and_b32 %[rhi], %[gbh], %[rhi]\n\t"
                 "v_mov_b32 %[elo], %[rlo]\n\tv_mov_b32 %[ehi], %[rhi]\n\tv_mov_b32 %[cv], 1\n\tv_or_b32 %[base], 0x80, %[lane]\n\t"
                 "s_mov_b64 exec, %[mS]\n\t"
                 "v_and_b32 %[rlo], %[nsl], %[rlo]\n\tv_and_b32 %[rhi], %[nsh], %[rhi]\n\t"
                 "v_mov_b32 %[elo], 0\n\tv_mov_b32 %[ehi], 0\n\tv_mov_b32 %[cv], 2\n\tv_mov_b32 %[base], %[lane]\n\t"
                 "s_mov_b64 exec, %[mN]\n\t"
                 "v_mov_b32 %[rlo], 0\n\tv_mov_b32 %[rhi], 0\n\tv_mov_b32 %[elo], 0\n\tv_mov_b32 %[ehi], 0\n\tv_mov_b32 %[cv], 0\n\tv_mov_b32 %[base], %[lane]\n\t"
                 "s_mov_b64 exec, -1"
                 : [rlo] "+v"(rlo), [rhi] "+v"(rhi), [elo] "=&v"(elo), [ehi] "=&v"(ehi), [cv] "=&v"(cv), [base] "=&v"(base)
                 : [mP] "s"(P), [mA] "s"(NA), [mB] "s"(NB), [mS] "s"(NS), [mN] "s"(GN), [g0l] "s"((u32)G0), [g0h] "s"((u32)(G0 >> 32)),
                   [gal] "s"((u32)GA), [gah] "s"((u32)(GA >> 32)), [gbl] "s"((u32)GB), [gbh] "s"((u32)(GB >> 32)), [nsl] "s"((u32)NS), [nsh] "s"((u32)(NS >> 32)),
                   [lane] "v"((u32)lane)
                 : "memory");
    signed char *cls = (signed char *)(Hp + WAVE) + 2 * FCM_MERGED_CAP;
    wave_sync();
    Hp[lane] = (u64)rlo | ((u64)rhi << 32);
    cls[lane] = (signed char)cv;
    const int nch = __popc(elo) + __popc(ehi);
    const int incl = wave_scan_i32(nch);
    const int tp = __builtin_amdgcn_readlane(incl, 63);
    if (tp > FCM_MERGED_CAP) { wave_sync(); return false; }
    es.d1 += __popcll(NB) - __popcll(NA);
    if (MAXT < 2 || tp == 0) { wave_sync(); return true; }
    {
        unsigned short *list = (unsigned short *)(Hp + WAVE) + (incl - nch);
        for (u32 c = elo; c; c &= c - 1u) *list++ = (unsigned short)(base | ((u32)(__ffs((int)c) - 1) << 8));
        for (u32 c = ehi; c; c &= c - 1u) *list++ = (unsigned short)(base | ((u32)(__ffs((int)c) + 31) << 8));
    }
    wave_sync();
    const unsigned short *rd = (const unsigned short *)(Hp + WAVE);
    int d2 = 0;
    for (int b0 = 0; b0 < tp; b0 += WAVE) {
        const int pi = b0 + lane;
        if (pi < tp) {
            const u32 e = rd[pi];
            const u32 y = e >> 8;
            const int sx = (e & 0x40u) ? -1 : ((e & 0x80u) ? 1 : 0);
            const int sg = sx != 0 ? sx : (int)cls[y];                        // (a P parent lists no S child: cls[y] is -1, 0 or +1 here)
            d2 += sg;
            if constexpr (MAXT >= 3) {
                const u64 nc = Hp[e & 0x3Fu] & Hp[y];
                if (nc) visit_sg<2, MAXT>(nc, sg, Hp, cls, NA, NB, NS, delta);
            }
        }
    }
    es.d2 += wave_sum_i32(d2);
    wave_sync();
    return true;
}

// ---- the three evaluations a simple move is made of (fast path) -------------
// Lv = the local vertex list (K, then big, small: lane k = big, lane k+1 = small),
// loaded by the caller.  The raw masks stay in registers; Hp = split
// graph (64 u64 in LDS).  Each returns FCM_NEEDS_WIDE when the extras do
// not fit; the caller then zeroes delta and redoes the proposal on the wide path.

// single_edge_flip on undirected edge (big,small): returns 0 if the pair is
// reciprocal (empty transition), 1 if big->small was flipped, 2 if small->big,
// -1 if the bitmap disagrees with the static table.
template <int MAXT>
__device__ __forceinline__ int flip_eval(const rsrc_t rows, u32 stride32, u32 Lv, int k,
                                         u64 *Hp, int lane, int tmax, fcm_acc_t<MAXT> (&delta)[MAXT + 1], EvScal &es, FcmGuard *guard = nullptr)
{
    const int s = k + 2;
    u64 myH = build_local(rows, stride32, Lv, s, lane);
    const u64 hk = rdlane64(myH, k), hk1 = rdlane64(myH, k + 1);
    const u32 ab = (u32)((hk1 >> k) & 1ull), ba = (u32)((hk >> (k + 1)) & 1ull);  // big->small, small->big
    if (ab == ba) return ab ? 0 : -1;
    const int iu = ab ? k : k + 1, iv = ab ? k + 1 : k;  // u->v present
    Cls c = classify(myH, iv, iu);
    // after the flip P and S are the same sets, M becomes {v->w, w->u}
    Cls c2;
    c2.P = c.P; c2.S = c.S;
    c2.M = (ab ? hk : hk1) & ballot((myH >> iv) & 1ull) & ~(3ull << k);
    if (!extras_fit(c, s) || !extras_fit(c2, s)) return FCM_NEEDS_WIDE;
    eval_nodes<MAXT>(myH, Hp, c, k, tmax, -1, lane, delta, es, nullptr, nullptr, guard);
    eval_nodes<MAXT>(myH, Hp, c2, k, tmax, +1, lane, delta, es, nullptr, nullptr, guard);
    return ab ? 1 : 2;
}

// double_edge_move step 1: subtract the simplices through one direction of the
// reciprocal pair (big,small).  coin=1 removes big->small.  Returns 1, or 0 if
// the pair is not reciprocal in the bitmap.
template <int MAXT>
__device__ __forceinline__ int del_eval(const rsrc_t rows, u32 stride32, u32 Lv, int k,
                                        u32 coin, u64 *Hp, int lane, int tmax, fcm_acc_t<MAXT> (&delta)[MAXT + 1], EvScal &es, FcmGuard *guard = nullptr)
{
    const int s = k + 2;
    const u64 myH = build_local(rows, stride32, Lv, s, lane);
    const u32 ab = (u32)((rdlane64(myH, k + 1) >> k) & 1ull), ba = (u32)((rdlane64(myH, k) >> (k + 1)) & 1ull);
    const int iu = coin ? k : k + 1, iv = coin ? k + 1 : k;
    const Cls c = classify(myH, iv, iu);
    if (!extras_fit(c, s)) return FCM_NEEDS_WIDE;
    eval_nodes<MAXT>(myH, Hp, c, k, tmax, -1, lane, delta, es, nullptr, nullptr, guard);
    return (ab & ba) ? 1 : 0;
}

// double_edge_move step 2: on the graph without dfrom->dto, add the reverse of
// the single edge of (big,small) and add the simplices through it.  fwd=1
// means big->small is the existing direction.  `myH` = the lane's raw in-mask of
// the local set Lv (k vertices of K, then big, small), built by the caller.
template <int MAXT>
__device__ __forceinline__ int add_eval_built(u64 myH, u32 Lv, int k, u32 fwd, u32 dfrom, u32 dto, u64 *Hp, int lane,
                                              int tmax, fcm_acc_t<MAXT> (&delta)[MAXT + 1], EvScal &es, FcmGuard *guard = nullptr)
{
    const int s = k + 2;
    const bool act = lane < s;
    const u64 mf = ballot(act && Lv == dfrom), mt = ballot(act && Lv == dto);
    if (mf && mt) {  // the pending removal, if both its endpoints are local
        const int fi = __ffsll((long long)mf) - 1, ti = __ffsll((long long)mt) - 1;
        if (lane == ti) myH &= ~(1ull << fi);
    }
    const int ia = fwd ? k : k + 1, ib = fwd ? k + 1 : k;  // a->b exists, add b->a
    if (lane == ia) myH |= (1ull << ib);
    const Cls c = classify(myH, ia, ib);
    if (!extras_fit(c, s)) return FCM_NEEDS_WIDE;
    eval_nodes<MAXT>(myH, Hp, c, k, tmax, +1, lane, delta, es, nullptr, nullptr, guard);
    return 1;
}

// A kernel argument re-read where it is used: a scalar load from the kernarg segment, no vector instruction.  The step kernels
// have 80 SGPRs at 8 waves per SIMD; what is kept in SGPRs across their loops is spilled to VGPR lanes with the s_load tuple it
// came in, and every use brings the whole tuple back by v_readlane (the list pointer in the multi-wave kernel: eight lanes per list
// load, 12 vector instructions per proposal; round 4, tools/knob_sq.sh).  The rest of the chain's context through a block of its
// own was measured and dropped (profiles/r04_kctx_block_dropped.diff): what it saves in v_readlane it spends in scalar moves.
template <int OFF>
__device__ __forceinline__ u64 fcm_karg64()
{
    u64 v;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"((u64)(size_t)__builtin_amdgcn_kernarg_segment_ptr()), "n"(OFF) : "memory");
    return v;
}
#ifndef FCM_NB_KARG
#define FCM_NB_KARG 1   // the one-wave kernel's clique moves take the list pointer that way too (default mix at 4096 chains + 2 %; its simple
#endif                  // moves and the cooperative kernel: no difference, left as they were)

// local vertex list of an adjacent pair: K, then big, small
__device__ __forceinline__ u32 load_list(const u32 *nb, u32 off, int k, u32 big, u32 small, int lane)
{
#if defined(MW_ABL) && (MW_ABL & 32)   // (ablation probe: no list loads -- made-up vertices)
    return lane < k ? (((u32)lane + 1u) * 40503u + big * 7u + off) % 997u : (lane == k ? big : small);
#endif
    return lane < k ? nb[off + lane] : (lane == k ? big : small);
}

// ===========================================================================
// Wide evaluator: local sets of 65..256 vertices, NW = ceil(s/64) mask words.
// Everything is wave-uniform: all lanes run the same DFS on the same values
// (masks live in LDS, the current frame in registers), so there is no per-lane
// stack and the register cost is a few dozen VGPRs whatever NW is.
// ===========================================================================
#define FCM_WIDE_LEVELS 16
struct Wide {
    u64 *H;        // [s][NW] out-masks of the local vertices
    u64 *cls;      // [3][4]  P, M, S
    u64 *stk;      // [FCM_WIDE_LEVELS][2][4]  saved (cand, rem) frames
    long long *cnt;  // [16] signed simplex counts by number of K-vertices
    int *stkph;    // [FCM_WIDE_LEVELS][2]     saved (ph, ph2)
    u32 *L;        // [64*NW] local vertex ids
    int NW;
};
// u64 words of dynamic LDS a workgroup needs for local sets of up to 64*NW vertices
__host__ __device__ constexpr inline unsigned fcm_lds_words(int NW)
{
    if (NW <= 1) return 3u * 64u;  // (64 spare) + Hp + the arc list of walk_nodes
    return 64u * NW * NW + 12u + FCM_WIDE_LEVELS * 8u + 16u + FCM_WIDE_LEVELS + 32u * NW;
}
__device__ __forceinline__ Wide wide_carve(u64 *smem, int NW)
{
    Wide W;
    W.NW = NW;
    W.H = smem;
    W.cls = W.H + 64 * NW * NW;
    W.stk = W.cls + 12;
    W.cnt = (long long *)(W.stk + FCM_WIDE_LEVELS * 8);
    W.stkph = (int *)(W.cnt + 16);
    W.L = (u32 *)(W.stkph + 2 * FCM_WIDE_LEVELS);
    return W;
}

__device__ __forceinline__ void wide_zero_counts(const Wide &W, int lane)
{
    if (lane < 16) W.cnt[lane] = 0;
    wave_sync();
}

// Induced out-adjacency into W.H; the vertex list is already in W.L[0..s).
__device__ __forceinline__ void wide_build(const Wide W, const u32 *rows, u32 stride32, int s, int lane)
{
    const int NW = W.NW;
    bool act[4];
    u32 woff[4], bit[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int j = g * 64 + lane;
        act[g] = g < NW && j < s;
        const u32 lv = act[g] ? W.L[j] : 0u;
        woff[g] = lv >> 5;
        bit[g] = lv & 31u;
    }
    for (int i0 = 0; i0 < s; i0 += 4) {
        u32 w[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 vi = W.L[min(i0 + q, s - 1)];
            const u32 *row = rows + (size_t)vi * stride32;
#pragma unroll
            for (int g = 0; g < 4; ++g) w[q][g] = act[g] ? row[woff[g]] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u64 m = ballot(act[g] && ((w[q][g] >> bit[g]) & 1u));
                if (g < NW && i0 + q < s) W.H[(i0 + q) * NW + g] = m;  // every lane writes the same value
            }
        }
    }
    wave_sync();
}

__device__ __forceinline__ bool wide_has(const Wide &W, int i, int j) { return (W.H[i * W.NW + (j >> 6)] >> (j & 63)) & 1ull; }
__device__ __forceinline__ void wide_set(const Wide &W, int i, int j, bool present)
{
    u64 &w = W.H[i * W.NW + (j >> 6)];
    const u64 b = 1ull << (j & 63);
    w = present ? (w | b) : (w & ~b);  // uniform read-modify-write, same value from every lane
}

// Uniform DFS over the classified local set (classes in W.cls).  Adds
// sign * (#simplices with t K-vertices) to W.cnt[t].
template <bool DETECT>
__device__ __forceinline__ void wide_dfs(const Wide W, int tmax, int sign, u32 *overflow)
{
    const int NW = W.NW;
    if (tmax < 1) return;
    u64 cand[4], rem[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        cand[g] = g < NW ? (W.cls[0 * 4 + g] | W.cls[1 * 4 + g] | W.cls[2 * 4 + g]) : 0ull;
        rem[g] = 0ull;
    }
    int ph = 0, ph2 = -1, level = 0;
    for (;;) {
        if ((rem[0] | rem[1] | rem[2] | rem[3]) == 0ull) {
            ph2 = max(ph2 + 1, ph);
            if (ph2 > 2) {  // node exhausted: back to the parent
                if (level == 0) break;
                --level;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    cand[g] = W.stk[(level * 2 + 0) * 4 + g];
                    rem[g] = W.stk[(level * 2 + 1) * 4 + g];
                }
                ph = W.stkph[2 * level];
                ph2 = W.stkph[2 * level + 1];
                continue;
            }
            int pc = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                rem[g] = g < NW ? (cand[g] & W.cls[ph2 * 4 + g]) : 0ull;
                pc += __popcll(rem[g]);
            }
            W.cnt[level + 1] += (long long)sign * pc;
            if (!(DETECT || level + 2 <= tmax)) {
#pragma unroll
                for (int g = 0; g < 4; ++g) rem[g] = 0ull;
            }
            continue;
        }
        // next child x of class ph2
        int x = 0;
        bool done = false;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bool take = !done && rem[g] != 0ull;
            x = take ? (g * 64 + __ffsll((long long)rem[g]) - 1) : x;
            rem[g] = take ? (rem[g] & (rem[g] - 1ull)) : rem[g];
            done = done || take;
        }
        u64 nc[4], any = 0ull;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u64 ge = 0ull;
            if (g < NW) {
                ge = W.cls[2 * 4 + g];
                if (ph2 <= 1) ge |= W.cls[1 * 4 + g];
                if (ph2 == 0) ge |= W.cls[0 * 4 + g];
                nc[g] = cand[g] & W.H[x * NW + g] & ge;
            } else {
                nc[g] = 0ull;
            }
            any |= nc[g];
        }
        if (any) {
            if (level + 2 <= tmax && level + 1 < FCM_WIDE_LEVELS) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    W.stk[(level * 2 + 0) * 4 + g] = cand[g];
                    W.stk[(level * 2 + 1) * 4 + g] = rem[g];
                    cand[g] = nc[g];
                    rem[g] = 0ull;
                }
                W.stkph[2 * level] = ph;
                W.stkph[2 * level + 1] = ph2;
                ++level;
                ph = ph2;
                ph2 = ph - 1;
            } else if (DETECT) {
                *overflow = 1u;  // simplices deeper than the tracked dimensions exist
            }
        }
    }
    wave_sync();
}

// classes of every local vertex relative to the edge iu->iv, into W.cls
__device__ __forceinline__ void wide_classify(const Wide W, int iu, int iv, int s, int lane)
{
    const int NW = W.NW;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g < NW) {
            const int j = g * 64 + lane;
            const bool a = j < s;
            const u64 inU = ballot(a && wide_has(W, a ? j : 0, iu)), inV = ballot(a && wide_has(W, a ? j : 0, iv));
            const u64 outU = W.H[iu * NW + g], outV = W.H[iv * NW + g];
            u64 nbm = ~0ull;
            if ((iu >> 6) == g) nbm &= ~(1ull << (iu & 63));
            if ((iv >> 6) == g) nbm &= ~(1ull << (iv & 63));
            W.cls[0 * 4 + g] = inU & inV & nbm;
            W.cls[1 * 4 + g] = outU & inV & nbm;
            W.cls[2 * 4 + g] = outU & outV & nbm;
        }
    }
    wave_sync();
}

__device__ __forceinline__ void wide_load_list(const Wide &W, const u32 *nb, u32 off, int k, u32 big, u32 small, int lane)
{
    for (int j = lane; j < k + 2; j += WAVE) W.L[j] = j < k ? nb[off + j] : (j == k ? big : small);
    wave_sync();
}

// The wide twins of flip_eval / del_eval / add_eval.  Results go to W.cnt.
__device__ __forceinline__ int wide_flip(const Wide W, const u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big,
                                      u32 small, int lane, int tmax)
{
    const int s = k + 2;
    wide_load_list(W, nb, off, k, big, small, lane);
    wide_build(W, rows, stride32, s, lane);
    const bool ab = wide_has(W, k, k + 1), ba = wide_has(W, k + 1, k);
    if (ab == ba) return ab ? 0 : -1;
    const int iu = ab ? k : k + 1, iv = ab ? k + 1 : k;
    wide_classify(W, iu, iv, s, lane);
    wide_dfs<false>(W, tmax, -1, nullptr);
    wide_set(W, iu, iv, false);
    wide_set(W, iv, iu, true);
    wave_sync();
    wide_classify(W, iv, iu, s, lane);
    wide_dfs<false>(W, tmax, +1, nullptr);
    return ab ? 1 : 2;
}
__device__ __forceinline__ bool wide_del(const Wide W, const u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big,
                                      u32 small, u32 coin, int lane, int tmax)
{
    const int s = k + 2;
    wide_load_list(W, nb, off, k, big, small, lane);
    wide_build(W, rows, stride32, s, lane);
    const bool ab = wide_has(W, k, k + 1), ba = wide_has(W, k + 1, k);
    const int iu = coin ? k : k + 1, iv = coin ? k + 1 : k;
    wide_classify(W, iu, iv, s, lane);
    wide_dfs<false>(W, tmax, -1, nullptr);
    return ab && ba;
}
__device__ __forceinline__ void wide_add(const Wide W, const u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big,
                                      u32 small, u32 fwd, u32 dfrom, u32 dto, int lane, int tmax)
{
    const int s = k + 2;
    wide_load_list(W, nb, off, k, big, small, lane);
    wide_build(W, rows, stride32, s, lane);
    int fi = -1, ti = -1;
    for (int g = 0; g < W.NW; ++g) {
        const int j = g * 64 + lane;
        const u32 lv = j < s ? W.L[j] : 0xFFFFFFFFu;
        const u64 mf = ballot(j < s && lv == dfrom), mt = ballot(j < s && lv == dto);
        if (mf) fi = g * 64 + __ffsll((long long)mf) - 1;
        if (mt) ti = g * 64 + __ffsll((long long)mt) - 1;
    }
    if (fi >= 0 && ti >= 0) wide_set(W, fi, ti, false);
    const int ia = fwd ? k : k + 1, ib = fwd ? k + 1 : k;
    wide_set(W, ib, ia, true);
    wave_sync();
    wide_classify(W, ib, ia, s, lane);
    wide_dfs<false>(W, tmax, +1, nullptr);
}

#include "fcm_xwide.hpp"
#include "fcm_clique.hpp"

// ===========================================================================
__device__ __forceinline__ u32 mw_lds_addr_of(const void *p) { return (u32)(size_t)(__attribute__((address_space(3))) const char *)p; }
// tallies of the one-wave kernel, u32 words in LDS: counters in words 0..9 (the first eight count events), then the OR
// of "count entry d was non-zero after a transition" masks and of the proposals' status words
enum { OT_ACCEPTED = 0, OT_EMPTY, OT_FLIP, OT_DMOVE, OT_CPERM, OT_CSWAP, OT_WIDE, OT_BIG, OT_SUMK, OT_CHANGES, OT_NZ, OT_STATUS,
       OT_PAIRS, OT_SHARED };   // clique moves: vertex pairs with a changed direction (one local build each); (pairs - 1) x clique size: rows of the
                                // clique's own vertices that the builds of one move read more than once
#define FCM_TALLY_LDS_WORDS 56u   // u64 words: tallies (8), then the chain's counts and bounds E[16] = {count, min, max}

// Step kernel
// ===========================================================================
// MINW = minimum waves per SIMD the register allocator must leave room for.
// CLIQUE = true adds the clique moves (fcm_clique.hpp); the simple-move kernel
// is compiled without them and keeps its register allocation.
template <int MAXT, int MINW, int CLIQUE, bool EXACT>   // CLIQUE: 0 simple moves only, 1 clique moves, 2 clique moves incl. local sets of 257..1024 vertices
__global__ __launch_bounds__(WAVE, MINW) void fcm_step_kernel(const FcmStepParams p)
{
    extern __shared__ u64 smem[];  // fcm_lds_words(p.maxnw) words
    u64 *Hp = smem + WAVE;  // fast path: the split graph (and the arc list behind it); the wide path reuses the region
    const int lane = threadIdx.x;
    const u32 chain = blockIdx.x;
    if (chain >= p.nchains) return;

    u32 *rows = p.rows + (size_t)chain * p.rows_per_chain;
    const rsrc_t rrows = make_rows_rsrc(rows, p.rows_per_chain * 4ull);
    u32 *dbl = p.dbl + (size_t)chain * p.dbl_stride;
    u64 *cnt_g = (u64 *)p.counts + (size_t)chain * FCM_DEV_MAX_COUNTS;
    u64 *st_g = (u64 *)p.stats + (size_t)chain * FCM_DEV_NSTATS;
    const u32 *nb = p.nb;

    const int NC = p.ncounts;
    const int tmax = EXACT ? MAXT : NC - 2;  // EXACT: the host picked the variant with MAXT == NC - 2
    const bool cl = lane < NC;
    // lane d holds count[d] and its bounds (zero-padded, src/util.rs:53-57)
    // (in LDS, behind the tallies -- E[lane] = {count, min, max} -- and read where a proposal is decided: six registers
    // fewer that live through the whole loop)
    u64 *ent = smem + fcm_lds_words(p.maxnw) + (CLIQUE != 0 ? fcm_clique_lds_words(p.chg_cap) : 0u) + 8u;
    {
        const u64 cnt = cl ? cnt_g[lane] : 0ull;
        const u64 bmin = cl ? p.bmin[lane] : 0ull;
        const u64 bmax = cl ? p.bmax[lane] : ~0ull;
        if (lane < 16) { ent[lane * 3 + 0] = cnt; ent[lane * 3 + 1] = bmin; ent[lane * 3 + 2] = bmax; }
    }

    u64 sampled = st_g[0];
    // The launch's tallies live in 16 words of LDS (OT_*: one masked ds_add and one ds_or per proposal), not in a dozen
    // 64-bit scalars carried -- spilled -- around the whole loop; they are added to the chain's stats row at the end.
    u32 *tly = (u32 *)(smem + fcm_lds_words(p.maxnw) + (CLIQUE != 0 ? fcm_clique_lds_words(p.chg_cap) : 0u));
    if (lane < 16) tly[lane] = 0u;
    wave_sync();
    u32 *slot_of = CLIQUE != 0 ? p.slot_of + (size_t)chain * p.U : nullptr;

    const u32 U = p.U, D = p.D;
    const u64 Mtot = (u64)U + D;
    const u32 k0 = (u32)p.seed, k1 = (u32)(p.seed >> 32);
    const u32 gchain = p.first_chain + chain;
    const u32 stride32 = p.stride32;
    const int maxnw = p.maxnw;
    u64 *xw_ws = p.xw_ws ? (u64 *)p.xw_ws + (size_t)chain * FCM_XW_WORDS : nullptr;

    // is the current state inside the bounds?  (decides whether an empty
    // transition is "accepted", src/lib.rs:186-187)
    bool in_bounds;
    {
        wave_sync();
        const u64 c0 = lane < 16 ? ent[lane * 3] : 0ull, mn = lane < 16 ? ent[lane * 3 + 1] : 0ull, mx = lane < 16 ? ent[lane * 3 + 2] : ~0ull;
        in_bounds = ballot(cl && (c0 < mn || c0 > mx)) == 0ull;
    }

    // (deep kernels count in 64 bits: their bound guards 2^62 unless the test hook FCM_TEST_GUARD_LIMIT has lowered it)
    FcmGuard guard = {(MAXT >= 7 && p.guard_limit == 0x7FFFFFFFull) ? (1ull << 62) : p.guard_limit, 0u};
    FCM_STAMP_DECL
    for (u64 done = 0; done < p.nprop; done += WAVE) {
        // ---- batch: lane s draws proposal `sampled + s` ------------------
        const u64 t = sampled + (u64)lane;
        u32 w[4];
        philox4x32_10((u32)t, (u32)(t >> 32), gchain, 0u, k0, k1, w);
        const int l_move = ((u64)w[0] < p.cum0) ? 0 : (((u64)w[0] < p.cum1) ? 1 : (((u64)w[0] < p.cum2) ? 2 : 3));
        const u32 l_coin = w[1];  // coin = bit 0; the clique moves use the whole word for the order pick
        const u64 x64 = (u64)w[2] | ((u64)w[3] << 32);
        const u64 l_idx = l_move >= 2 ? x64 : __umul64hi(x64, l_move == 0 ? Mtot : (u64)D);
        FcmEdgeEntry l_e = {0u, 0u, 0u, 0u};
        if (l_move == 0 && l_idx < U) l_e = p.etab[l_idx];
        // double-edge moves: the first two single-edge candidates (block sub=1)
        // are drawn here too, once per 64 proposals instead of once per move
        u64 l_c0 = 0ull, l_c1 = 0ull;
        if (l_move == 1) {
            u32 v[4];
            philox4x32_10((u32)t, (u32)(t >> 32), gchain, 1u, k0, k1, v);
            l_c0 = __umul64hi((u64)v[0] | ((u64)v[1] << 32), Mtot);
            l_c1 = __umul64hi((u64)v[2] | ((u64)v[3] << 32), Mtot);
            if (l_c0 < U) l_e = p.etab[l_c0];  // candidate 0 wins ~9 times in 10: have its table entry ready
        }
        // ... and the reciprocal pair its slot names *now*.  The slot list is mutable, so this is a
        // guess (right unless that very slot is rewritten within the next 64 proposals); the move
        // re-reads the slot in the same round trip as its vertex lists and falls back if it changed.
        u32 l_ed = 0u;
        FcmEdgeEntry l_de = {0u, 0u, 0u, 0u};
        if (l_move == 1 && D > 0) {
            l_ed = dbl[(u32)l_idx];
            l_de = p.etab[l_ed];
        }

        const int nbatch = (int)min((u64)WAVE, p.nprop - done);
        for (int sidx = 0; sidx < nbatch; ++sidx) {
            const int move = (int)rdlane((u32)l_move, sidx);
            const u32 w1 = rdlane(l_coin, sidx);
            const u32 coin = w1 & 1u;
            const u64 idx = rdlane64(l_idx, sidx);

            fcm_acc_t<MAXT> delta[MAXT + 1];
#pragma unroll
            for (int q = 0; q <= MAXT; ++q) delta[q] = 0;
            EvScal es = {0, 0};

            bool nonempty = false, used_wide = false, used_xw = false;
            u32 t_sumk = 0u, t_changes = 0u, t_big = 0u, t_wide = 0u, pst = 0u, acc_inc = 0u;   // this proposal's share of the tallies
            u32 t_pairs = 0u, t_shared = 0u;
            // pending commit (uniform)
            u32 c_clr_from = 0, c_clr_to = 0, c_set_from = 0, c_set_to = 0;
            // the two bitmap words a commit rewrites, read while the build is in flight so that the
            // commit is two plain stores instead of two read-modify-write round trips
            u32 c_clr_old = 0, c_set_old = 0;
            bool c_have_old = false;
            u32 c_slot = 0, c_newdbl = 0;
            bool is_dmove = false;
            int clq_npairs = 0;
            long long wide_d = 0;

            if (move == 0) {
                // ---- single_edge_flip (src/lib.rs:292-299) -----------------
                if (Mtot > 0 && idx < U) {
                    const u32 a = rdlane(l_e.big, sidx), b = rdlane(l_e.small, sidx);
                    const u32 off = rdlane(l_e.nb_off, sidx);
                    const int k = (int)rdlane(l_e.k, sidx);
                    int res = FCM_NEEDS_WIDE;
                    // (tiny local sets: the build reads these very words, a read-modify-write then hits cache)
                    const bool pre = k + 2 > 8;
                    u32 w_ab = 0u, w_ba = 0u;
                    if (pre) { w_ab = rows[(size_t)a * stride32 + (b >> 5)]; w_ba = rows[(size_t)b * stride32 + (a >> 5)]; }
                    if (k + 2 <= WAVE) {
                        FCM_STAMP_AT(0);                               // decode
                        const u32 Lv = load_list(nb, off, k, a, b, lane);
                        FCM_STAMP_AT(1);                               // flip: vertex list round trip
#ifdef FCM_STAMP
                        {   // same work as flip_eval, with a stamp between build and count
                            u64 myH = build_local(rrows, stride32, Lv, k + 2, lane);
                            FCM_STAMP_AT(2);                           // flip: build
                            const u64 hk = rdlane64(myH, k), hk1 = rdlane64(myH, k + 1);
                            const u32 ab = (u32)((hk1 >> k) & 1ull), ba = (u32)((hk >> (k + 1)) & 1ull);
                            if (ab == ba) res = ab ? 0 : -1;
                            else {
                                const int iu = ab ? k : k + 1, iv = ab ? k + 1 : k;
                                Cls c = classify(myH, iv, iu);
                                Cls c2; c2.P = c.P; c2.S = c.S;
                                c2.M = (ab ? hk : hk1) & ballot((myH >> iv) & 1ull) & ~(3ull << k);
                                if (!extras_fit(c, k + 2) || !extras_fit(c2, k + 2)) res = FCM_NEEDS_WIDE;
                                else {
                                    eval_nodes<MAXT>(myH, Hp, c, k, tmax, -1, lane, delta, es, stamp_acc, &stamp_t);
                                    eval_nodes<MAXT>(myH, Hp, c2, k, tmax, +1, lane, delta, es, stamp_acc, &stamp_t);
                                    res = ab ? 1 : 2;
                                }
                            }
                            FCM_STAMP_AT(3);                           // flip: two evaluations
                        }
#else
                        res = flip_eval<MAXT>(rrows, stride32, Lv, k, Hp, lane, tmax, delta, es, &guard);
#endif
                        if (MAXT >= 7 && guard.tripped) {   // the bound on the 32-bit local counts was passed (it is generous): count this
                            guard.tripped = 0u;             // proposal with the wide evaluator, whose counts are 64-bit
#pragma unroll
                            for (int q = 0; q <= MAXT; ++q) delta[q] = 0;
                            es.d1 = es.d2 = 0;
                            res = FCM_NEEDS_WIDE;
                        }
                    }
                    if (res == FCM_NEEDS_WIDE) {
                        if (k + 2 <= 64 * maxnw) {
                            const Wide W = wide_carve(smem, maxnw);
                            wide_zero_counts(W, lane);
                            res = wide_flip(W, rows, stride32, nb, off, k, a, b, lane, tmax);
                            used_wide = true;
                        } else if (xw_ws && k + 2 <= 64 * FCM_XW_MAXNW) {   // 257..1024 local vertices: masks in the chain's workspace
                            res = xw_flip(xw_ws, rows, stride32, nb, off, k, a, b, lane, tmax);
                            used_wide = true; used_xw = true;
                        } else {
                            res = -1;
                        }
                    }
                    if (res < 0) pst |= 1u;  // table says adjacent, bitmap says not
                    if (res > 0) {
                        nonempty = true;
                        c_clr_from = res == 1 ? a : b; c_clr_to = res == 1 ? b : a;
                        c_set_from = c_clr_to; c_set_to = c_clr_from;
                        c_clr_old = res == 1 ? w_ab : w_ba; c_set_old = res == 1 ? w_ba : w_ab;
                        c_have_old = pre;
                        t_sumk += (u32)k;
                        if (k + 2 > 48) t_big = 1u;
                    }
                }
            } else if (move == 1) {
                // ---- double_edge_move (src/lib.rs:304-325) -----------------
                if (D > 0) {
                    FCM_STAMP_AT(0);
                    const u32 slot = (u32)idx;
                    const u32 ed = dbl[slot];                       // the live entry ...
                    FcmEdgeEntry de;                                // ... and the batch pass's guess of its pair
                    const u32 ed_guess = rdlane(l_ed, sidx);
                    de.big = rdlane(l_de.big, sidx); de.small = rdlane(l_de.small, sidx);
                    de.nb_off = rdlane(l_de.nb_off, sidx); de.k = rdlane(l_de.k, sidx);
                    u32 Lv1 = ((int)de.k + 2 <= WAVE) ? load_list(nb, de.nb_off, (int)de.k, de.big, de.small, lane) : 0u;
                    // Up to 64 candidate draws for the single edge (two per Philox block
                    // sub = 1..32), first valid in order wins (uniform directed edge, retry
                    // while reciprocal: :308-313).  Candidates 0 and 1 come from the batch
                    // draw.  Whether a candidate pair is single or reciprocal is read off the
                    // masks of its own local build (bits k<->k+1): no separate probe of the
                    // bitmap, and the build is the one step 2 needs anyway.
                    const u64 tt = sampled;  // this proposal's step index
                    u64 cand = rdlane64(l_c0, sidx);
                    u64 cand_next = rdlane64(l_c1, sidx);
                    FcmEdgeEntry ce;
                    ce.big = rdlane(l_e.big, sidx); ce.small = rdlane(l_e.small, sidx);
                    ce.nb_off = rdlane(l_e.nb_off, sidx); ce.k = rdlane(l_e.k, sidx);
                    bool have_ce = true, found = false;
                    u32 rfwd = 0u, Lv2 = 0u;
                    u64 myH2 = 0ull;
#pragma nounroll
                    for (int ci = 0; ci < WAVE && !found; ++ci) {
                        if (ci >= 2 && (ci & 1) == 0) {  // rare: draw block sub = ci/2 + 1 now
                            u32 v[4];
                            philox4x32_10((u32)tt, (u32)(tt >> 32), gchain, (u32)(ci >> 1) + 1u, k0, k1, v);
                            cand = __umul64hi((u64)v[0] | ((u64)v[1] << 32), Mtot);
                            cand_next = __umul64hi((u64)v[2] | ((u64)v[3] << 32), Mtot);
                        } else if (ci >= 1) {
                            cand = cand_next;
                        }
                        if (cand < U) {
                            if (!have_ce) ce = p.etab[cand];
                            const int ck = (int)ce.k;
                            if (ck + 2 <= WAVE) {
                                Lv2 = load_list(nb, ce.nb_off, ck, ce.big, ce.small, lane);
                                myH2 = build_local(rrows, stride32, Lv2, ck + 2, lane);
                                const u32 f = (u32)(rdlane64(myH2, ck + 1) >> ck) & 1u, bwd = (u32)(rdlane64(myH2, ck) >> (ck + 1)) & 1u;
                                if (!(f | bwd)) pst |= 1u;
                                found = (f ^ bwd) != 0u;
                                rfwd = f;
                            } else {  // wide candidate: look at its two words directly
                                const u32 wf = rows[(size_t)ce.big * stride32 + (ce.small >> 5)];
                                const u32 wb = rows[(size_t)ce.small * stride32 + (ce.big >> 5)];
                                const u32 f = (wf >> (ce.small & 31u)) & 1u, bwd = (wb >> (ce.big & 31u)) & 1u;
                                found = (f ^ bwd) != 0u;
                                rfwd = f;
                            }
                        }
                        have_ce = false;
                    }
                    FCM_STAMP_AT(4);                                   // double move: lists + candidate build
                    if (found) {
                        if (ed != ed_guess) {   // the slot was rewritten since the batch draw: take the live pair
                            de = p.etab[ed];
                            Lv1 = ((int)de.k + 2 <= WAVE) ? load_list(nb, de.nb_off, (int)de.k, de.big, de.small, lane) : 0u;
                        }
                        const u32 r = (u32)cand;
                        const u32 rbig = ce.big, rsmall = ce.small, roff = ce.nb_off;
                        const int rk = (int)ce.k;
                        const u32 ea = rfwd ? rbig : rsmall, eb = rfwd ? rsmall : rbig;  // ea->eb is the single edge
                        // delme: coin ? (big->small) : (small->big) of the reciprocal pair (:316-320)
                        const u32 dfrom = coin ? de.big : de.small, dto = coin ? de.small : de.big;
                        nonempty = true; is_dmove = true;
                        const int dk = (int)de.k;
                        if (dk + 2 > 8 || (int)ce.k + 2 > 8) {
                            c_clr_old = rows[(size_t)dfrom * stride32 + (dto >> 5)];
                            c_set_old = rows[(size_t)eb * stride32 + (ea >> 5)];
                            c_have_old = true;
                        }
                        bool go_wide = dk + 2 > WAVE || rk + 2 > WAVE;
                        bool okd = true;
                        if (!go_wide) {
                            // (1) remove delme: subtract simplices through it
                            const int r1 = del_eval<MAXT>(rrows, stride32, Lv1, dk, coin, Hp, lane, tmax, delta, es, &guard);
                            go_wide = r1 == FCM_NEEDS_WIDE;
                            okd = r1 != 0;
                            if (!go_wide) {
                                // (2) add eb->ea on the graph without delme: add simplices through it
                                const int r2 = add_eval_built<MAXT>(myH2, Lv2, rk, rfwd, dfrom, dto, Hp, lane, tmax, delta, es, &guard);
                                go_wide = r2 == FCM_NEEDS_WIDE;
                            }
                            if (MAXT >= 7 && guard.tripped) { guard.tripped = 0u; go_wide = true; }   // (as for flips: 64-bit counts on the wide path)
                        }
                        if (go_wide) {
#pragma unroll
                            for (int q = 0; q <= MAXT; ++q) delta[q] = 0;
                            es.d1 = es.d2 = 0;
                            if (dk + 2 > 64 * maxnw || rk + 2 > 64 * maxnw) {
                                if (xw_ws && dk + 2 <= 64 * FCM_XW_MAXNW && rk + 2 <= 64 * FCM_XW_MAXNW) {
                                    okd = xw_del(xw_ws, rows, stride32, nb, de.nb_off, dk, de.big, de.small, coin, lane, tmax);
                                    xw_add(xw_ws, rows, stride32, nb, roff, rk, rbig, rsmall, rfwd, dfrom, dto, lane, tmax, true);
                                    used_wide = true; used_xw = true;
                                } else {
                                    pst |= 1u;
                                }
                            } else {
                                const Wide W = wide_carve(smem, maxnw);
                                wide_zero_counts(W, lane);
                                okd = wide_del(W, rows, stride32, nb, de.nb_off, dk, de.big, de.small, coin, lane, tmax);
                                wide_add(W, rows, stride32, nb, roff, rk, rbig, rsmall, rfwd, dfrom, dto, lane, tmax);
                                used_wide = true;
                            }
                        }
                        FCM_STAMP_AT(5);                               // double move: second build + two evaluations
                        if (!okd) pst |= 2u;  // slot list says reciprocal, bitmap says not
                        c_clr_from = dfrom; c_clr_to = dto;
                        c_set_from = eb; c_set_to = ea;
                        c_slot = slot; c_newdbl = r;
                        t_sumk += de.k + (u32)rk;
                        if (dk + 2 > 48 || rk + 2 > 48) t_big = 1u;
                    }
                }
            } else {
                // ---- clique_permute / clique_swap (src/lib.rs:214-290) -------
                if constexpr (CLIQUE != 0) {
                    const CliqueLds CL = clique_carve(smem + fcm_lds_words(maxnw));
#ifdef FCM_STAMP
                    u64 *clq_sacc = stamp_acc, *clq_stt = &stamp_t;
#else
                    u64 *clq_sacc = nullptr, *clq_stt = nullptr;
#endif
                    const CliqueResult cr = clique_propose<MAXT, CLIQUE == 2>(p, rows, rrows, smem, CL, move, w1, idx, sampled, gchain, k0, k1, lane, tmax, maxnw, delta, es, clq_sacc, clq_stt, &guard);
                    pst |= cr.status;
                    t_wide += cr.n_wide;
                    if (cr.nchg > 0) {
                        nonempty = true;
                        clq_npairs = cr.npairs;
                        wide_d = cr.wide_d;
                        t_sumk += (u32)cr.sum_k;
                        t_changes += (u32)cr.nchg;
                        if (cr.npairs > 0) { t_pairs = (u32)cr.npairs; t_shared = move == 2 ? (u32)(cr.npairs - 1) * (u32)cr.n_d : 0u; }   // (a permutation's pairs all lie inside the clique)
                    }
                } else {
                    pst |= 4u;  // this kernel variant was built without the clique moves
                }
            }

            FCM_STAMP_AT(0);
            // ---- sampled += 1; Bounds::check; accept or drop ---------------
            sampled += 1;
            u64 nzm = 0ull;
            if (!nonempty) {
                if (in_bounds) acc_inc = 1u;
            } else {
                long long myd = 0;
                if (used_xw) {
                    t_wide += 1u;
                    if (lane >= 2 && lane < 16 && lane - 1 <= tmax) myd = xw_count(xw_ws, lane - 1);
                    wave_sync();
                } else if (used_wide) {
                    t_wide += 1u;
                    const Wide W = wide_carve(smem, maxnw);
                    if (lane >= 2 && lane < 16 && lane - 1 <= tmax) myd = W.cnt[lane - 1];
                    wave_sync();
                } else {
                    fcm_lane_guard<MAXT>(delta, guard);
                    if (lane == 2) myd = (long long)es.d1;   // levels 1 and 2: the evaluations' node and arc counts (scalars)
                    if (lane == 3 && tmax >= 2) myd = (long long)es.d2;
#pragma unroll
                    for (int tq = 3; tq <= MAXT; ++tq) {
                        if (tq <= tmax) {
                            const fcm_acc_t<MAXT> sum = wave_sum_acc(delta[tq]);
                            if (lane == tq + 1) myd = (long long)sum;
                        }
                    }
                }
                myd += wide_d;  // clique moves: evaluations that went through the wide path
                const int el = lane < 16 ? lane : 15;   // (lanes >= NC hold {0, 0, ~0})
                const u64 cnt = ent[el * 3], bmin = ent[el * 3 + 1], bmax = ent[el * 3 + 2];
                const u64 ncnt = cnt + (u64)myd;
                if (ballot(cl && myd < 0 && cnt < (u64)(-myd))) pst |= 8u;  // reference assert, src/lib.rs:65
                // flag_count never shrinks in length (src/lib.rs:72-74)
                nzm = ballot(cl && ncnt != 0ull);
                const bool ok = ballot(cl && (ncnt < bmin || ncnt > bmax)) == 0ull;
                if (ok) {
                    acc_inc = 1u;
                    in_bounds = true;
                    if (cl) ent[lane * 3] = ncnt;
                    if (move >= 2) {
                        if constexpr (CLIQUE != 0) {  // bits are already in place; hand over the reciprocal-pair slots
                            const CliqueLds CL = clique_carve(smem + fcm_lds_words(maxnw));
                            pst |= clique_update_slots(dbl, slot_of, CL, clq_npairs, lane);
                        }
                    } else {
                        if (lane == 0) {
                            u32 *pc = rows + (size_t)c_clr_from * stride32 + (c_clr_to >> 5);
                            u32 *ps = rows + (size_t)c_set_from * stride32 + (c_set_to >> 5);
                            const u32 bc = 1u << (c_clr_to & 31u), bs = 1u << (c_set_to & 31u);
                            if (c_have_old) {
                                if (pc == ps) {   // both changes in one word (double-edge move only)
                                    *pc = (c_clr_old & ~bc) | bs;
                                } else {
                                    *pc = c_clr_old & ~bc;
                                    *ps = c_set_old | bs;
                                }
                            } else {
                                *pc &= ~bc;
                                *ps |= bs;
                            }
                            if (is_dmove) {
                                if constexpr (CLIQUE != 0) {
                                    slot_of[dbl[c_slot]] = FCM_NOSLOT;
                                    slot_of[c_newdbl] = c_slot;
                                }
                                dbl[c_slot] = c_newdbl;
                            }
                        }
                        wave_sync();
                    }
                } else if (move >= 2) {
                    if constexpr (CLIQUE != 0) {
                        const CliqueLds CL = clique_carve(smem + fcm_lds_words(maxnw));
                        clique_revert(rows, stride32, CL, clq_npairs, lane);
                    }
                }
            }
            {   // this proposal's tallies: lanes 0..9 and 12..13 add, lanes 10..11 OR (all 64 lanes are active here: uniform branches only)
                const u32 kind = !nonempty ? (1u << OT_EMPTY) : (move == 2 ? (1u << OT_CPERM) : (move == 3 ? (1u << OT_CSWAP) : (is_dmove ? (1u << OT_DMOVE) : (1u << OT_FLIP))));
                const u64 im = (u64)(kind | (acc_inc << OT_ACCEPTED) | (t_big << OT_BIG));
                u32 inc = lane_in(im) ? 1u : 0u;
                inc = lane_in(1ull << OT_WIDE) ? t_wide : inc;
                inc = lane_in(1ull << OT_SUMK) ? t_sumk : inc;
                inc = lane_in(1ull << OT_CHANGES) ? t_changes : inc;
                if constexpr (CLIQUE != 0) {
                    inc = lane_in(1ull << OT_PAIRS) ? t_pairs : inc;
                    inc = lane_in(1ull << OT_SHARED) ? t_shared : inc;
                }
                const u32 orv = lane_in(1ull << OT_NZ) ? (u32)nzm : pst;
                const u32 taddr = mw_lds_addr_of(tly) + (u32)lane * 4u;
                asm volatile("s_mov_b64 exec, 0x33ff\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, 0xc00\n\tds_or_b32 %0, %2\n\ts_mov_b64 exec, -1"
                             :: "v"(taddr), "v"(inc), "v"(orv) : "memory");
            }
            FCM_STAMP_AT(6);                                           // reductions, bounds, commit
        }
        FCM_STAMP_AT(7);                                               // (batch boundary)
    }

    wave_sync();
    if (cl) cnt_g[lane] = ent[lane * 3];
    const u32 tl = lane < 16 ? tly[lane] : 0u;
    const u32 nzall = rdlane(tl, OT_NZ);
    const u32 nlen = nzall ? (u32)(32 - __clz((int)nzall)) : 0u;
    const u32 stw = rdlane(tl, OT_STATUS) | (guard.tripped ? 256u : 0u);   // (guard: a local count may have passed 2^31: refuse rather than wrap)
    if (lane == 0) {
        st_g[0] = sampled; st_g[1] += rdlane(tl, OT_ACCEPTED); st_g[2] += rdlane(tl, OT_EMPTY); st_g[3] += rdlane(tl, OT_FLIP);
        st_g[4] += rdlane(tl, OT_DMOVE); st_g[5] += rdlane(tl, OT_SUMK); if (nlen > st_g[6]) st_g[6] = nlen; st_g[7] |= stw;
        st_g[8] += rdlane(tl, OT_CPERM); st_g[9] += rdlane(tl, OT_CSWAP); st_g[10] += rdlane(tl, OT_CHANGES);
        st_g[12] += rdlane(tl, OT_WIDE); st_g[13] += rdlane(tl, OT_BIG);
        if constexpr (CLIQUE != 0) { st_g[16] += rdlane(tl, OT_PAIRS); st_g[17] += rdlane(tl, OT_SHARED); }   // (FCM_STAT_PAIRS, FCM_STAT_SHARED_ROWS: slots of their own)
#ifdef FCM_STAMP
        for (int q = 0; q < 8; ++q) p.dbgbuf[(size_t)chain * 8 + q] += stamp_acc[q];
#endif
    }
}
