// fcm_step_cq.hpp — the step kernel for move mixes WITH clique moves (reference default [0.1, 0.1, 0.6, 0.2],
// src/bin/sample.rs:17,101; clique_permute / clique_swap, src/lib.rs:214-290).  Included by fcm_step_variant.hip for
// the c<depth> (rows of one cache line) and d<depth> (longer rows) tags; 2 <= depth <= 6, like the multi-wave kernel.
//
// One 64-lane workgroup per chain.  What differs from the one-wave kernel's clique path (fcm_clique.hpp, clique_propose):
//
//  * Every changed vertex pair of a move is evaluated on the PRE-MOVE bitmap.  The reference applies a move's change
//    edges one by one and recounts in between (State::apply_transition, src/lib.rs:61-79); the telescoping sum needs
//    pair x counted on the graph "pairs before x already changed".  Every earlier pair lies in the same clique(s), so
//    wherever it matters both its endpoints are in pair x's local set: the earlier changes are XORed into the local
//    masks in registers (cq_patch) instead of being stored to memory and read back.  Nothing is written until the move
//    is accepted -- one pass of stores, all pairs at once -- and a rejected move needs no revert (the reference's
//    revert_transition, :81-95, has nothing to undo).
//  * The pairs' builds no longer depend on each other through memory, so they are software-pipelined: while pair x is
//    evaluated, the rows of pair x+1 are in flight (a fixed 24 whole-row loads per pair, so that the wait counts are
//    static) and the vertex list of pair x+2 is on its way.
//  * Lean register allocation: the chain's context lives in LDS, the simple moves (20 % of the default mix) run
//    out of line through the multi-wave kernel's exact run (mw_exact_call), so the pair loop is all the hot code
//    there is.  (In the one-wave kernel a quarter of the VALU instructions were reloads of spilled scalars.)
//
// Local sets beyond 64 vertices, split graphs that do not fit 64 nodes and local sets of 257..1024 vertices take the
// wide / workspace evaluators pair by pair, with the same patches applied to their masks.
#pragma once
#include <type_traits>

// tallies: the OT_* words of the one-wave kernel (fcm_kernels_common.hpp)
#define CQ_TALLY_WORDS 8u   // u64 words: OT_* (12 u32), then the last setup's nchg, npairs, status, n_d
#define CQ_ORDERS 8         // clique orders the kernel takes (<= 8 count entries means cliques of <= 8 vertices)
#define CQ_TABLE_WORDS (4u * CQ_ORDERS + 4u)   // u64 words: cl_base | cl_count | clp_base | cumo | clq, clq_pairs, (cl_orders, chg_cap), spare
__host__ __device__ constexpr inline unsigned fcm_cq_lds_words(int NW, unsigned chg_cap)
{
    return fcm_mw_lds_words(NW, 1) + fcm_clique_lds_words(chg_cap) + CQ_TALLY_WORDS + CQ_TABLE_WORDS;
}
struct CqLds {
    u64 *mine, *wide, *tables;
    u32 *tly;
    CliqueLds CL;
};
// (the tables and the tallies in front of the clique arrays: their place does not depend on the pair list's capacity)
__device__ __forceinline__ CqLds cq_carve(u64 *smem, int maxnw)
{
    CqLds L;
    L.mine = smem + MW_SHARED_WORDS + MW_RING_WORDS(1u);
    L.wide = L.mine + MW_WAVE_WORDS;
    L.tables = L.wide + fcm_lds_words(maxnw);
    L.tly = (u32 *)(L.tables + CQ_TABLE_WORDS);
    L.CL = clique_carve(L.tables + CQ_TABLE_WORDS + CQ_TALLY_WORDS);
    return L;
}
enum { CS_NCHG = 12, CS_NPAIRS, CS_STATUS, CS_ND };   // u32 words of the tally block the setup leaves its scalars in

// ---- out of line: the proposal of a clique move (clique_setup) with its context from LDS; what it finds goes to the
// clique arrays (pair list in CL.chg, d in CL.d, the in-masks over d before / after the move in CL.oldm / CL.newm) and
// to the CS_* words.
__device__ __attribute__((noinline)) void cq_setup_call(u64 *smem, u32 tv, u32 q)
{
    const int lane = threadIdx.x & (WAVE - 1);
    q = mw_uni(q);
    const u32 *ctx = (const u32 *)(smem + MW_CTX_OFF);
    const u32 cv = lane < MC_WORDS ? ctx[lane] : 0u;
    const int maxnw = (int)rdlane(cv, MC_MAXNW);
    const CqLds L = cq_carve(smem, maxnw);
    const u64 oc = mw_uni64(L.tables[4 * CQ_ORDERS + 2]);
    const u32 chg_cap = (u32)(oc >> 32);
    const MwChain C = mw_chain_from_lds(ctx, lane);
    const u64 seed = (u64)rdlane(cv, MC_SEED) | ((u64)rdlane(cv, MC_SEED + 1) << 32);
    const u64 sampled0 = (u64)rdlane(cv, MC_SAMPLED0) | ((u64)rdlane(cv, MC_SAMPLED0 + 1) << 32);
    CliqueTables T;
    T.cl_base = (const uint64_t *)L.tables; T.cl_count = T.cl_base + CQ_ORDERS; T.clp_base = T.cl_base + 2 * CQ_ORDERS; T.cumo = T.cl_base + 3 * CQ_ORDERS;
    T.clq = (const u32 *)mw_uni64(L.tables[4 * CQ_ORDERS]); T.clq_pairs = (const u32 *)mw_uni64(L.tables[4 * CQ_ORDERS + 1]);
    T.cl_orders = (int)(u32)oc;
    T.etab = C.etab; T.chg_cap = chg_cap; T.stride32 = C.stride32;
    const int move = (int)(rdlane(tv, 0) & 0xFFu);
    const u32 w1 = rdlane(tv, 1);
    const u64 x64 = (u64)rdlane(tv, 2) | ((u64)rdlane(tv, 3) << 32);
    const CliqueResult cr = clique_setup(T, C.rows, L.CL, move, w1, x64, sampled0 + q, rdlane(cv, MC_GCHAIN), (u32)seed, (u32)(seed >> 32), lane, nullptr, nullptr);
    wave_sync();
    if (lane < 32) { L.CL.oldm[lane] = cr.oldt; L.CL.newm[lane] = cr.newt; }   // (the rows themselves are not needed any more)
    if (lane == 0) { L.tly[CS_NCHG] = (u32)cr.nchg; L.tly[CS_NPAIRS] = (u32)cr.npairs; L.tly[CS_STATUS] = cr.status; L.tly[CS_ND] = (u32)cr.n_d; }
    wave_sync();
}

// ---- whole-row build, split into its two halves for pipelining (n <= 1024: rows of one cache line) -------------------
// issue: 24 loads, always -- rows beyond the list re-read the row of its last vertex (the list's lanes beyond s repeat
// `small`), a line the cache holds: a fixed number of loads in flight keeps the compiler's s_waitcnt counts static
template <int Q = 0>
__device__ __forceinline__ void cq_issue_rows(const rsrc_t rsrc, u32 Lv, u32 sel, u32 dw, u32 (&w)[24])
{
    if constexpr (Q < 24) {
        const u32 v = (u32)__builtin_amdgcn_ds_bpermute((int)(sel + 8u * Q), (int)Lv);
        w[Q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (v << 7) + dw, 0, 0);
        cq_issue_rows<Q + 1>(rsrc, Lv, sel, dw, w);
    }
}
// consume: rows 0..47 of the local adjacency (those below s); rows 48..63 in a second trip of their own
__device__ __forceinline__ u64 cq_consume_rows(const rsrc_t rsrc, const u32 (&w)[24], u32 Lv, u32 sel, u32 dw, int s, int lane)
{
    const u32 src = (Lv >> 5) * 4u, bpos = Lv & 31u;
    u32 hlo = 0u, hhi = 0u;
    build128_consume<0, 0, 6, 24>(w, src, bpos, s, hlo, hhi);
    if (s > 48) {
        u32 w2[8];
        build_regs_any(w2);
        build128_issue<6, 6, 8, 8>(rsrc, Lv, sel, dw, s, w2);
        build128_consume<6, 6, 8, 8>(w2, src, bpos, s, hlo, hhi);
    }
    const u64 h = (u64)hlo | ((u64)hhi << 32);
    return lane < s ? (h & (s >= 64 ? ~0ull : ((1ull << s) - 1ull))) : 0ull;
}

// The rows in flight are carried around the pair loop as 24 separate scalars: carried as an array, hipcc's SROA turns
// them into ONE 24-element vector value (a 32-register tuple with a phi at the loop head) and every element write
// copies the tuple through scratch.
struct CqRows { u32 a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16, a17, a18, a19, a20, a21, a22, a23; };
__device__ __forceinline__ void cq_rows_get(const CqRows &r, u32 (&w)[24])
{
    w[0] = r.a0; w[1] = r.a1; w[2] = r.a2; w[3] = r.a3; w[4] = r.a4; w[5] = r.a5; w[6] = r.a6; w[7] = r.a7;
    w[8] = r.a8; w[9] = r.a9; w[10] = r.a10; w[11] = r.a11; w[12] = r.a12; w[13] = r.a13; w[14] = r.a14; w[15] = r.a15;
    w[16] = r.a16; w[17] = r.a17; w[18] = r.a18; w[19] = r.a19; w[20] = r.a20; w[21] = r.a21; w[22] = r.a22; w[23] = r.a23;
}
__device__ __forceinline__ void cq_rows_put(CqRows &r, const u32 (&w)[24])
{
    r.a0 = w[0]; r.a1 = w[1]; r.a2 = w[2]; r.a3 = w[3]; r.a4 = w[4]; r.a5 = w[5]; r.a6 = w[6]; r.a7 = w[7];
    r.a8 = w[8]; r.a9 = w[9]; r.a10 = w[10]; r.a11 = w[11]; r.a12 = w[12]; r.a13 = w[13]; r.a14 = w[14]; r.a15 = w[15];
    r.a16 = w[16]; r.a17 = w[17]; r.a18 = w[18]; r.a19 = w[19]; r.a20 = w[20]; r.a21 = w[21]; r.a22 = w[22]; r.a23 = w[23];
}

// ---- the earlier pairs' changes, XORed into the local in-masks ------------------------------------------------------
// myH: lane j's in-mask over the local list Lv (bit i = L[i] -> L[j]) as read from the pre-move bitmap.  Pair x =
// (xi, xj) in d indices, xi < xj; the pairs are listed ascending (i, j), so "before x" = (i, j) < (xi, xj).  oldt / newt:
// lane t's in-masks over d (bit u = d[u] -> d[t]) before and after the whole move; dvv: d[lane].
// For every d-vertex u that is in the list: the lanes of the d-vertices t whose edge u -> t changed before x flip bit
// pos(u).
__device__ __forceinline__ u64 cq_patch(u64 myH, u32 Lv, int s, u32 dvv, u32 oldt, u32 newt, int n_d, int xi, int xj, int lane)
{
    // chg_in[t]: bits u with (pair {t,u} before x) and (edge u -> t changed)
    const u32 lt_i = (1u << xi) - 1u, lt_j = (1u << xj) - 1u;
    const u32 M = lane < xi ? 0xFFFFFFFFu : (lane == xi ? lt_j : (lt_i | (lane < xj ? (1u << xi) : 0u)));
    const u32 chg_in = (oldt ^ newt) & M;
    if (ballot(chg_in != 0u) == 0ull) return myH;                 // nothing before x changed (always so for the first pair)
    // tj: this lane's d index, 32 if its vertex is not in d; pos: lane u <- the list position of d[u] | 0x100
    u32 tj = 32u;
    for (int u = 0; u < n_d; ++u) tj = (lane < s && Lv == rdlane(dvv, u)) ? (u32)u : tj;
    const u32 posv = (u32)__builtin_amdgcn_ds_permute((int)((tj < 32u ? tj : 63u) * 4u), (int)((u32)lane | 0x100u));
    // this lane's changes as a target, fetched from lane tj
    const u32 mine = (u32)__builtin_amdgcn_ds_bpermute((int)((tj & 31u) * 4u), (int)chg_in);
    u32 xlo = 0u, xhi = 0u;
    for (int u = 0; u < n_d; ++u) {
        const u32 pu = rdlane(posv, u);
        if (!(pu & 0x100u)) continue;                             // d[u] is not in this pair's local set: no simplex through the pair holds it
        const u32 b = __builtin_amdgcn_ubfe(mine, (u32)u, 1u);
        const u32 pos = pu & 63u;
        if (pos < 32u) xlo |= b << pos; else xhi |= b << (pos - 32u);
    }
    const bool isd = tj < 32u;
    return myH ^ (isd ? ((u64)xlo | ((u64)xhi << 32)) : 0ull);
}

// per-pair scalars, read off the pair list (lane x of the chunk holds pair x)
struct CqPair {
    u32 big, small, off;
    int k, xi, xj;
    u32 o_bs, o_sb, n_bs, n_sb;
};
__device__ __forceinline__ CqPair cq_pair_at(u32 pw0, u32 pk, u32 poff, u32 pa, u32 pb, int x)
{
    CqPair P;
    const u32 w0 = rdlane(pw0, x), a = rdlane(pa, x), b = rdlane(pb, x);
    P.k = (int)rdlane(pk, x); P.off = rdlane(poff, x);
    P.xi = (int)(w0 & 0xFFu); P.xj = (int)((w0 >> 8) & 0xFFu);
    const u32 o2 = (w0 >> 16) & 3u, n2 = (w0 >> 20) & 3u;
    const bool agb = a > b;
    P.big = agb ? a : b; P.small = agb ? b : a;
    P.o_bs = agb ? (o2 & 1u) : (o2 >> 1); P.o_sb = agb ? (o2 >> 1) : (o2 & 1u);
    P.n_bs = agb ? (n2 & 1u) : (n2 >> 1); P.n_sb = agb ? (n2 >> 1) : (n2 & 1u);
    return P;
}

// position of vertex v in a list of s vertices in LDS, or -1
__device__ __forceinline__ int cq_find(const u32 *L, int s, u32 v, int lane)
{
    for (int base = 0; base < s; base += WAVE) {
        const int j = base + lane;
        const u64 m = ballot(j < s && L[j] == v);
        if (m) return base + __ffsll((long long)m) - 1;
    }
    return -1;
}

// A pair that the fast evaluator does not take (more than 64 local vertices, or a split graph of more than 64 nodes):
// the wide evaluator (masks in LDS) or, beyond 256 vertices, the one with its masks in the chain's HBM workspace, on
// masks built from the pre-move bitmap with the changes of the pairs before x set into them.  Counts go to the
// evaluator's 64-bit counters, which the caller has zeroed for this move and reads at its end.
template <bool XW>
__device__ __attribute__((noinline)) u32 cq_wide_pair(u64 *wide_lds, int maxnw, u64 *xw, const u32 *rows, u32 stride32, const u32 *nb, const CliqueLds CL, int x,
                                         u32 big, u32 small, u32 off, int k, u32 dirs, int tmax)
{
    const int lane = threadIdx.x & (WAVE - 1);
    x = (int)mw_uni((u32)x); k = (int)mw_uni((u32)k); dirs = mw_uni(dirs); big = mw_uni(big); small = mw_uni(small); off = mw_uni(off);
    const int s = k + 2;
    const u32 n_bs = (dirs >> 2) & 1u, n_sb = (dirs >> 3) & 1u;
    const bool need_bs = ((dirs ^ (dirs >> 2)) & 1u) != 0u, need_sb = (((dirs >> 1) ^ (dirs >> 3)) & 1u) != 0u;
    u32 status = 0u;
    if (s <= 64 * maxnw) {
        const Wide W = wide_carve(wide_lds, maxnw);
        wide_load_list(W, nb, off, k, big, small, lane);
        wide_build(W, rows, stride32, s, lane);
        for (int y = 0; y < x; ++y) {                              // the pairs before x, as the move leaves them
            const u32 w0 = CL.chg[4 * y];
            const int fi = cq_find(W.L, s, CL.d[w0 & 0xFFu], lane), fj = cq_find(W.L, s, CL.d[(w0 >> 8) & 0xFFu], lane);
            if (fi < 0 || fj < 0) continue;
            const u32 n2 = (w0 >> 20) & 3u;
            wide_set(W, fi, fj, (n2 & 1u) != 0u);
            wide_set(W, fj, fi, (n2 & 2u) != 0u);
        }
        wave_sync();
        for (int dir = 0; dir < 2; ++dir) {
            if (!(dir == 0 ? need_bs : need_sb)) continue;
            const int iu = dir == 0 ? k : k + 1, iv = dir == 0 ? k + 1 : k;
            wide_set(W, iu, iv, true);                             // the edge is there while it is counted (the other direction does not matter)
            wave_sync();
            wide_classify(W, iu, iv, s, lane);
            wide_dfs<false>(W, tmax, (dir == 0 ? n_bs : n_sb) ? +1 : -1, nullptr);
        }
    } else if (XW && xw && s <= 64 * FCM_XW_MAXNW) {
        const XWide X = xw_carve(xw, s);
        long long keep = 0;
        if (lane < 16) keep = X.cnt[lane];
        xw_load_list(X, nb, off, k, big, small, lane);            // (zeroes the counters: put this move's back)
        if (lane < 16) X.cnt[lane] = keep;
        xw_sync();
        xw_build(X, rows, stride32, s, lane);
        for (int y = 0; y < x; ++y) {
            const u32 w0 = CL.chg[4 * y];
            const int fi = cq_find(X.L, s, CL.d[w0 & 0xFFu], lane), fj = cq_find(X.L, s, CL.d[(w0 >> 8) & 0xFFu], lane);
            if (fi < 0 || fj < 0) continue;
            const u32 n2 = (w0 >> 20) & 3u;
            xw_set(X, fi, fj, (n2 & 1u) != 0u, lane);
            xw_set(X, fj, fi, (n2 & 2u) != 0u, lane);
        }
        for (int dir = 0; dir < 2; ++dir) {
            if (!(dir == 0 ? need_bs : need_sb)) continue;
            const int iu = dir == 0 ? k : k + 1, iv = dir == 0 ? k + 1 : k;
            xw_set(X, iu, iv, true, lane);
            xw_dfs(X, xw_classify(X, iu, iv, s, lane), tmax, (dir == 0 ? n_bs : n_sb) ? +1 : -1, lane);
        }
    } else {
        status = 1u;
    }
    return status;
}

// The directed changes of an accepted move, all pairs at once (lanes over pairs; atomics: two pairs may share a word).
__device__ __forceinline__ void cq_commit(u32 *rows, u32 stride32, const CliqueLds CL, int npairs, int lane)
{
    for (int x = lane; x < npairs; x += WAVE) {
        const u32 w0 = CL.chg[4 * x];
        const u32 a = CL.d[w0 & 0xFFu], b = CL.d[(w0 >> 8) & 0xFFu];
        const u32 n2 = (w0 >> 20) & 3u, ch = n2 ^ ((w0 >> 16) & 3u);
        if (ch & 1u) {
            u32 *word = rows + (size_t)a * stride32 + (b >> 5);
            const u32 bit = 1u << (b & 31u);
            if (n2 & 1u) atomicOr(word, bit); else atomicAnd(word, ~bit);
        }
        if (ch & 2u) {
            u32 *word = rows + (size_t)b * stride32 + (a >> 5);
            const u32 bit = 1u << (a & 31u);
            if (n2 & 2u) atomicOr(word, bit); else atomicAnd(word, ~bit);
        }
    }
    wave_sync();
}

struct CqEval {
    u64 sum_k;
    u32 status, n_wide, any_wide, any_xw;
};

// one pair on the fast evaluator: classes around (big, small) from the patched in-masks, one evaluation per changed
// direction.  Returns false if the split graph of a needed direction does not fit 64 nodes.
template <int MAXT>
__device__ __forceinline__ bool cq_eval_pair(u64 myH, u64 *Hp, const CqPair &P, int tmax, int lane, int (&delta)[MAXT + 1], EvScal &es, FcmGuard *guard, u32 &status)
{
    const int k = P.k, s = k + 2;
    const bool need_bs = P.o_bs != P.n_bs, need_sb = P.o_sb != P.n_sb;
    // in-masks: the evaluator runs on the transposed graph (build_local)
    const u64 inB = rdlane64(myH, k), inS = rdlane64(myH, k + 1);
    if ((u32)((inS >> k) & 1ull) != P.o_bs || (u32)((inB >> (k + 1)) & 1ull) != P.o_sb) status |= 1u;   // the bitmap and the move's OLD disagree
    const u64 outB = ballot_bit(myH, k), outS = ballot_bit(myH, k + 1);
    const u64 nbm = ~(3ull << k);
    Cls cbs, csb;   // classes around big->small and around small->big, as classify(.., k+1, k) / (.., k, k+1) give them
    cbs.P = csb.P = outB & outS & nbm;
    cbs.S = csb.S = inB & inS & nbm;
    cbs.M = inS & outB & nbm;
    csb.M = inB & outS & nbm;
    if ((need_bs && !extras_fit(cbs, s)) || (need_sb && !extras_fit(csb, s))) return false;
#pragma nounroll
    for (int dir = 0; dir < 2; ++dir) {   // (a loop over one inlined evaluator: the classes differ in M only)
        if (!(dir == 0 ? need_bs : need_sb)) continue;
        Cls c = cbs;
        c.M = dir == 0 ? cbs.M : csb.M;
        eval_nodes<MAXT>(myH, Hp, c, k, tmax, (dir == 0 ? P.n_bs : P.n_sb) ? +1 : -1, lane, delta, es, nullptr, nullptr, guard);
    }
    return true;
}

// All changed pairs of a clique move (CL.chg, npairs of them), on the pre-move bitmap.
template <int MAXT, bool ROWS128, bool XW>
__device__ __forceinline__ CqEval cq_pairs(const MwChain &C, u64 *Hp, u64 *wide_lds, int maxnw, const CliqueLds CL, int npairs, int n_d,
                                           int lane, int (&delta)[MAXT + 1], EvScal &es, FcmGuard *guard)
{
    CqEval R = {0ull, 0u, 0u, 0u, 0u};
    const int tmax = MAXT;
    const rsrc_t rr = make_rows_rsrc(C.rows, C.rows_bytes);
    const u32 stride32 = ROWS128 ? 32u : C.stride32;
    const u32 dvv = lane < n_d ? CL.d[lane] : 0xFFFFFFFFu;
    const u32 oldt = lane < 32 ? CL.oldm[lane] : 0u, newt = lane < 32 ? CL.newm[lane] : 0u;   // in-masks over d before / after the move (cq_setup_call)
    u32 sel = lane >= 32 ? 4u : 0u;
    asm volatile("" : "+v"(sel));
    const u32 dw = (u32)(lane & 31) * 4u;
    for (int base = 0; base < npairs; base += WAVE) {
        const int cnt = min(WAVE, npairs - base);
        // lane x of the chunk: pair base + x
        u32 pw0 = 0u, pk = 0u, poff = 0u, pa = 0u, pb = 0u;
        if (lane < cnt) {
            const uint4 c = *(const uint4 *)(CL.chg + 4 * (base + lane));
            pw0 = c.x; pk = c.z; poff = c.w;
            pa = CL.d[c.x & 0xFFu]; pb = CL.d[(c.x >> 8) & 0xFFu];
        }
        auto slow = [&](const CqPair &P, int x) {
            const u32 dirs = P.o_bs | (P.o_sb << 1) | (P.n_bs << 2) | (P.n_sb << 3);
            if (!R.any_wide) {
                const Wide W = wide_carve(wide_lds, maxnw);
                wide_zero_counts(W, lane);
                if (XW && C.xw) { if (lane < 16) xw_carve(C.xw, 64).cnt[lane] = 0; xw_sync(); }
                R.any_wide = 1u;
            }
            if (P.k + 2 > 64 * maxnw) R.any_xw = 1u;
            R.status |= cq_wide_pair<XW>(wide_lds, maxnw, C.xw, C.rows, stride32, C.nb, CL, base + x, P.big, P.small, P.off, P.k, dirs, tmax);
            R.n_wide += 1u;
        };
        auto tally_k = [&](const CqPair &P) { R.sum_k += (u64)P.k * (u64)((P.o_bs != P.n_bs ? 1 : 0) + (P.o_sb != P.n_sb ? 1 : 0)); };
        if constexpr (ROWS128) {
            // Software pipeline, one row buffer: as soon as pair x's rows have been consumed, the rows of pair x+1 are
            // requested into the same registers and the list of pair x+2 after them; both round trips run behind pair x's
            // patches and evaluations.  Every load outstanding when a pair's rows are waited for was issued a whole
            // evaluation phase earlier.
            CqRows rows_in_flight;
            // (pair data is read off the lanes where it is needed, not carried around the loop in SGPRs)
            auto list_of = [&](int x) -> u32 {
                const CqPair P = cq_pair_at(pw0, pk, poff, pa, pb, x);
                return P.k + 2 <= WAVE ? load_list(C.nb, P.off, P.k, P.big, P.small, lane) : 0u;
            };
            u32 Lvc = list_of(0), Lvn = cnt > 1 ? list_of(1) : 0u;
            { u32 w[24]; cq_issue_rows(rr, Lvc, sel, dw, w); cq_rows_put(rows_in_flight, w); }
#pragma nounroll
            for (int x = 0; x < cnt; ++x) {
                const int kc = (int)rdlane(pk, x);
                const bool fast = kc + 2 <= WAVE;
                u64 myH = 0ull;
                { u32 w[24]; cq_rows_get(rows_in_flight, w); if (fast) myH = cq_consume_rows(rr, w, Lvc, sel, dw, kc + 2, lane); }
                if (x + 1 < cnt) { u32 w[24]; cq_issue_rows(rr, Lvn, sel, dw, w); cq_rows_put(rows_in_flight, w); }   // rows of pair x+1 (its list came in during the last evaluations)
                const u32 Lvnn = x + 2 < cnt ? list_of(x + 2) : 0u;                                                      // list of pair x+2
                const CqPair Pc = cq_pair_at(pw0, pk, poff, pa, pb, x);
                tally_k(Pc);
                bool done = false;
                if (fast) {
                    myH = cq_patch(myH, Lvc, kc + 2, dvv, oldt, newt, n_d, Pc.xi, Pc.xj, lane);
                    done = cq_eval_pair<MAXT>(myH, Hp, Pc, tmax, lane, delta, es, guard, R.status);
                }
                if (!done) slow(Pc, x);
                Lvc = Lvn; Lvn = Lvnn;
            }
        } else {
            for (int x = 0; x < cnt; ++x) {
                const CqPair P = cq_pair_at(pw0, pk, poff, pa, pb, x);
                tally_k(P);
                bool done = false;
                if (P.k + 2 <= WAVE) {
                    const u32 Lv = load_list(C.nb, P.off, P.k, P.big, P.small, lane);
                    u64 myH = build_local_loop16(rr, stride32, Lv, P.k + 2, lane);
                    myH = cq_patch(myH, Lv, P.k + 2, dvv, oldt, newt, n_d, P.xi, P.xj, lane);
                    done = cq_eval_pair<MAXT>(myH, Hp, P, tmax, lane, delta, es, guard, R.status);
                }
                if (!done) slow(P, x);
            }
        }
    }
    return R;
}

// ---------------------------------------------------------------------------------------------------------------------
template <int MAXT, bool ROWS128, bool XW>
__device__ __forceinline__ void cq_wave(const FcmStepParams &p, u64 *smem)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const u32 chain = blockIdx.x;
    const u32 N = (u32)p.nprop;
    if (N == 0) return;
    const int maxnw = p.maxnw < 2 ? 2 : p.maxnw;

    // the multi-wave kernel's LDS layout with W = 1 (its table fill and exact run are called as they are), then the
    // clique move's arrays and this kernel's tallies
    u64 *ent = smem;                                       // E[i] = {count i, flag, bmin i, bmax i}
    u32 *ctl = (u32 *)(smem + MW_HEAD_OFF);
    u32 *ctx = (u32 *)(smem + MW_CTX_OFF);
    u32 *vis = (u32 *)(smem + MW_VIS_OFF);
    u32 *stage = (u32 *)(smem + MW_SHARED_WORDS);          // record 0 of the ring: where a simple move's exact run leaves its record
    const CqLds L = cq_carve(smem, maxnw);
    u64 *mine_lds = L.mine;
    u64 *Hp = mine_lds;
    const u32 *T = (const u32 *)(mine_lds + 128);
    u64 *wide_lds = L.wide;
    const CliqueLds CL = L.CL;
    u32 *tly = L.tly;
    u64 *st_g = (u64 *)p.stats + (size_t)chain * FCM_DEV_NSTATS;
    u64 *cnt_g = (u64 *)p.counts + (size_t)chain * FCM_DEV_MAX_COUNTS;

    MwChain C;
    C.rows = p.rows + (size_t)chain * p.rows_per_chain;
    C.dbl = p.dbl + (size_t)chain * p.dbl_stride;
    C.nb = p.nb; C.etab = p.etab;
    C.rows_bytes = p.rows_per_chain * 4ull;
    C.xw = p.xw_ws ? (u64 *)p.xw_ws + (size_t)chain * FCM_XW_WORDS : nullptr;
    C.U = p.U; C.D = p.D; C.stride32 = p.stride32;
    u32 *slot_of = p.slot_of + (size_t)chain * p.U;
    const u64 sampled0 = st_g[0];
    {
        const int NC = p.ncounts;
        const bool cl = lane < NC;
        const u64 c0 = cl ? cnt_g[lane] : 0ull;
        const u64 mn = cl ? p.bmin[lane] : 0ull, mx = cl ? p.bmax[lane] : ~0ull;   // zero-padded (src/util.rs:53-57)
        const bool inb = ballot(cl && (c0 < mn || c0 > mx)) == 0ull;
        if (lane < 9) { ent[lane * 4 + 0] = c0; ent[lane * 4 + 1] = (lane == 0 && inb) ? 1ull : 0ull; ent[lane * 4 + 2] = mn; ent[lane * 4 + 3] = mx; }
        if (lane < 16) vis[lane] = MW_NONE;
        if (lane < 16) tly[lane] = 0u;
        if (lane == 0) {
            ctl[0] = 0u;
            *(smem + MW_TALLY_OFF) = 0ull;
            *(u64 *)(ctx + MC_ROWS) = (u64)C.rows; *(u64 *)(ctx + MC_DBL) = (u64)C.dbl; *(u64 *)(ctx + MC_NB) = (u64)C.nb;
            *(u64 *)(ctx + MC_ETAB) = (u64)C.etab; *(u64 *)(ctx + MC_ROWS_BYTES) = C.rows_bytes; *(u64 *)(ctx + MC_SEED) = p.seed;
            *(u64 *)(ctx + MC_SAMPLED0) = sampled0;
            *(u64 *)(ctx + MC_CUM0) = p.cum0; *(u64 *)(ctx + MC_CUM1) = p.cum1; *(u64 *)(ctx + MC_CUM2) = p.cum2;
            ctx[MC_U] = C.U; ctx[MC_D] = C.D; ctx[MC_STRIDE32] = C.stride32; ctx[MC_GCHAIN] = p.first_chain + chain;
            ctx[MC_MAXNW] = (u32)maxnw; ctx[MC_W] = 1u; *(u64 *)(ctx + MC_GUARD) = p.guard_limit; *(u64 *)(ctx + MC_XW) = (u64)C.xw;
            L.tables[4 * CQ_ORDERS] = (u64)p.clq; L.tables[4 * CQ_ORDERS + 1] = (u64)p.clq_pairs; L.tables[4 * CQ_ORDERS + 2] = (u64)(u32)p.cl_orders | ((u64)p.chg_cap << 32);
        }
        if (lane < CQ_ORDERS) {
            L.tables[lane] = p.cl_base[lane]; L.tables[CQ_ORDERS + lane] = p.cl_count[lane];
            L.tables[2 * CQ_ORDERS + lane] = p.clp_base[lane]; L.tables[3 * CQ_ORDERS + lane] = p.cumo[lane];
        }
        wave_sync();
    }
    FcmGuard guard = {MAXT >= 6 ? p.guard_limit : 0x7FFFFFFFull, 0u};
    u32 ti = MW_TBL_N;

    for (u32 q = 0; q < N; ++q) {
        if (ti >= MW_TBL_N) { mw_fill_table(smem, 0u, q); ti = 0u; }
        const u32 tv = lane < (int)MW_TBL_WORDS ? T[ti * MW_TBL_WORDS + lane] : 0u;
        ++ti;
        const int move = (int)(rdlane(tv, 0) & 0xFFu);

        // what the decision needs, from either kind of move
        int myd = 0;                       // lane d: the change of count[d]
        long long wide_d = 0;              // ... its share that came through a 64-bit evaluator
        u32 nonempty = 0u, pst = 0u;
        u32 t_sumk = 0u, t_changes = 0u, t_wide = 0u, t_big = 0u, kind = 1u << OT_EMPTY;
        int npairs = 0;
        u32 sv = 0u, w_clr = 0u, w_set = 0u;

        if (move < 2) {
            // ---- single_edge_flip / double_edge_move (src/lib.rs:292-325): the multi-wave kernel's exact run, out of line
            mw_exact_call<MAXT, ROWS128>(smem, 0u, tv, q, (u32)(stage - (u32 *)smem));
            const u32 *out = (const u32 *)(mine_lds + 64);
            const u32 xv = lane < 18 ? out[lane] : 0u;
            myd = lane < 16 ? (int)xv : 0;
            w_clr = rdlane(xv, 16); w_set = rdlane(xv, 17);
            sv = lane < SR_WORDS ? stage[lane] : 0u;
            wave_sync();
            const u32 flg = rdlane(sv, SR_FLAGS);
            nonempty = flg & SRF_NONEMPTY;
            pst = rdlane(sv, SR_SUS);
            t_sumk = (flg >> 8) & 0xFFFu; t_wide = (flg >> 2) & 1u; t_big = (flg >> 3) & 1u;
            if (nonempty) kind = (flg & SRF_DMOVE) ? (1u << OT_DMOVE) : (1u << OT_FLIP);
        } else {
            // ---- clique_permute / clique_swap (src/lib.rs:214-290)
            cq_setup_call(smem, tv, q);
            const u32 csv = lane < 16 ? tly[lane] : 0u;
            const int nchg = (int)rdlane(csv, CS_NCHG);
            pst = rdlane(csv, CS_STATUS);
            if (nchg > 0) {
                nonempty = 1u;
                npairs = (int)rdlane(csv, CS_NPAIRS);
                kind = move == 2 ? (1u << OT_CPERM) : (1u << OT_CSWAP);
                int delta[MAXT + 1];
#pragma unroll
                for (int t = 0; t <= MAXT; ++t) delta[t] = 0;
                EvScal es = {0, 0};
                const CqEval ev = cq_pairs<MAXT, ROWS128, XW>(C, Hp, wide_lds, maxnw, CL, npairs, (int)rdlane(csv, CS_ND), lane, delta, es, &guard);
                pst |= ev.status;
                t_sumk = (u32)ev.sum_k; t_changes = (u32)nchg; t_wide = ev.n_wide;
                fcm_lane_guard<MAXT>(delta, guard);
                myd = lane_in(4ull) ? es.d1 : 0;
                if (MAXT >= 2) myd = lane_in(8ull) ? es.d2 : myd;
#pragma unroll
                for (int tq = 3; tq <= MAXT; ++tq) {
                    const int sum = wave_sum_i32(delta[tq]);
                    myd = lane_in(1ull << (tq + 1)) ? sum : myd;
                }
                if (ev.any_wide) {
                    const Wide W = wide_carve(wide_lds, maxnw);
                    if (lane >= 2 && lane < 16 && lane - 1 <= MAXT) wide_d = W.cnt[lane - 1];
                    if (XW && ev.any_xw && lane >= 2 && lane < 16 && lane - 1 <= MAXT) wide_d += xw_count(C.xw, lane - 1);
                    wave_sync();
                }
            }
        }

        // ---- sampled += 1; Bounds::check; accept or drop (src/lib.rs:185-191)
        const u32 eoff = (u32)min(lane, 8) * 32u;
        const uint4 dyn = *(const uint4 *)((const char *)ent + eoff);
        const uint4 stat = *(const uint4 *)((const char *)ent + eoff + 16u);
        const u64 bmin = (u64)stat.x | ((u64)stat.y << 32), bmax = (u64)stat.z | ((u64)stat.w << 32);
        const u64 cnt = (u64)dyn.x | ((u64)dyn.y << 32);
        const u32 in_bounds = rdlane(dyn.z, 0);
        const u64 ncnt = cnt + (u64)((long long)myd + wide_d);
        const u64 outside = ballot(ncnt < bmin) | ballot(ncnt > bmax);
        const u32 commit = outside == 0ull ? nonempty : 0u;
        if (nonempty && (ballot((long long)ncnt < 0) & 0xFFull)) pst |= 8u;   // reference assert, src/lib.rs:65 (counts stay far below 2^63)
        // flag_count grows to post's length when the transition is applied, accepted or not, and never shrinks (src/lib.rs:72-74, 89-91)
        const u32 nzm = nonempty ? (u32)(ballot(ncnt != 0ull) & 0xFFull) : 0u;
        if (commit) {
            if (lane < 8) ent[lane * 4] = ncnt;
            if (lane == 0 && !in_bounds) ent[1] = 1ull;
            if (move >= 2) {
                cq_commit(C.rows, C.stride32, CL, npairs, lane);
                pst |= clique_update_slots(C.dbl, slot_of, CL, npairs, lane);
            } else {
                const u32 flg = rdlane(sv, SR_FLAGS);
                const u32 wid_clr = rdlane(sv, SR_WCLR), wid_set = rdlane(sv, SR_WSET);
                const u32 bit_clr = 1u << ((flg >> 20) & 31u), bit_set = 1u << ((flg >> 25) & 31u);
                u32 nclr = w_clr & ~bit_clr, nset = w_set | bit_set;
                if (wid_clr == wid_set) { nclr |= bit_set; nset = nclr; }
                if (lane == 0) {
                    C.rows[wid_clr] = nclr;
                    C.rows[wid_set] = nset;
                    if (flg & SRF_DMOVE) {   // the slot's pair stops being reciprocal, the single edge's pair becomes so
                        const u32 slot = rdlane(sv, SR_DSLOT), was = rdlane(sv, SR_ID1), now = rdlane(sv, SR_ID2);
                        slot_of[was] = FCM_NOSLOT;
                        slot_of[now] = slot;
                        C.dbl[slot] = now;
                    }
                }
            }
            wave_sync();
        }
        {   // tallies (OT_*): lanes 0..9 add, lanes 10..11 OR
            const u32 acc_inc = commit | ((nonempty ^ 1u) & in_bounds);        // an empty transition is accepted iff the state is inside the bounds (:186-187)
            const u64 im = (u64)(kind | (acc_inc << OT_ACCEPTED) | (t_big << OT_BIG));
            u32 inc = lane_in(im) ? 1u : 0u;
            inc = lane_in(1ull << OT_WIDE) ? t_wide : inc;
            inc = lane_in(1ull << OT_SUMK) ? t_sumk : inc;
            inc = lane_in(1ull << OT_CHANGES) ? t_changes : inc;
            const u32 orv = lane_in(1ull << OT_NZ) ? nzm : pst;
            const u32 taddr = mw_lds_addr(tly) + (u32)lane * 4u;
            asm volatile("s_mov_b64 exec, 0x3ff\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, 0xc00\n\tds_or_b32 %0, %2\n\ts_mov_b64 exec, -1"
                         :: "v"(taddr), "v"(inc), "v"(orv) : "memory");
        }
    }

    wave_sync();
    if (lane < p.ncounts) cnt_g[lane] = ent[lane * 4];
    const u32 tl = lane < 16 ? tly[lane] : 0u;
    const u32 nzall = rdlane(tl, OT_NZ);
    const u32 nlen = nzall ? (u32)(32 - __clz((int)nzall)) : 0u;
    const u32 stw = rdlane(tl, OT_STATUS) | (guard.tripped ? 256u : 0u);
    if (lane == 0) {
        st_g[0] = sampled0 + N; st_g[1] += rdlane(tl, OT_ACCEPTED); st_g[2] += rdlane(tl, OT_EMPTY); st_g[3] += rdlane(tl, OT_FLIP);
        st_g[4] += rdlane(tl, OT_DMOVE); st_g[5] += rdlane(tl, OT_SUMK); if (nlen > st_g[6]) st_g[6] = nlen; st_g[7] |= stw;
        st_g[8] += rdlane(tl, OT_CPERM); st_g[9] += rdlane(tl, OT_CSWAP); st_g[10] += rdlane(tl, OT_CHANGES);
        st_g[12] += rdlane(tl, OT_WIDE); st_g[13] += rdlane(tl, OT_BIG);
    }
}

template <int MAXT, bool ROWS128, bool XW>
__global__ __launch_bounds__(WAVE, 4) void fcm_step_cq_kernel(const FcmStepParams p)
{
    extern __shared__ u64 smem[];
    if (blockIdx.x >= p.nchains) return;
    cq_wave<MAXT, ROWS128, XW>(p, smem);
}
