// fcm_step_cq.hpp — the cooperative step kernel for move mixes WITH clique moves (reference default [0.1, 0.1, 0.6, 0.2],
// src/bin/sample.rs:17,101; clique_permute / clique_swap, src/lib.rs:214-290).  Included by fcm_step_variant.hip for the
// c<depth> tags; 2 <= depth <= 6 (up to 8 count entries), like the multi-wave kernel of the simple moves.
//
// W waves per chain (W = 1, 2, 4, 8: a launch parameter), one proposal at a time.  What differs from the one-wave
// kernel's clique path (fcm_clique.hpp, clique_propose):
//
//  * Every changed vertex pair of a move is evaluated on the PRE-MOVE bitmap.  The reference applies a move's change
//    edges one by one and recounts in between (State::apply_transition, src/lib.rs:61-79); the telescoping sum needs
//    pair x counted on the graph "pairs before x already changed".  Every earlier pair lies in the same clique(s), so
//    wherever it matters both its endpoints are in pair x's local set: the earlier changes are XORed into the local
//    masks in registers (cq_patch) instead of being stored to memory and read back.  Nothing is written until the move
//    is accepted -- one pass of stores, all pairs at once -- and a rejected move needs no revert (the reference's
//    revert_transition, :81-95, has nothing to undo).
//  * That makes the pairs of a move independent of each other: wave w of the chain's workgroup takes the pairs
//    w, w+W, ...; the waves' count changes meet in LDS, wave 0 decides and commits.  With few chains per GPU (the
//    per-GPU shares of the 8-GPU configs: 1024, 256 chains) this is what keeps the chip busy: a move's ~6 pair
//    evaluations run side by side instead of one after the other.  (With 4096 chains one wave per chain already fills
//    the 4 waves per SIMD that 128 VGPRs allow, and W = 1.)
//  * The chain's context lives in LDS; the proposal itself (clique_setup) and the simple moves (20 % of the default
//    mix; the multi-wave kernel's exact run, mw_exact_call) are out-of-line calls made by wave 0.
//
// Local sets beyond 64 vertices, split graphs that do not fit 64 nodes and local sets of 257..1024 vertices are left to
// wave 0, which takes them one by one on the wide / workspace evaluators with the same patches set into their masks.
#pragma once

// LDS map in u64 words (W = waves per chain):
//   the multi-wave kernel's layout for ONE wave (its table fill and exact run are called as they are, by wave 0):
//     shared words | ring (record 0 = where a simple move's exact run leaves its record) | wave 0: Hp, arc list, draw table, tallies | wide evaluator
//   tables (clique buckets, order thresholds) | tallies + the setup's scalars | dix | deferred pairs | per-wave sums [W][16] | clique arrays
//   | Hp + arc list of the waves 1 .. W-1
#define CQ_TALLY_WORDS 9u   // OT_* (14 u32), then the last setup's nchg, npairs, status, n_d
#define CQ_ORDERS 8         // clique orders the kernel takes (<= 8 count entries means cliques of <= 8 vertices)
#define CQ_TABLE_WORDS (4u * CQ_ORDERS + 4u)   // cl_base | cl_count | clp_base | cumo | clq, clq_pairs, (cl_orders, chg_cap), spare
#define CQ_DIX_WORDS 128u   // one byte per vertex of a graph of <= 1024 vertices: its index in d, plus one
#define CQ_DEFER_WORDS 34u  // [0]: number of deferred pairs (u32), then up to 128 pair indices (u16)
#define CQ_PSUM_WORDS 16u   // per wave: 16 x i32 count changes, then status, sum_k, spare x 2 ... (32 u32)
#define CQ_MAXW 8u
// (Hp + arc list of the waves 1, 2, 3 lie IN the wide evaluator's region: that evaluator runs on wave 0 only -- a simple move's
//  exact run, the deferred pairs of a clique move -- while the other waves wait at a barrier; with W = 2 the workgroup then
//  stays within the 10 KiB that let 16 of them share a CU.  Waves 4 .. 7 have theirs behind the clique arrays.)
__host__ __device__ constexpr inline unsigned fcm_cq_alias_waves(int NW) { return fcm_lds_words(NW) / 128u < 3u ? fcm_lds_words(NW) / 128u : 3u; }
__host__ __device__ constexpr inline unsigned fcm_cq_lds_words(int NW, unsigned chg_cap, unsigned W)
{
    return fcm_mw_lds_words(NW, 1) + CQ_TABLE_WORDS + CQ_TALLY_WORDS + CQ_DIX_WORDS + CQ_DEFER_WORDS + W * CQ_PSUM_WORDS + fcm_clique_lds_words(chg_cap)
           + (W - 1u > fcm_cq_alias_waves(NW) ? (W - 1u - fcm_cq_alias_waves(NW)) * 128u : 0u);
}
struct CqLds {
    u64 *mine, *wide, *tables;
    u32 *tly, *defer, *psum;
    unsigned char *dix;
    CliqueLds CL;
    u64 *hp_more;     // Hp + arc list of wave w >= 1: cq_hp(L, w)
    unsigned alias;   // waves whose Hp lies in the wide evaluator's region
};
__device__ __forceinline__ u64 *cq_hp(const CqLds &L, u32 w)
{
    return w == 0u ? L.mine : (w <= L.alias ? L.wide + (size_t)(w - 1u) * 128u : L.hp_more + (size_t)(w - 1u - L.alias) * 128u);
}
// (everything whose place does not depend on the pair list's capacity comes first; W from the launch)
__device__ __forceinline__ CqLds cq_carve(u64 *smem, int maxnw, u32 W, u32 chg_cap)
{
    CqLds L;
    L.mine = smem + MW_SHARED_WORDS + MW_RING_WORDS(1u);
    L.wide = L.mine + MW_WAVE_WORDS;
    L.tables = L.wide + fcm_lds_words(maxnw);
    L.tly = (u32 *)(L.tables + CQ_TABLE_WORDS);
    L.dix = (unsigned char *)(L.tables + CQ_TABLE_WORDS + CQ_TALLY_WORDS);
    L.defer = (u32 *)(L.tables + CQ_TABLE_WORDS + CQ_TALLY_WORDS + CQ_DIX_WORDS);
    L.psum = (u32 *)(L.tables + CQ_TABLE_WORDS + CQ_TALLY_WORDS + CQ_DIX_WORDS + CQ_DEFER_WORDS);
    u64 *cl = L.tables + CQ_TABLE_WORDS + CQ_TALLY_WORDS + CQ_DIX_WORDS + CQ_DEFER_WORDS + W * CQ_PSUM_WORDS;
    L.CL = clique_carve(cl);
    L.hp_more = cl + fcm_clique_lds_words(chg_cap);
    L.alias = fcm_cq_alias_waves(maxnw);
    return L;
}
enum { CS_NCHG = 14, CS_NPAIRS, CS_STATUS, CS_ND };   // u32 words of the tally block the setup leaves its scalars in
enum { PS_STATUS = 16, PS_SUMK = 17 };               // u32 words of a wave's sums behind its 16 count changes

// ---- out of line (wave 0): the proposal of a clique move (clique_setup) with its context from LDS; what it finds goes
// to the clique arrays (pair list in CL.chg, d in CL.d, the in-masks over d before / after the move in CL.oldm / CL.newm)
// and to the CS_* words.
__device__ __attribute__((noinline)) void cq_setup_call(u64 *smem, u32 tv, u32 q, u32 W)
{
    const int lane = threadIdx.x & (WAVE - 1);
    q = mw_uni(q); W = mw_uni(W);
    const u32 *ctx = (const u32 *)(smem + MW_CTX_OFF);
    const u32 cv = lane < MC_WORDS ? ctx[lane] : 0u;
    const int maxnw = (int)rdlane(cv, MC_MAXNW);
    const u64 oc = mw_uni64((smem + MW_SHARED_WORDS + MW_RING_WORDS(1u) + MW_WAVE_WORDS + fcm_lds_words(maxnw))[4 * CQ_ORDERS + 2]);
    const u32 chg_cap = (u32)(oc >> 32);
    const CqLds L = cq_carve(smem, maxnw, W, chg_cap);
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
    if (lane == 0) { L.tly[CS_NCHG] = (u32)cr.nchg; L.tly[CS_NPAIRS] = (u32)cr.npairs; L.tly[CS_STATUS] = cr.status; L.tly[CS_ND] = (u32)cr.n_d; L.defer[0] = 0u; }
    wave_sync();
}

// ---- the earlier pairs' changes, XORed into the local in-masks ------------------------------------------------------
// myH: lane j's in-mask over the local list Lv (bit i = L[i] -> L[j]) as read from the pre-move bitmap.  Pair x =
// (xi, xj) in d indices, xi < xj; the pairs are listed ascending (i, j), so "before x" = (i, j) < (xi, xj).  oldt / newt:
// lane t's in-masks over d (bit u = d[u] -> d[t]) before and after the whole move; dvv: d[lane].
// For every d-vertex u that is in the list: the lanes of the d-vertices t whose edge u -> t changed before x flip bit
// pos(u).
// dix (graphs of <= 1024 vertices): LDS table vertex -> index in d plus one, 0 for the vertices not in d; null otherwise.
__device__ __forceinline__ u64 cq_patch(u64 myH, u32 Lv, int s, u32 dvv, u32 oldt, u32 newt, int n_d, int xi, int xj, int lane, const unsigned char *dix)
{
    // chg_in[t]: bits u with (pair {t,u} before x) and (edge u -> t changed)
    const u32 lt_i = (1u << xi) - 1u, lt_j = (1u << xj) - 1u;
    const u32 M = lane < xi ? 0xFFFFFFFFu : (lane == xi ? lt_j : (lt_i | (lane < xj ? (1u << xi) : 0u)));
    const u32 chg_in = (oldt ^ newt) & M;
    if (ballot(chg_in != 0u) == 0ull) return myH;                 // nothing before x changed (always so for the first pair)
    // tj: this lane's d index, 32 or more if its vertex is not in d (lanes beyond the list repeat its last vertex: not theirs)
    u32 tj = 32u;
    if (dix) {
        tj = lane < s ? (u32)dix[Lv] - 1u : 32u;
    } else {
        for (int u = 0; u < n_d; ++u) tj = (lane < s && Lv == rdlane(dvv, u)) ? (u32)u : tj;
    }
    const bool isd = tj < 32u;
    // posv: lane u <- the list position of d[u] | 0x100 (lanes that are no d-vertex send to lane 63: n_d <= 32)
    const u32 posv = (u32)__builtin_amdgcn_ds_permute((int)((isd ? tj : 63u) * 4u), (int)((u32)lane | 0x100u));
    // this lane's changes as a target, fetched from lane tj
    // (every lane takes part: ds_bpermute reads nothing from a lane that is masked off)
    const u32 mine = (u32)__builtin_amdgcn_ds_bpermute((int)((tj & 31u) * 4u), (int)chg_in);
    u64 X = 0ull;
    for (int u = 0; u < n_d; ++u) {
        const u32 pu = rdlane(posv, u);
        if (!(pu & 0x100u)) continue;                             // d[u] is not in this pair's local set: no simplex through the pair holds it
        X |= (u64)__builtin_amdgcn_ubfe(mine, (u32)u, 1u) << (pu & 63u);
    }
    return myH ^ (isd ? X : 0ull);
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

// one pair on the fast evaluator: classes around (big, small) from the patched in-masks, one evaluation per changed
// direction.  Returns false if the split graph of a needed direction does not fit 64 nodes.
template <int MAXT>
__device__ __forceinline__ bool cq_eval_pair(u64 myH, u64 *Hp, const CqPair &P, int tmax, int lane, fcm_acc_t<MAXT> (&delta)[MAXT + 1], EvScal &es, FcmGuard *guard, u32 &status)
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
    if constexpr (MAXT <= 6) {
        // both directions change: the pair is flipped (10 <-> 01; a clique move hands a pair another pair's pattern, never none):
        // one signed evaluation for the two (eval_flip_merged) -- the direction that goes is the one that is there now
        if (need_bs && need_sb && (P.o_bs ^ P.o_sb) != 0u) {
            const u64 MA = P.o_bs ? cbs.M : csb.M, MB = P.o_bs ? csb.M : cbs.M;
            if (flip_merged_fits(cbs.P, MA, MB, cbs.S, k) && eval_flip_merged<MAXT>(myH, Hp, cbs.P, MA, MB, cbs.S, k, lane, delta, es)) return true;
        }
    }
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

// This wave's share of the changed pairs of a clique move (CL.chg, npairs of them): pairs wv, wv + W, ..., each on the
// pre-move bitmap.  Pairs the fast evaluator does not take are put on the deferred list (wave 0 takes them afterwards).
template <int MAXT>
__device__ __forceinline__ void cq_pairs(const MwChain &C, u64 *Hp, const CliqueLds CL, const unsigned char *dix, u32 *defer, int npairs, int n_d, u32 wv, u32 W,
                                         int lane, fcm_acc_t<MAXT> (&delta)[MAXT + 1], EvScal &es, FcmGuard *guard, u32 &status, u32 &sum_k)
{
    const int tmax = MAXT;
    const rsrc_t rr = make_rows_rsrc(C.rows, C.rows_bytes);
    const u32 dvv = lane < n_d ? CL.d[lane] : 0xFFFFFFFFu;
    const u32 oldt = lane < 32 ? CL.oldm[lane] : 0u, newt = lane < 32 ? CL.newm[lane] : 0u;   // in-masks over d before / after the move (cq_setup_call)
    for (int base = 0; base < npairs; base += WAVE) {
        const int cnt = min(WAVE, npairs - base);
        // lane x of the chunk: pair base + x
        u32 pw0 = 0u, pk = 0u, poff = 0u, pa = 0u, pb = 0u;
        if (lane < cnt) {
            const uint4 c = *(const uint4 *)(CL.chg + 4 * (base + lane));
            pw0 = c.x; pk = c.z; poff = c.w;
            pa = CL.d[c.x & 0xFFu]; pb = CL.d[(c.x >> 8) & 0xFFu];
        }
        // this wave's pairs of the chunk; the list of the next one is requested before the current one is evaluated
        auto list_of = [&](int x) -> u32 {
            const CqPair P = cq_pair_at(pw0, pk, poff, pa, pb, x);
            return P.k + 2 <= WAVE ? load_list(C.nb, P.off, P.k, P.big, P.small, lane) : 0u;
        };
        const int x0 = (int)((wv + W - ((u32)base % W)) % W);   // (pairs are dealt by their index in the whole list)
        u32 Lvn = x0 < cnt ? list_of(x0) : 0u;
#pragma nounroll
        for (int x = x0; x < cnt; x += (int)W) {
            const u32 Lv = Lvn;
            const CqPair P = cq_pair_at(pw0, pk, poff, pa, pb, x);
            sum_k += (u32)P.k * (u32)((P.o_bs != P.n_bs ? 1 : 0) + (P.o_sb != P.n_sb ? 1 : 0));
            bool done = false;
            if (P.k + 2 <= WAVE) {
                u64 myH = build_local(rr, C.stride32, Lv, P.k + 2, lane);
                if (x + (int)W < cnt) Lvn = list_of(x + (int)W);
                myH = cq_patch(myH, Lv, P.k + 2, dvv, oldt, newt, n_d, P.xi, P.xj, lane, dix);
                done = cq_eval_pair<MAXT>(myH, Hp, P, tmax, lane, delta, es, guard, status);
            } else if (x + (int)W < cnt) {
                Lvn = list_of(x + (int)W);
            }
            if (!done && lane == 0) {
                const u32 at = atomicAdd(&defer[0], 1u);
                ((unsigned short *)(defer + 1))[at & 127u] = (unsigned short)(base + x);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// (WFIX: the waves per chain as a compile-time constant -- 1, 2, 4, 8: the launcher picks the instantiation, as for the multi-wave
//  kernel -- or 0: from p.mw_waves at run time)
template <int MAXT, int WFIX = 0>
__device__ __forceinline__ void cq_wave(const FcmStepParams &p, u64 *smem)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const u32 wv = WFIX == 1 ? 0u : mw_uni(threadIdx.x >> 6);
    const u32 W = WFIX ? (u32)WFIX : (p.mw_waves >= 2 ? p.mw_waves : 1u);   // waves of this chain's workgroup (blockDim.x / 64)
    const u32 chain = blockIdx.x;
    const u32 N = (u32)p.nprop;
    if (N == 0) return;
    const int maxnw = p.maxnw < 2 ? 2 : p.maxnw;

    u64 *ent = smem;                                       // E[i] = {count i, flag, bmin i, bmax i}
    u32 *ctl = (u32 *)(smem + MW_HEAD_OFF);
    u32 *ctx = (u32 *)(smem + MW_CTX_OFF);
    u32 *vis = (u32 *)(smem + MW_VIS_OFF);
    u32 *stage = (u32 *)(smem + MW_SHARED_WORDS);          // record 0 of the ring: where a simple move's exact run leaves its record
    const CqLds L = cq_carve(smem, maxnw, W, p.chg_cap);
    u64 *mine_lds = L.mine;
    u64 *Hp = cq_hp(L, wv);
    const u32 *T = (const u32 *)(mine_lds + 128);
    u64 *wide_lds = L.wide;
    const CliqueLds CL = L.CL;
    u32 *tly = L.tly;
    u32 *my_psum = L.psum + (size_t)wv * 32u;
    const unsigned char *dix = p.n <= 1024u ? L.dix : nullptr;
    u64 *st_g = (u64 *)p.stats + (size_t)chain * FCM_DEV_NSTATS;
    u64 *cnt_g = (u64 *)p.counts + (size_t)chain * FCM_DEV_MAX_COUNTS;
    auto barrier = [&]() { if (W > 1u) mw_barrier(); };

    MwChain C;
    C.rows = p.rows + (size_t)chain * p.rows_per_chain;
    C.dbl = p.dbl + (size_t)chain * p.dbl_stride;
    C.nb = p.nb; C.etab = p.etab;
    C.rows_bytes = p.rows_per_chain * 4ull;
    C.xw = p.xw_ws ? (u64 *)p.xw_ws + (size_t)chain * FCM_XW_WORDS : nullptr;
    C.U = p.U; C.D = p.D; C.stride32 = p.stride32;
    u32 *slot_of = p.slot_of + (size_t)chain * p.U;
    const u64 sampled0 = st_g[0];
    if (wv == 0) {
        const int NC = p.ncounts;
        const bool cl = lane < NC;
        const u64 c0 = cl ? cnt_g[lane] : 0ull;
        const u64 mn = cl ? p.bmin[lane] : 0ull, mx = cl ? p.bmax[lane] : ~0ull;   // zero-padded (src/util.rs:53-57)
        const bool inb = ballot(cl && (c0 < mn || c0 > mx)) == 0ull;
        if (lane < 9) { ent[lane * 4 + 0] = c0; ent[lane * 4 + 1] = (lane == 0 && inb) ? 1ull : 0ull; ent[lane * 4 + 2] = mn; ent[lane * 4 + 3] = mx; }
        if (lane < 16) vis[lane] = MW_NONE;
        if (lane < 18) tly[lane] = 0u;
        for (u32 i = (u32)lane; i < CQ_DIX_WORDS; i += WAVE) ((u64 *)L.dix)[i] = 0ull;
        if (lane == 0) {
            ctl[0] = 0u;
            *(smem + MW_TALLY_OFF) = 0ull;
            *(u64 *)(ctx + MC_ROWS) = (u64)C.rows; *(u64 *)(ctx + MC_DBL) = (u64)C.dbl; *(u64 *)(ctx + MC_NB) = (u64)C.nb;
            *(u64 *)(ctx + MC_ETAB) = (u64)C.etab; *(u64 *)(ctx + MC_ROWS_BYTES) = C.rows_bytes; *(u64 *)(ctx + MC_SEED) = p.seed;
            *(u64 *)(ctx + MC_SAMPLED0) = sampled0;
            *(u64 *)(ctx + MC_CUM0) = p.cum0; *(u64 *)(ctx + MC_CUM1) = p.cum1; *(u64 *)(ctx + MC_CUM2) = p.cum2;
            ctx[MC_U] = C.U; ctx[MC_D] = C.D; ctx[MC_STRIDE32] = C.stride32; ctx[MC_GCHAIN] = p.first_chain + chain;
            ctx[MC_MAXNW] = (u32)maxnw; ctx[MC_W] = 1u; *(u64 *)(ctx + MC_GUARD) = p.guard_limit; *(u64 *)(ctx + MC_XW) = (u64)C.xw;   // (MC_W = 1: the exact run's own layout)
            ctx[MC_CLIM] = p.commit_words; ctx[MC_CLIM + 1] = 0u;
            L.tables[4 * CQ_ORDERS] = (u64)p.clq; L.tables[4 * CQ_ORDERS + 1] = (u64)p.clq_pairs;
            L.tables[4 * CQ_ORDERS + 2] = (u64)(u32)p.cl_orders | ((u64)p.chg_cap << 32);
        }
        if (lane < CQ_ORDERS) {
            L.tables[lane] = p.cl_base[lane]; L.tables[CQ_ORDERS + lane] = p.cl_count[lane];
            L.tables[2 * CQ_ORDERS + lane] = p.clp_base[lane]; L.tables[3 * CQ_ORDERS + lane] = p.cumo[lane];
        }
        wave_sync();
    }
    barrier();
    FcmGuard guard = {MAXT >= 6 ? p.guard_limit : 0x7FFFFFFFull, 0u};
    u32 ti = MW_TBL_N;

    for (u32 q = 0; q < N; ++q) {
        if (ti >= MW_TBL_N) {
            if (wv == 0) mw_fill_table(smem, 0u, q);
            barrier();
            ti = 0u;
        }
        const u32 tv = lane < (int)MW_TBL_WORDS ? T[ti * MW_TBL_WORDS + lane] : 0u;
        ++ti;
        const int move = (int)(rdlane(tv, 0) & 0xFFu);

        // what wave 0's decision needs, from either kind of move
        int myd = 0;                       // lane d: the change of count[d]
        long long wide_d = 0;              // ... its share that came through a 64-bit evaluator
        u32 nonempty = 0u, pst = 0u;
        u32 t_sumk = 0u, t_changes = 0u, t_wide = 0u, t_big = 0u, t_pairs = 0u, t_shared = 0u, kind = 1u << OT_EMPTY;
        int npairs = 0;
        u32 sv = 0u, w_clr = 0u, w_set = 0u;

        if (move < 2) {
            // ---- single_edge_flip / double_edge_move (src/lib.rs:292-325): wave 0, the multi-wave kernel's exact run, out of line
            if (wv == 0) {
                if (C.stride32 == 32u) mw_exact_call<MAXT, true>(smem, 0u, tv, q, (u32)(stage - (u32 *)smem));   // rows of one cache line: the whole-row build
                else mw_exact_call<MAXT, false>(smem, 0u, tv, q, (u32)(stage - (u32 *)smem));
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
            }
        } else {
            // ---- clique_permute / clique_swap (src/lib.rs:214-290)
            if (wv == 0) {
                cq_setup_call(smem, tv, q, W);
                if (dix) {   // vertex -> index in d, for the patches (put back to zero after the decision)
                    const u32 nd = tly[CS_ND];
                    if (tly[CS_NCHG] && lane < (int)nd) L.dix[CL.d[lane]] = (unsigned char)(lane + 1);
                    wave_sync();
                }
            }
            barrier();                                                           // the pair list is there
            const u32 csv = lane < 18 ? tly[lane] : 0u;
            const int nchg = (int)rdlane(csv, CS_NCHG), n_d = (int)rdlane(csv, CS_ND);
            pst = rdlane(csv, CS_STATUS);
            if (nchg > 0) {
                nonempty = 1u;
                npairs = (int)rdlane(csv, CS_NPAIRS);
                kind = move == 2 ? (1u << OT_CPERM) : (1u << OT_CSWAP);
                int delta[MAXT + 1];
#pragma unroll
                for (int t = 0; t <= MAXT; ++t) delta[t] = 0;
                EvScal es = {0, 0};
                u32 my_status = 0u, my_sumk = 0u;
                cq_pairs<MAXT>(C, Hp, CL, dix, L.defer, npairs, n_d, wv, W, lane, delta, es, &guard, my_status, my_sumk);
                fcm_lane_guard<MAXT>(delta, guard);
                if (guard.tripped) my_status |= 256u;                            // a local count may have passed 2^31: refuse rather than wrap
                myd = lane_in(4ull) ? es.d1 : 0;
                if (MAXT >= 2) myd = lane_in(8ull) ? es.d2 : myd;
#pragma unroll
                for (int tq = 3; tq <= MAXT; ++tq) {
                    const int sum = wave_sum_i32(delta[tq]);
                    myd = lane_in(1ull << (tq + 1)) ? sum : myd;
                }
                if (W > 1u) {                                                    // the waves' shares meet in LDS
                    if (lane < 16) my_psum[lane] = (u32)myd;
                    if (lane == 0) { my_psum[PS_STATUS] = my_status; my_psum[PS_SUMK] = my_sumk; }
                    barrier();
                    if (wv == 0) {
                        long long tot = 0;
                        u32 st = 0u, sk = 0u;
                        for (u32 w = 0; w < W; ++w) {
                            tot += lane < 16 ? (long long)(int)L.psum[w * 32u + (u32)lane] : 0ll;
                            st |= L.psum[w * 32u + PS_STATUS]; sk += L.psum[w * 32u + PS_SUMK];
                        }
                        myd = 0; wide_d = tot;                                   // (64-bit: W sums of 32-bit wave sums)
                        my_status = st; my_sumk = sk;
                    }
                }
                if (wv == 0) {
                    pst |= my_status;
                    t_sumk = my_sumk; t_changes = (u32)nchg;
                    if (npairs > 0) { t_pairs = (u32)npairs; t_shared = move == 2 ? (u32)(npairs - 1) * (u32)n_d : 0u; }   // (a permutation's pairs all lie inside the clique)
                    // the pairs left over for the wide evaluators, one by one
                    const u32 ndefer = mw_uni(L.defer[0]);
                    if (ndefer) {
                        const Wide Wd = wide_carve(wide_lds, maxnw);
                        wide_zero_counts(Wd, lane);
                        if (C.xw) { if (lane < 16) xw_carve(C.xw, 64).cnt[lane] = 0; xw_sync(); }
                        bool any_xw = false;
                        for (u32 i = 0; i < ndefer && i < 128u; ++i) {
                            const int x = (int)mw_uni(((const unsigned short *)(L.defer + 1))[i]);
                            const u32 w0 = CL.chg[4 * x], kk = CL.chg[4 * x + 2], off = CL.chg[4 * x + 3];
                            const u32 a = CL.d[w0 & 0xFFu], b = CL.d[(w0 >> 8) & 0xFFu];
                            const u32 o2 = (w0 >> 16) & 3u, n2 = (w0 >> 20) & 3u;
                            const bool agb = a > b;
                            const u32 dirs = (agb ? (o2 & 1u) : (o2 >> 1)) | ((agb ? (o2 >> 1) : (o2 & 1u)) << 1) | ((agb ? (n2 & 1u) : (n2 >> 1)) << 2) | ((agb ? (n2 >> 1) : (n2 & 1u)) << 3);
                            if ((int)kk + 2 > 64 * maxnw) any_xw = true;
                            pst |= cq_wide_pair<true>(wide_lds, maxnw, C.xw, C.rows, C.stride32, C.nb, CL, x, agb ? a : b, agb ? b : a, off, (int)kk, dirs, MAXT);
                        }
                        if (ndefer > 128u) pst |= 32u;
                        t_wide = ndefer;
                        if (lane >= 2 && lane < 16 && lane - 1 <= MAXT) wide_d += Wd.cnt[lane - 1];
                        if (any_xw && lane >= 2 && lane < 16 && lane - 1 <= MAXT) wide_d += xw_count(C.xw, lane - 1);
                        wave_sync();
                    }
                }
            }
        }

        if (wv == 0) {
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
                    // (indices out of the exact run's record: held against the chain's sizes before anything is stored, as in the multi-wave kernel)
                    const u32 slot = rdlane(sv, SR_DSLOT), was = rdlane(sv, SR_ID1), now = rdlane(sv, SR_ID2);
                    const bool bad = max(wid_clr, wid_set) >= p.commit_words || ((flg & SRF_DMOVE) && (slot >= C.D || was >= C.U || now >= C.U));
                    if (bad) pst |= 0x200u;
                    else if (lane == 0) {
                        C.rows[wid_clr] = nclr;
                        C.rows[wid_set] = nset;
                        if (flg & SRF_DMOVE) {   // the slot's pair stops being reciprocal, the single edge's pair becomes so
                            slot_of[was] = FCM_NOSLOT;
                            slot_of[now] = slot;
                            C.dbl[slot] = now;
                        }
                    }
                }
                wave_sync();
            }
            if (move >= 2 && dix && nonempty) {   // the lookup table back to zero
                const u32 nd = tly[CS_ND];
                if (lane < (int)nd) L.dix[CL.d[lane]] = 0;
                wave_sync();
            }
            {   // tallies (OT_*): lanes 0..9 and 12..13 add, lanes 10..11 OR
                const u32 acc_inc = commit | ((nonempty ^ 1u) & in_bounds);        // an empty transition is accepted iff the state is inside the bounds (:186-187)
                const u64 im = (u64)(kind | (acc_inc << OT_ACCEPTED) | (t_big << OT_BIG));
                u32 inc = lane_in(im) ? 1u : 0u;
                inc = lane_in(1ull << OT_WIDE) ? t_wide : inc;
                inc = lane_in(1ull << OT_SUMK) ? t_sumk : inc;
                inc = lane_in(1ull << OT_CHANGES) ? t_changes : inc;
                inc = lane_in(1ull << OT_PAIRS) ? t_pairs : inc;
                inc = lane_in(1ull << OT_SHARED) ? t_shared : inc;
                const u32 orv = lane_in(1ull << OT_NZ) ? nzm : pst;
                const u32 taddr = mw_lds_addr(tly) + (u32)lane * 4u;
                asm volatile("s_mov_b64 exec, 0x33ff\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, 0xc00\n\tds_or_b32 %0, %2\n\ts_mov_b64 exec, -1"
                             :: "v"(taddr), "v"(inc), "v"(orv) : "memory");
            }
            if (W > 1u && commit) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the commit's stores are in memory before the other waves go on
        }
        if (W > 1u) {                                                            // end of the proposal: the chain's state is the same for every wave
            barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }

    if (wv != 0) return;
    wave_sync();
    if (lane < p.ncounts) cnt_g[lane] = ent[lane * 4];
    const u32 tl = lane < 16 ? tly[lane] : 0u;
    const u32 nzall = rdlane(tl, OT_NZ);
    const u32 nlen = nzall ? (u32)(32 - __clz((int)nzall)) : 0u;
    u32 stw = rdlane(tl, OT_STATUS);
    if (lane == 0) {
        st_g[0] = sampled0 + N; st_g[1] += rdlane(tl, OT_ACCEPTED); st_g[2] += rdlane(tl, OT_EMPTY); st_g[3] += rdlane(tl, OT_FLIP);
        st_g[4] += rdlane(tl, OT_DMOVE); st_g[5] += rdlane(tl, OT_SUMK); if (nlen > st_g[6]) st_g[6] = nlen; st_g[7] |= stw;
        st_g[8] += rdlane(tl, OT_CPERM); st_g[9] += rdlane(tl, OT_CSWAP); st_g[10] += rdlane(tl, OT_CHANGES);
        st_g[12] += rdlane(tl, OT_WIDE); st_g[13] += rdlane(tl, OT_BIG);
        st_g[16] += rdlane(tl, OT_PAIRS); st_g[17] += rdlane(tl, OT_SHARED);   // FCM_STAT_PAIRS, FCM_STAT_SHARED_ROWS
    }
}

// (W is a launch parameter: the block is W x 64 threads.  4 waves per SIMD: at most 128 VGPRs.  A lean build in 64 VGPRs -- 8
//  waves per SIMD, W = 2 at 4096 chains -- was built and measured in round 4: 6.6e7 proposals/s against the one-wave kernel's
//  9.3e7 there; profiles/r04_cq_lean_dropped.diff)
template <int MAXT, int WFIX = 0>
__global__ __launch_bounds__(CQ_MAXW * WAVE, 4) void fcm_step_cq_kernel(const FcmStepParams p)
{
    extern __shared__ u64 smem[];
    if (blockIdx.x >= p.nchains) return;
    cq_wave<MAXT, WFIX>(p, smem);
}
