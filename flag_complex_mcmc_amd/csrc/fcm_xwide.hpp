// fcm_xwide.hpp — evaluator for local sets of 257 .. 1024 vertices (simple moves).
//
// The reference recounts whatever neighbourhood an edge has (src/lib.rs:62-71).  Here the fast evaluator takes local
// sets of <= 64 vertices and the wide one (masks of <= 4 words in LDS) <= 256; a pair whose common neighbourhood is
// larger -- a hub pair of a connectome -- comes here.  It is the rare path and written for footprint, not speed:
// the masks live in a per-chain workspace in global memory (allocated only when the graph has such a pair), and a
// mask's NW <= 16 words are spread over the lanes (lane g holds word g), so the walk costs two 64-bit registers per
// lane whatever NW is.  Same scheme as the wide evaluator otherwise: one wave-uniform DFS over the classified local
// set, phase order P* M* S*, popcounts at the leaves.
#pragma once

#define FCM_XW_MAXNW 16                      // 1024 local vertices
#define FCM_XW_LEVELS 16
// workspace per chain, u64 words: H[1024][16] | stk[16 levels][2][16] + phases | cnt[16] | L[1024 u32]
#define FCM_XW_H_WORDS (1024u * 16u)
#define FCM_XW_STK_WORDS (FCM_XW_LEVELS * 2u * 16u + FCM_XW_LEVELS)
#define FCM_XW_WORDS (FCM_XW_H_WORDS + FCM_XW_STK_WORDS + 16u + 512u)

struct XWide {
    u64 *H;          // [s][NW] out-masks of the local vertices
    u64 *stk;        // [level][2][16] saved (cand, rem) words, then [level] saved phases
    long long *cnt;  // [16] signed simplex counts by number of K-vertices
    u32 *L;          // [s] local vertex ids
    int NW;
};
__device__ __forceinline__ XWide xw_carve(u64 *ws, int s)
{
    XWide X;
    X.H = ws;
    X.stk = ws + FCM_XW_H_WORDS;
    X.cnt = (long long *)(X.stk + FCM_XW_STK_WORDS);
    X.L = (u32 *)(X.cnt + 16);
    X.NW = (s + 63) >> 6;
    return X;
}
// (one wave writes and reads the workspace: program order is memory order at wavefront scope)
__device__ __forceinline__ void xw_sync() { wave_sync(); }

__device__ __forceinline__ void xw_load_list(const XWide &X, const u32 *nb, u32 off, int k, u32 big, u32 small, int lane)
{
    for (int j = lane; j < k + 2; j += WAVE) X.L[j] = j < k ? nb[off + j] : (j == k ? big : small);
    if (lane < 16) X.cnt[lane] = 0;
    xw_sync();
}

// induced out-adjacency: for every block of 64 columns, lane j tests its column's bit in row i; the ballot is the word
__device__ __forceinline__ void xw_build(const XWide &X, const u32 *rows, u32 stride32, int s, int lane)
{
    const int NW = X.NW;
    for (int g = 0; g < NW; ++g) {
        const int j = g * 64 + lane;
        const bool act = j < s;
        const u32 lv = act ? X.L[j] : 0u;
        const u32 woff = lv >> 5, bit = lv & 31u;
        for (int i0 = 0; i0 < s; i0 += 4) {
            u32 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32 vi = X.L[min(i0 + q, s - 1)];
                w[q] = act ? rows[(size_t)vi * stride32 + woff] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u64 m = ballot(act && ((w[q] >> bit) & 1u));
                if (i0 + q < s && lane == 0) X.H[(size_t)(i0 + q) * NW + g] = m;
            }
        }
    }
    xw_sync();
}

__device__ __forceinline__ bool xw_has(const XWide &X, int i, int j) { return (X.H[(size_t)i * X.NW + (j >> 6)] >> (j & 63)) & 1ull; }
__device__ __forceinline__ void xw_set(const XWide &X, int i, int j, bool present, int lane)
{
    if (lane == 0) {
        u64 &w = X.H[(size_t)i * X.NW + (j >> 6)];
        const u64 b = 1ull << (j & 63);
        w = present ? (w | b) : (w & ~b);
    }
    xw_sync();
}

struct XCls { u64 P, M, S; };   // per lane: word `lane` of the three class masks (lanes >= NW: 0)

// classes of every local vertex relative to the edge iu->iv
__device__ __forceinline__ XCls xw_classify(const XWide &X, int iu, int iv, int s, int lane)
{
    const int NW = X.NW;
    u64 inU = 0ull, inV = 0ull;
    for (int g = 0; g < NW; ++g) {
        const int j = g * 64 + lane;
        const bool a = j < s;
        const u64 mu = ballot(a && xw_has(X, a ? j : 0, iu)), mv = ballot(a && xw_has(X, a ? j : 0, iv));
        if (lane == g) { inU = mu; inV = mv; }
    }
    const bool mine = lane < NW;
    const u64 outU = mine ? X.H[(size_t)iu * NW + lane] : 0ull, outV = mine ? X.H[(size_t)iv * NW + lane] : 0ull;
    u64 nbm = ~0ull;
    if ((iu >> 6) == lane) nbm &= ~(1ull << (iu & 63));
    if ((iv >> 6) == lane) nbm &= ~(1ull << (iv & 63));
    XCls c;
    c.P = inU & inV & nbm;
    c.M = outU & inV & nbm;
    c.S = outU & outV & nbm;
    return c;
}

// adds sign * (#simplices with t K-vertices) to X.cnt[t]
__device__ __forceinline__ void xw_dfs(const XWide &X, const XCls &c, int tmax, int sign, int lane, u32 *overflow = nullptr)
{
    const int NW = X.NW;
    if (tmax < 1) return;
    u64 cand = c.P | c.M | c.S, rem = 0ull;
    long long acc = 0;   // lane t holds the count of simplices with t K-vertices
    u32 *stkph = (u32 *)(X.stk + FCM_XW_LEVELS * 2u * 16u);
    int ph = 0, ph2 = -1, level = 0;
    for (;;) {
        if (ballot(rem != 0ull) == 0ull) {
            ph2 = max(ph2 + 1, ph);
            if (ph2 > 2) {  // node exhausted: back to the parent
                if (level == 0) break;
                --level;
                if (lane < 16) { cand = X.stk[(level * 2 + 0) * 16 + lane]; rem = X.stk[(level * 2 + 1) * 16 + lane]; }
                const u32 pp = stkph[level];
                ph = (int)(pp & 0xFFu); ph2 = (int)(pp >> 8) - 1;
                ph = (int)__builtin_amdgcn_readfirstlane(ph); ph2 = (int)__builtin_amdgcn_readfirstlane(ph2);
                continue;
            }
            rem = cand & (ph2 == 0 ? c.P : (ph2 == 1 ? c.M : c.S));
            const int pc = wave_sum_i32(__popcll(rem));
            if (lane == level + 1) acc += (long long)sign * pc;
            // (with `overflow` given -- the counter's second pass -- the children of the last tracked level are still looked at,
            //  as wide_dfs<DETECT> does: a non-empty child set there means simplices beyond the tracked dimensions exist)
            if (!(overflow != nullptr || level + 2 <= tmax)) rem = 0ull;
            continue;
        }
        // next child x of class ph2: the lowest set bit of the lowest non-empty word
        const u64 m = ballot(rem != 0ull);
        const int g0 = __ffsll((long long)m) - 1;
        const u64 r = rdlane64(rem, g0);
        const int x = g0 * 64 + __ffsll((long long)r) - 1;
        if (lane == g0) rem &= rem - 1ull;
        u64 ge = c.S;
        if (ph2 <= 1) ge |= c.M;
        if (ph2 == 0) ge |= c.P;
        const u64 nc = lane < NW ? (cand & X.H[(size_t)x * NW + lane] & ge) : 0ull;
        if (ballot(nc != 0ull)) {
            if (level + 2 <= tmax && level + 1 < FCM_XW_LEVELS) {
                if (lane < 16) { X.stk[(level * 2 + 0) * 16 + lane] = cand; X.stk[(level * 2 + 1) * 16 + lane] = rem; }
                if (lane == 0) stkph[level] = (u32)ph | ((u32)(ph2 + 1) << 8);
                xw_sync();
                cand = nc; rem = 0ull;
                ++level;
                ph = ph2;
                ph2 = ph - 1;
            } else if (overflow) {
                *overflow = 1u;  // simplices deeper than the tracked dimensions exist
            }
        }
    }
    if (lane < 16) X.cnt[lane] += acc;
    xw_sync();
}

// The twins of wide_flip / wide_del / wide_add.  Results accumulate in X.cnt.
__device__ __forceinline__ int xw_flip(u64 *ws, const u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big, u32 small, int lane, int tmax)
{
    const int s = k + 2;
    const XWide X = xw_carve(ws, s);
    xw_load_list(X, nb, off, k, big, small, lane);
    xw_build(X, rows, stride32, s, lane);
    const bool ab = xw_has(X, k, k + 1), ba = xw_has(X, k + 1, k);
    if (ab == ba) return ab ? 0 : -1;
    const int iu = ab ? k : k + 1, iv = ab ? k + 1 : k;
    xw_dfs(X, xw_classify(X, iu, iv, s, lane), tmax, -1, lane);
    xw_set(X, iu, iv, false, lane);
    xw_set(X, iv, iu, true, lane);
    xw_dfs(X, xw_classify(X, iv, iu, s, lane), tmax, +1, lane);
    return ab ? 1 : 2;
}
// keep: the counts of an earlier half of the same proposal stay (a double move's removal before its addition)
__device__ __forceinline__ bool xw_del(u64 *ws, const u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big, u32 small, u32 coin, int lane, int tmax)
{
    const int s = k + 2;
    const XWide X = xw_carve(ws, s);
    xw_load_list(X, nb, off, k, big, small, lane);
    xw_build(X, rows, stride32, s, lane);
    const bool ab = xw_has(X, k, k + 1), ba = xw_has(X, k + 1, k);
    const int iu = coin ? k : k + 1, iv = coin ? k + 1 : k;
    xw_dfs(X, xw_classify(X, iu, iv, s, lane), tmax, -1, lane);
    return ab && ba;
}
__device__ __forceinline__ void xw_add(u64 *ws, const u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big, u32 small, u32 fwd, u32 dfrom,
                                        u32 dto, int lane, int tmax, bool keep_counts)
{
    const int s = k + 2;
    const XWide X = xw_carve(ws, s);
    long long keep = 0;
    if (keep_counts && lane < 16) keep = X.cnt[lane];
    xw_load_list(X, nb, off, k, big, small, lane);
    if (keep_counts && lane < 16) X.cnt[lane] = keep;
    xw_sync();
    xw_build(X, rows, stride32, s, lane);
    int fi = -1, ti = -1;
    for (int g = 0; g < X.NW; ++g) {
        const int j = g * 64 + lane;
        const u32 lv = j < s ? X.L[j] : 0xFFFFFFFFu;
        const u64 mf = ballot(j < s && lv == dfrom), mt = ballot(j < s && lv == dto);
        if (mf) fi = g * 64 + __ffsll((long long)mf) - 1;
        if (mt) ti = g * 64 + __ffsll((long long)mt) - 1;
    }
    if (fi >= 0 && ti >= 0) xw_set(X, fi, ti, false, lane);
    const int ia = fwd ? k : k + 1, ib = fwd ? k + 1 : k;
    xw_set(X, ib, ia, true, lane);
    xw_dfs(X, xw_classify(X, ib, ia, s, lane), tmax, +1, lane);
}
// one evaluation of an edge present in the bitmap (a changed pair of a clique move): sign * (#simplices through it) is added
// to the counts; keep_counts: those of the move's earlier pairs stay.  fwd = 1: big->small.
__device__ __forceinline__ bool xw_edge(u64 *ws, const u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big, u32 small, u32 fwd,
                                        int sign, int lane, int tmax, bool keep_counts)
{
    const int s = k + 2;
    const XWide X = xw_carve(ws, s);
    long long keep = 0;
    if (keep_counts && lane < 16) keep = X.cnt[lane];
    xw_load_list(X, nb, off, k, big, small, lane);
    if (keep_counts && lane < 16) X.cnt[lane] = keep;
    xw_sync();
    xw_build(X, rows, stride32, s, lane);
    const int iu = fwd ? k : k + 1, iv = fwd ? k + 1 : k;
    const bool present = xw_has(X, iu, iv);
    xw_dfs(X, xw_classify(X, iu, iv, s, lane), tmax, sign, lane);
    return present;
}
__device__ __forceinline__ long long xw_count(u64 *ws, int idx) { return xw_carve(ws, 64).cnt[idx]; }
