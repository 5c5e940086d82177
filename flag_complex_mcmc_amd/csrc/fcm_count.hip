// fcm_count.hip — global simplex count kernel, the row-broadcast kernel and
// the launch dispatchers.
#include "fcm_kernels_common.hpp"

// ---------------------------------------------------------------------------
// Global count: flagser_count.  One wave per directed edge u->v (grid-stride):
// the simplices whose first two vertices are u, v have their remaining
// vertices in C = out(u) & out(v); count directed simplices inside C (all of
// class S: every vertex comes after v).
// ---------------------------------------------------------------------------
#define FCM_COUNT_MAXT 14   // dims up to 15
#define FCM_COUNT_MAXNW 4   // |C| <= 256

__global__ __launch_bounds__(WAVE) void fcm_count_kernel(const FcmCountParams p)
{
    extern __shared__ u64 smem[];  // fcm_lds_words(FCM_COUNT_MAXNW)
    const Wide W = wide_carve(smem, FCM_COUNT_MAXNW);
    u64 *Hs = smem;      // fast path: 64 masks
    u32 *Lc = W.L;       // candidate list, both paths
    const int lane = threadIdx.x;
    const u32 nwords = (p.n + 31u) >> 5;
    const rsrc_t rrows = make_rows_rsrc(p.rows, (u64)p.n * p.stride32 * 4ull);
    u64 acc[FCM_COUNT_MAXT + 1];
#pragma unroll
    for (int q = 0; q <= FCM_COUNT_MAXT; ++q) acc[q] = 0ull;
    u32 overflow = 0, toolarge = 0;

    for (u64 e = blockIdx.x; e < p.m; e += gridDim.x) {
        const u32 u = p.edges[2 * e], v = p.edges[2 * e + 1];
        const u32 *ru = p.rows + (size_t)u * p.stride32;
        const u32 *rv = p.rows + (size_t)v * p.stride32;
        u32 total = 0;
        for (u32 w0 = 0; w0 < nwords; w0 += WAVE) {
            const u32 w = w0 + lane;
            u32 x = (w < nwords) ? (ru[w] & rv[w]) : 0u;
            const u32 c = __popc(x);
            u32 inc = c;  // inclusive scan over the wave
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const u32 y = __shfl_up(inc, o, WAVE);
                if (lane >= o) inc += y;
            }
            const u32 tot = rdlane(inc, WAVE - 1);
            u32 pos = total + inc - c;
            while (x) {
                const u32 b = __ffs((int)x) - 1;
                x &= x - 1;
                if (pos < WAVE * FCM_COUNT_MAXNW) Lc[pos] = w * 32u + b;
                ++pos;
            }
            total += tot;
        }
        if (total == 0) continue;
        if (total > WAVE * FCM_COUNT_MAXNW) {   // more than 256 common out-neighbours: left to the second pass (fcm_count_xw_kernel)
            if (lane == 0) {
                const u32 at = atomicAdd(&p.xlist[0], 1u);
                if (at < p.xcap) p.xlist[1 + at] = (u32)e;
            }
            continue;
        }
        wave_sync();
        if (total <= WAVE) {
            const u32 Lv = lane < (int)total ? Lc[lane] : 0u;
            const u64 myH = build_local(rrows, p.stride32, Lv, (int)total, lane);
            wave_sync();
            Hs[lane] = myH;
            wave_sync();
            // every vertex of C comes after v: one class, the raw masks are the clique graph.  64-bit counts per lane:
            // one edge of a dense graph can carry more than 2^31 simplices of the higher dimensions.
            if (lane < (int)total) acc[1] += 1;
            if (myH) visit<1, FCM_COUNT_MAXT, true, u64>(myH, Hs, FCM_COUNT_MAXT, +1, acc, overflow);
        } else {
            Wide V = W;
            V.NW = (int)((total + 63u) >> 6);
            wide_zero_counts(V, lane);
            wide_build(V, p.rows, p.stride32, (int)total, lane);
            if (lane < 12) V.cls[lane] = 0ull;
            wave_sync();
            if (lane < V.NW) {
                const int rem = (int)total - 64 * lane;
                V.cls[2 * 4 + lane] = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
            }
            wave_sync();
            wide_dfs<true>(V, FCM_COUNT_MAXT, +1, &overflow);
            if (lane == 0) {
#pragma unroll
                for (int q = 1; q <= FCM_COUNT_MAXT; ++q) acc[q] += (u64)V.cnt[q];
            }
        }
        wave_sync();
    }
#pragma unroll
    for (int q = 1; q <= FCM_COUNT_MAXT; ++q) {
        const u64 s = (u64)wave_sum_i64((long long)acc[q]);
        if (lane == 0 && s) atomicAdd((unsigned long long *)&p.counts[q + 1], (unsigned long long)s);
    }
    if (ballot(toolarge != 0u) && lane == 0) atomicOr(&p.flags[0], 1u);
    if (ballot(overflow != 0u) && lane == 0) atomicOr(&p.flags[1], 1u);
}

// Second pass: the directed edges whose endpoints have 257 .. 1024 common out-neighbours, one wave each, masks in a
// workspace in global memory (fcm_xwide.hpp).
__global__ __launch_bounds__(WAVE) void fcm_count_xw_kernel(const FcmCountParams p, u32 nflagged)
{
    const int lane = threadIdx.x;
    if (blockIdx.x >= nflagged) return;
    const u64 e = p.xlist[1 + blockIdx.x];
    u64 *ws = (u64 *)p.xw_ws + (size_t)blockIdx.x * FCM_XW_WORDS;
    const u32 nwords = (p.n + 31u) >> 5;
    const u32 u = p.edges[2 * e], v = p.edges[2 * e + 1];
    const u32 *ru = p.rows + (size_t)u * p.stride32, *rv = p.rows + (size_t)v * p.stride32;
    XWide X = xw_carve(ws, 64);
    u32 total = 0;
    for (u32 w0 = 0; w0 < nwords; w0 += WAVE) {   // compact out(u) & out(v) into the list
        const u32 w = w0 + lane;
        u32 x = (w < nwords) ? (ru[w] & rv[w]) : 0u;
        const u32 c = __popc(x);
        const u32 inc = (u32)wave_scan_i32((int)c);
        u32 pos = total + inc - c;
        while (x) {
            const u32 b = __ffs((int)x) - 1;
            x &= x - 1;
            if (pos < 64u * FCM_XW_MAXNW) X.L[pos] = w * 32u + b;
            ++pos;
        }
        total += rdlane(inc, WAVE - 1);
    }
    if (total > 64u * FCM_XW_MAXNW) { if (lane == 0) atomicOr(&p.flags[0], 1u); return; }
    if (lane < 16) X.cnt[lane] = 0;
    xw_sync();
    X.NW = (int)((total + 63u) >> 6);
    xw_build(X, p.rows, p.stride32, (int)total, lane);
    XCls c;   // every vertex comes after v: one class
    c.P = 0ull; c.M = 0ull;
    const int rem = (int)total - 64 * lane;
    c.S = lane < X.NW ? (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)) : 0ull;
    u32 overflow = 0u;
    xw_dfs(X, c, FCM_COUNT_MAXT, +1, lane, &overflow);
    if (lane >= 1 && lane <= FCM_COUNT_MAXT && X.cnt[lane]) atomicAdd((unsigned long long *)&p.counts[lane + 1], (unsigned long long)X.cnt[lane]);
    if (overflow && lane == 0) atomicOr(&p.flags[1], 1u);
}

// rows[c] = base for every chain c; 16 B per lane, coalesced.
__global__ __launch_bounds__(256) void fcm_broadcast_rows_kernel(uint4 *__restrict__ rows, const uint4 *__restrict__ base,
                                                               u64 vec_per_chain, u32 nchains)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < vec_per_chain; i += stride) {
        const uint4 v = base[i];
        for (u32 c = 0; c < nchains; ++c) rows[(u64)c * vec_per_chain + i] = v;
    }
}

// ---------------------------------------------------------------------------
// State::apply_transition as a callable (fcm_sampler_apply_transition): the induced adjacency of a vertex list
// (the reference's Graph::subgraph, src/lib.rs:63,71) read off one chain's bitmap, and set_edge on it (:68-70).
// Neither is on the stepping path.
// ---------------------------------------------------------------------------
// out[i][w] (u32 words, nlw per row): bit j of the row = list[i] -> list[j].  One wave per row.
__global__ __launch_bounds__(WAVE) void fcm_gather_sub_kernel(const u32 *__restrict__ rows, u32 stride32, const u32 *__restrict__ list, u32 nl,
                                                              u32 nlw, u32 *__restrict__ out)
{
    const int lane = threadIdx.x;
    const u32 i = blockIdx.x;
    if (i >= nl) return;
    const u32 *row = rows + (size_t)list[i] * stride32;
    for (u32 j0 = 0; j0 < nl; j0 += WAVE) {
        const u32 j = j0 + (u32)lane;
        bool bit = false;
        if (j < nl && j != i) { const u32 v = list[j]; bit = (row[v >> 5] >> (v & 31u)) & 1u; }
        const u64 m = ballot(bit);
        if (lane == 0) {
            out[(size_t)i * nlw + (j0 >> 5)] = (u32)m;
            if ((j0 >> 5) + 1u < nlw) out[(size_t)i * nlw + (j0 >> 5) + 1u] = (u32)(m >> 32);
        }
    }
}
// changes[c] = {from, to, present}: applied in order by one thread (an edge may be named twice)
__global__ void fcm_set_edges_kernel(u32 *rows, u32 stride32, const u32 *__restrict__ changes, u32 nchanges)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (u32 c = 0; c < nchanges; ++c) {
        const u32 a = changes[3 * c], b = changes[3 * c + 1];
        u32 *w = rows + (size_t)a * stride32 + (b >> 5);
        const u32 bit = 1u << (b & 31u);
        *w = changes[3 * c + 2] ? (*w | bit) : (*w & ~bit);
    }
}

// ---------------------------------------------------------------------------
// The batched State API: State::apply_transition / revert_transition (src/lib.rs:61-95) on every chain of the batch in one
// launch, one wave per chain -- what a search over many States at once needs (the reference's all_cxs runs 100 of them on OS
// threads, src/bin/all_cxs.rs:33-86).  The transition changes the directions of one adjacent pair; (pre, post) are the
// reference's vectors: flagser_count of the induced subgraph on N(a) cap N(b) + {a, b} before and after the set_edge calls
// (a full count of that subgraph, not the through-the-edge counts of the step kernels): the local build of the step kernels,
// then a walk over every ordered clique, lane = first vertex, 64-bit counts.
// ---------------------------------------------------------------------------
#define FCM_APPLY_MAXT 16   // cliques of up to 16 vertices: dimensions 0 .. 15, every entry a count vector of the ABI has
__device__ __forceinline__ int apply_count_local(u64 myH, u64 *Hs, int s, int lane, u64 (&tot)[FCM_APPLY_MAXT + 1])
{
    u64 acc[FCM_APPLY_MAXT + 1];
#pragma unroll
    for (int q = 0; q <= FCM_APPLY_MAXT; ++q) acc[q] = 0ull;
    u32 overflow = 0u;
    wave_sync();
    Hs[lane] = myH;
    wave_sync();
    if (lane < s) acc[1] = 1ull;                                         // the vertex itself: a clique of one
    if (myH) visit<1, FCM_APPLY_MAXT, true, u64>(myH, Hs, FCM_APPLY_MAXT, +1, acc, overflow);
    int len = 0;
#pragma unroll
    for (int q = 1; q <= FCM_APPLY_MAXT; ++q) {
        tot[q] = (u64)wave_sum_i64((long long)acc[q]);
        if (tot[q]) len = q;                                             // flagser_count stops at the last dimension present
    }
    return ballot(overflow != 0u) ? -1 : len;
}

__global__ __launch_bounds__(WAVE) void fcm_apply_batch_kernel(const FcmApplyParams p)
{
    __shared__ u64 Hs[WAVE];
    const int lane = threadIdx.x;
    const u32 c = blockIdx.x;
    if (c >= p.nchains) return;
    const u32 e = p.pair[c];
    if (e == FCM_TR_SKIP) return;
    const FcmEdgeEntry ent = p.etab[e];
    const u32 big = rdlane(ent.big, 0), small = rdlane(ent.small, 0), off = rdlane(ent.nb_off, 0);
    const int k = (int)rdlane(ent.k, 0), s = k + 2;
    u32 *rows = p.rows + (size_t)c * p.rows_per_chain;
    u64 *cnt = (u64 *)p.counts + (size_t)c * FCM_DEV_MAX_COUNTS;
    u64 *st = (u64 *)p.stats + (size_t)c * FCM_DEV_NSTATS;
    u32 *wbs = rows + (size_t)big * p.stride32 + (small >> 5), *wsb = rows + (size_t)small * p.stride32 + (big >> 5);
    const u32 bit_s = 1u << (small & 31u), bit_b = 1u << (big & 31u);
    const u32 op = p.ops[c], op_bs = op & 3u, op_sb = (op >> 2) & 3u;
    const u32 vbs = *wbs, vsb = *wsb;
    const u32 o_bs = (vbs & bit_s) ? 1u : 0u, o_sb = (vsb & bit_b) ? 1u : 0u;
    const u32 n_bs = op_bs == 0u ? o_bs : (op_bs == 1u ? 1u : 0u), n_sb = op_sb == 0u ? o_sb : (op_sb == 1u ? 1u : 0u);
    u32 status = FCM_TRS_OK;
    if (!(o_bs | o_sb)) status = FCM_TRS_TABLE;                          // the table says adjacent, the bitmap says not
    else if (!(n_bs | n_sb) || ((o_bs & o_sb) != (n_bs & n_sb))) status = FCM_TRS_UNSUPPORTED;   // pr(G) or the number of reciprocal pairs would change
    u64 pre[FCM_APPLY_MAXT + 1], post[FCM_APPLY_MAXT + 1];
    int lpre = 0, lpost = 0;
    if (!p.revert) {
        if (s > WAVE) { if (lane == 0) p.status[c] = FCM_TRS_HOST; return; }   // (left to the one-chain path)
        if (status == FCM_TRS_OK) {
            const rsrc_t rr = make_rows_rsrc(rows, p.rows_per_chain * 4ull);
            const u32 Lv = load_list(p.nb, off, k, big, small, lane);
            u64 myH = build_local(rr, p.stride32, Lv, s, lane);          // in-masks: lane j, bit i = L[i] -> L[j] (the same cliques, read backwards)
            lpre = apply_count_local(myH, Hs, s, lane, pre);
            // big = lane k, small = lane k + 1: big -> small is bit k of small's mask, small -> big bit k + 1 of big's
            if (lane == k + 1) myH = n_bs ? (myH | (1ull << k)) : (myH & ~(1ull << k));
            if (lane == k) myH = n_sb ? (myH | (1ull << (k + 1))) : (myH & ~(1ull << (k + 1)));
            lpost = apply_count_local(myH, Hs, s, lane, post);
            if (lpre < 0 || lpost < 0) status = FCM_TRS_DEEP;
        }
    } else {
        lpre = (int)p.lens[2 * c]; lpost = (int)p.lens[2 * c + 1];
#pragma unroll
        for (int q = 1; q <= FCM_APPLY_MAXT; ++q) { pre[q] = p.pre[(size_t)c * FCM_DEV_MAX_COUNTS + q - 1]; post[q] = p.post[(size_t)c * FCM_DEV_MAX_COUNTS + q - 1]; }
    }
    // flag_count -= sub (asserting, :64-67 / :85-88), resized up to the other vector's length, += add (:72-77 / :89-94): apply
    // subtracts pre and adds post, revert the other way round.  Lane d holds entry d.
    const int nc = (int)p.ncounts;
    const u64 len0 = st[FCM_STAT_COUNT_LEN_DEV];
    u64 mine = lane < nc ? cnt[lane] : 0ull, subv = 0ull, addv = 0ull;
    int lsub = p.revert ? lpost : lpre, ladd = p.revert ? lpre : lpost;
#pragma unroll
    for (int q = 1; q <= FCM_APPLY_MAXT; ++q) {
        if (lane == q - 1) { subv = p.revert ? post[q] : pre[q]; addv = p.revert ? pre[q] : post[q]; }
    }
    const bool in_sub = lane < lsub && lane < nc && lane < (int)len0, in_add = lane < ladd && lane < nc;
    if (status == FCM_TRS_OK && ballot(in_sub && mine < subv)) status = FCM_TRS_ASSERT;
    if (status == FCM_TRS_OK) {
        if (in_sub) mine -= subv;
        if (in_add) mine += addv;
        if (lane < nc) cnt[lane] = mine;
        if (lane == 0) {
            const u64 nl = (u64)(ladd < nc ? ladd : nc);
            if (nl > len0) st[FCM_STAT_COUNT_LEN_DEV] = nl;              // flag_count.resize: never shrinks
            if (n_bs != o_bs) *wbs = n_bs ? (vbs | bit_s) : (vbs & ~bit_s);
            if (n_sb != o_sb) *wsb = n_sb ? (vsb | bit_b) : (vsb & ~bit_b);
        }
    }
    if (!p.revert && lane < FCM_DEV_MAX_COUNTS) {
        u64 a = 0ull, b = 0ull;
#pragma unroll
        for (int q = 1; q <= FCM_APPLY_MAXT; ++q) if (lane == q - 1) { a = status == FCM_TRS_OK || status == FCM_TRS_ASSERT ? pre[q] : 0ull; b = status == FCM_TRS_OK || status == FCM_TRS_ASSERT ? post[q] : 0ull; }
        p.pre[(size_t)c * FCM_DEV_MAX_COUNTS + lane] = a;
        p.post[(size_t)c * FCM_DEV_MAX_COUNTS + lane] = b;
    }
    if (lane == 0) {
        if (!p.revert) { p.lens[2 * c] = (u32)(lpre > 0 ? lpre : 0); p.lens[2 * c + 1] = (u32)(lpost > 0 ? lpost : 0); }
        p.status[c] = status;
    }
}

// Transition::single_edge_flip (src/lib.rs:292-299) drawn on every chain at once: r = mulhi64(x, U + D) names a directed
// edge; out = {pair id, from, to}, or FCM_TR_SKIP for the empty transition.  One thread per chain.
__global__ void fcm_flip_draw_kernel(const FcmFlipDrawParams p)
{
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.nchains) return;
    u32 *o = p.out + 3 * (size_t)c;
    o[0] = FCM_TR_SKIP; o[1] = 0u; o[2] = 0u;
    const u64 M = (u64)p.U + p.D;
    if (M == 0) return;
    const u64 r = __umul64hi(p.x[c], M);
    if (r >= p.U) return;                                                // the second direction of a reciprocal pair
    const FcmEdgeEntry ent = p.etab[r];
    const u32 *rows = p.rows + (size_t)c * p.rows_per_chain;
    u32 bs, sb;
    if (p.sparse) {
        const u32 w = rows[(2 * r) >> 5];
        bs = (w >> ((2 * r) & 31)) & 1u; sb = (w >> ((2 * r + 1) & 31)) & 1u;
    } else {
        bs = (rows[(size_t)ent.big * p.stride32 + (ent.small >> 5)] >> (ent.small & 31u)) & 1u;
        sb = (rows[(size_t)ent.small * p.stride32 + (ent.big >> 5)] >> (ent.big & 31u)) & 1u;
    }
    if (bs == sb) { if (!bs) o[0] = 0xFFFFFFFEu; return; }               // reciprocal: empty transition; (absent from the bitmap: the host reports it)
    o[0] = (u32)r; o[1] = bs ? ent.big : ent.small; o[2] = bs ? ent.small : ent.big;
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
extern "C" int fcm_launch_apply_batch(const FcmApplyParams *p, void *stream)
{
    if (p->nchains == 0) return 0;
    hipLaunchKernelGGL(fcm_apply_batch_kernel, dim3(p->nchains), dim3(WAVE), 0, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}
extern "C" int fcm_launch_flip_draw(const FcmFlipDrawParams *p, void *stream)
{
    if (p->nchains == 0) return 0;
    hipLaunchKernelGGL(fcm_flip_draw_kernel, dim3((p->nchains + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}
extern "C" int fcm_launch_gather_sub(const uint32_t *rows, uint32_t stride32, const uint32_t *list, uint32_t nl, uint32_t nlw, uint32_t *out, void *stream)
{
    if (nl == 0) return 0;
    hipLaunchKernelGGL(fcm_gather_sub_kernel, dim3(nl), dim3(WAVE), 0, (hipStream_t)stream, rows, stride32, list, nl, nlw, out);
    return (int)hipGetLastError();
}
extern "C" int fcm_launch_set_edges(uint32_t *rows, uint32_t stride32, const uint32_t *changes, uint32_t nchanges, void *stream)
{
    if (nchanges == 0) return 0;
    hipLaunchKernelGGL(fcm_set_edges_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rows, stride32, changes, nchanges);
    return (int)hipGetLastError();
}

typedef int (*fcm_step_launcher)(const FcmStepParams *, void *);
#define FCM_DECL_STEP(tag) int fcm_launch_step_##tag##_0(const FcmStepParams *, void *); int fcm_launch_step_##tag##_1(const FcmStepParams *, void *);
extern "C" {
FCM_DECL_STEP(6) FCM_DECL_STEP(14) FCM_DECL_STEP(x2) FCM_DECL_STEP(x3) FCM_DECL_STEP(x4) FCM_DECL_STEP(x5) FCM_DECL_STEP(x6)
int fcm_launch_step_6_2(const FcmStepParams *, void *); int fcm_launch_step_14_2(const FcmStepParams *, void *);
int fcm_launch_step_m2_0(const FcmStepParams *, void *); int fcm_launch_step_m3_0(const FcmStepParams *, void *);
int fcm_launch_step_m4_0(const FcmStepParams *, void *); int fcm_launch_step_m5_0(const FcmStepParams *, void *);
int fcm_launch_step_m6_0(const FcmStepParams *, void *);
int fcm_launch_step_n2_0(const FcmStepParams *, void *); int fcm_launch_step_n3_0(const FcmStepParams *, void *);
int fcm_launch_step_n4_0(const FcmStepParams *, void *); int fcm_launch_step_n5_0(const FcmStepParams *, void *);
int fcm_launch_step_n6_0(const FcmStepParams *, void *);
int fcm_launch_step_s2_0(const FcmStepParams *, void *); int fcm_launch_step_s3_0(const FcmStepParams *, void *);
int fcm_launch_step_s4_0(const FcmStepParams *, void *); int fcm_launch_step_s5_0(const FcmStepParams *, void *);
int fcm_launch_step_s6_0(const FcmStepParams *, void *);
int fcm_launch_step_c2_1(const FcmStepParams *, void *); int fcm_launch_step_c3_1(const FcmStepParams *, void *);
int fcm_launch_step_c4_1(const FcmStepParams *, void *); int fcm_launch_step_c5_1(const FcmStepParams *, void *);
int fcm_launch_step_c6_1(const FcmStepParams *, void *);
}

// tmax = tracked depth (count entries - 2); clique: kernel variant with the clique moves
extern "C" int fcm_launch_step(const FcmStepParams *p, int tmax, int clique, void *stream)
{
    static const fcm_step_launcher exact[5][2] = {
        {fcm_launch_step_x2_0, fcm_launch_step_x2_1}, {fcm_launch_step_x3_0, fcm_launch_step_x3_1},
        {fcm_launch_step_x4_0, fcm_launch_step_x4_1}, {fcm_launch_step_x5_0, fcm_launch_step_x5_1},
        {fcm_launch_step_x6_0, fcm_launch_step_x6_1}};
    const int c = clique ? 1 : 0;
    if (clique == 2 && tmax >= 2 && tmax <= 6 && p->mw_waves >= 2) {   // several waves per chain, in-order commit (simple moves)
        static const fcm_step_launcher mc[5] = {fcm_launch_step_m2_0, fcm_launch_step_m3_0, fcm_launch_step_m4_0,
                                                fcm_launch_step_m5_0, fcm_launch_step_m6_0};   // rows of one cache line
        static const fcm_step_launcher nc[5] = {fcm_launch_step_n2_0, fcm_launch_step_n3_0, fcm_launch_step_n4_0,
                                                fcm_launch_step_n5_0, fcm_launch_step_n6_0};   // longer rows
        static const fcm_step_launcher sc[5] = {fcm_launch_step_s2_0, fcm_launch_step_s3_0, fcm_launch_step_s4_0,
                                                fcm_launch_step_s5_0, fcm_launch_step_s6_0};   // sparse state
        return (p->sparse ? sc : (p->stride32 == 32u ? mc : nc))[tmax - 2](p, stream);
    }
    if (clique == 3 && tmax >= 2 && tmax <= 6) {   // move mixes with clique moves: pairs on the pre-move bitmap, W waves per chain (fcm_step_cq.hpp)
        static const fcm_step_launcher cc[5] = {fcm_launch_step_c2_1, fcm_launch_step_c3_1, fcm_launch_step_c4_1, fcm_launch_step_c5_1, fcm_launch_step_c6_1};
        return cc[tmax - 2](p, stream);
    }
    if (c && p->xw_ws) return tmax <= 6 ? fcm_launch_step_6_2(p, stream) : fcm_launch_step_14_2(p, stream);   // clique moves on a graph with a local set beyond 256 vertices
    if (tmax >= 2 && tmax <= 6) return exact[tmax - 2][c](p, stream);
    if (tmax <= 6) return c ? fcm_launch_step_6_1(p, stream) : fcm_launch_step_6_0(p, stream);
    return c ? fcm_launch_step_14_1(p, stream) : fcm_launch_step_14_0(p, stream);
}

extern "C" int fcm_launch_count(const FcmCountParams *p, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (p->m == 0) return 0;
    const u64 maxgrid = 256ull * 32ull;
    dim3 grid((unsigned)(p->m < maxgrid ? p->m : maxgrid)), block(WAVE);
    hipLaunchKernelGGL(fcm_count_kernel, grid, block, sizeof(u64) * fcm_lds_words(FCM_COUNT_MAXNW), st, *p);
    return (int)hipGetLastError();
}

extern "C" int fcm_launch_count_xw(const FcmCountParams *p, uint32_t nflagged, void *stream)
{
    if (nflagged == 0) return 0;
    hipLaunchKernelGGL(fcm_count_xw_kernel, dim3(nflagged), dim3(WAVE), 0, (hipStream_t)stream, *p, nflagged);
    return (int)hipGetLastError();
}

extern "C" int fcm_launch_broadcast_rows(uint32_t *rows, const uint32_t *base, uint64_t words_per_chain,
                                         uint32_t nchains, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const u64 vec = words_per_chain / 4;  // rows are 128-B multiples
    u64 blocks = (vec + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(fcm_broadcast_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       (uint4 *)rows, (const uint4 *)base, vec, nchains);
    return (int)hipGetLastError();
}
