// fcm_clique.hpp — device code for the clique moves, clique_permute and
// clique_swap (reference src/lib.rs:214-290).  Included by
// fcm_kernels_common.hpp; only the CLIQUE=true kernel variant instantiates it.
//
// Both moves transplant orientation patterns between the vertex pairs of one
// or two maximal cliques.  With d = the <= 32 vertices involved and pd the
// index permutation the reference calls perm / perm_d, the move is
//     NEW[pd[i]][pd[j]] = OLD[i][j]      over the pairs (i,j) it touches,
// where OLD is the current adjacency among d.  OLD is gathered as one 32-bit
// row mask per lane, the column permutation is a handful of bit moves, the row
// permutation one LDS scatter; NEW ^ OLD is the reference's change_edges (as a
// set).  The changes are applied to the bitmap one vertex pair at a time.  The
// number of simplices through a->b does not depend on whether b->a is there
// (a simplex holds a and b in one order only), and the vertex classes around
// the pair do not depend on the pair's own two bits, so one build of the
// pair's neighbourhood serves both of its directions: a removal subtracts
// E(a->b), an addition adds it, both on the graph with every earlier pair's
// changes in place -- the same telescoping sum as the reference's edge-by-edge
// State::apply_transition.  Pair ids come from the static per-clique table
// clq_pairs; on a rejection the bits are put back (State::revert_transition,
// src/lib.rs:81-95).
#pragma once

#define FCM_CHG_ADD 0x80000000u
#define FCM_NOSLOT 0xFFFFFFFFu
#ifdef FCM_STAMP
#define CLQ_STAMP(slot) do { const u64 _n = fcm_stamp(); sacc[slot] += _n - *stt; *stt = _n; } while (0)
#else
#define CLQ_STAMP(slot) do { } while (0)
#endif

struct CliqueLds {
    u32 *d;       // [32] vertices involved
    u32 *rowbuf;  // [32] scatter buffer for the row permutation
    u32 *oldm;    // [32] OLD rows
    u32 *newm;    // [32] NEW rows
    u32 *p1;      // [32] position of d[x] in the first clique (FCM_NOSLOT: not a member)
    u32 *p2;      // [32] ... in the second clique
    u32 *chg;     // changed vertex pairs, 4 words each: i | j<<8 | old<<16 | new<<20 (d indices i<j; bit 0 = i->j, bit 1 = j->i),
                  // etab index, k, nb_off
};
// u64 words of LDS behind the evaluator region
__host__ __device__ inline unsigned fcm_clique_lds_words(unsigned chg_cap) { return 96u + chg_cap; }
__device__ __forceinline__ CliqueLds clique_carve(u64 *base)
{
    CliqueLds L;
    L.d = (u32 *)base;
    L.rowbuf = L.d + 32;
    L.oldm = L.d + 64;
    L.newm = L.d + 96;
    L.p1 = L.d + 128;
    L.p2 = L.d + 160;
    L.chg = L.d + 192;
    return L;
}

// one evaluation of an edge present in the bitmap on the wide path
__device__ __forceinline__ bool wide_edge(const Wide W, const u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big,
                                          u32 small, u32 fwd, int sign, int lane, int tmax)
{
    const int s = k + 2;
    wide_load_list(W, nb, off, k, big, small, lane);
    wide_build(W, rows, stride32, s, lane);
    const int iu = fwd ? k : k + 1, iv = fwd ? k + 1 : k;
    const bool present = wide_has(W, iu, iv);
    wide_classify(W, iu, iv, s, lane);
    wide_dfs<false>(W, tmax, sign, nullptr);
    return present;
}

// etab index of the adjacent pair (big, small): etab is sorted by (big, small),
// efirst[v] = first index with big == v; the lanes scan that vertex's run.
__device__ __forceinline__ u32 find_pair(const FcmEdgeEntry *etab, const u32 *efirst, u32 big, u32 small, int lane)
{
    const u32 lo = efirst[big], hi = efirst[big + 1];
    for (u32 base = lo; base < hi; base += WAVE) {
        const u32 idx = base + lane;
        const u32 sm = idx < hi ? etab[idx].small : 0xFFFFFFFFu;
        const u64 m = ballot(sm == small);
        if (m) return base + (u32)__ffsll((long long)m) - 1u;
    }
    return FCM_NOSLOT;
}

// j = floor(word * m / 2^32)
__device__ __forceinline__ u32 mulhi32(u32 w, u32 m) { return __umulhi(w, m); }

struct CliqueResult {
    int nchg;         // directed edges changed (0 = empty transition)
    int npairs;       // vertex pairs with a change (entries of CL.chg)
    u64 sum_k;
    long long wide_d; // this lane's share of deltas that came through the wide path
    u32 status;
    u32 n_wide;       // pairs of this move that took a multi-word evaluator
    int n_d;          // vertices involved (entries of CL.d)
    u32 oldt, newt;   // per lane t < n_d: in-masks over d before / after the move (bit u = d[u] -> d[t]), among the touched pairs
};

// One direction of a changed pair on the wide evaluator (its counts are 64-bit): the edge is in the bitmap while it is
// counted -- set before an addition, cleared after a removal.  The count changes go to res.wide_d.
__device__ __forceinline__ void clique_dir_wide(u64 *smem, int maxnw, u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big, u32 small,
                                                u32 fwd, bool add, u32 *word, u32 bit, int lane, int tmax, CliqueResult &res)
{
    const Wide W = wide_carve(smem, maxnw);
    if (add) {
        if (lane == 0) *word |= bit;
        wave_sync();
    }
    wide_zero_counts(W, lane);
    if (!wide_edge(W, rows, stride32, nb, off, k, big, small, fwd, add ? +1 : -1, lane, tmax)) res.status |= 1u;
    if (lane >= 2 && lane < 16 && lane - 1 <= tmax) res.wide_d += W.cnt[lane - 1];
    wave_sync();
    if (!add) {
        if (lane == 0) *word &= ~bit;
        wave_sync();
    }
    res.n_wide += 1u;
}

// What the setup of a clique move needs of the launch parameters (the cooperative kernel hands them over from LDS).
struct CliqueTables {
    const u32 *clq, *clq_pairs;
    const FcmEdgeEntry *etab;
    const uint64_t *cl_base, *cl_count, *clp_base, *cumo;   // [FCM_DEV_MAX_COUNTS] each
    int cl_orders;
    u32 chg_cap, stride32;
};
__device__ __forceinline__ CliqueTables clique_tables(const FcmStepParams &p)
{
    return CliqueTables{p.clq, p.clq_pairs, p.etab, p.cl_base, p.cl_count, p.clp_base, p.cumo, p.cl_orders, p.chg_cap, p.stride32};
}

// The proposal itself (src/lib.rs:214-290): picks the clique(s), draws the permutation(s), gathers OLD, forms NEW and
// lists the vertex pairs whose orientation changes in CL.chg (4 words per pair: d indices and old / new 2-bit states,
// etab index, k, nb_off), ascending (i, j).  Nothing is written to the bitmap.  move == 2: clique_permute, 3: clique_swap.
__device__ __forceinline__ CliqueResult clique_setup(const CliqueTables &p, const u32 *rows, const CliqueLds CL, int move, u32 w1, u64 x64, u64 step,
                                                     u32 gchain, u32 k0, u32 k1, int lane, u64 *sacc, u64 *stt)
{
    CliqueResult res = {0, 0, 0ull, 0ll, 0u, 0u, 0, 0u, 0u};
    const u32 stride32 = p.stride32;
    // ---- clique_order_distribution.sample, cliques.choose (src/lib.rs:215-216, 235-237)
    // (the tables may sit in LDS -- the cooperative kernel -- where the compiler takes loaded values for per-lane ones:
    //  everything read from them is pinned wave-uniform, or the loops below would run under exec masks)
    auto uni32 = [](u32 v) -> u32 { return (u32)__builtin_amdgcn_readfirstlane((int)v); };
    auto uni64 = [&](u64 v) -> u64 { return (u64)uni32((u32)v) | ((u64)uni32((u32)(v >> 32)) << 32); };
    const int n_orders = (int)uni32((u32)p.cl_orders);
    int oi = 0;
    while (oi < n_orders - 1 && (u64)w1 >= uni64(p.cumo[oi])) ++oi;
    oi = (int)uni32((u32)oi);
    const int o = oi + 1;
    const u64 cnt = uni64(p.cl_count[oi]);
    if (cnt == 0) { res.status = 16u; return res; }
    const u32 *bucket = p.clq + uni64(p.cl_base[oi]);
    const u32 NONE = 0xFFFFFFFFu;
    const u64 c1 = __umul64hi(x64, cnt);
    u64 c2 = c1;
    const u32 m1 = lane < o ? bucket[c1 * (u64)o + lane] : NONE;
    int n_c = o, n_a = 0, n_d = o;
    u32 dv = m1;  // d[lane]
    if (lane < 32) { CL.p1[lane] = lane < o ? (u32)lane : NONE; CL.p2[lane] = NONE; }
    if (move == 3) {
        u32 v[4];
        philox4x32_10((u32)step, (u32)(step >> 32), gchain, 1u, k0, k1, v);
        c2 = __umul64hi((u64)v[0] | ((u64)v[1] << 32), cnt);
        const u32 m2 = lane < o ? bucket[c2 * (u64)o + lane] : NONE;
        bool in2 = false, in1 = false;  // m1[lane] in m2, m2[lane] in m1
        int j1 = 0;                     // where m2[lane] sits in m1
        for (int j = 0; j < o; ++j) {
            in2 = in2 || (m1 == rdlane(m2, j));
            const bool hit = m2 == rdlane(m1, j);
            in1 = in1 || hit;
            if (hit) j1 = j;
        }
        const u64 omask = (1ull << o) - 1ull;
        const u64 cm1 = ballot(lane < o && in2), cm2 = ballot(lane < o && in1);
        n_c = __popcll(cm1);
        n_a = o - n_c;
        n_d = o + n_a;
        const u64 below = (1ull << lane) - 1ull;
        wave_sync();
        // d = c ++ (m1 - c) ++ (m2 - c), each part in its clique's order (vec_intersect / vec_setminus, src/util.rs:34-50)
        if (lane < o) {
            const int pos1 = in2 ? __popcll(cm1 & below) : n_c + __popcll(~cm1 & omask & below);
            CL.d[pos1] = m1;
            CL.p1[pos1] = (u32)lane;
            const int pos2 = in1 ? __popcll(cm1 & ((1ull << j1) - 1ull)) : n_c + n_a + __popcll(~cm2 & omask & below);
            if (!in1) CL.d[pos2] = m2;
            CL.p2[pos2] = (u32)lane;
        }
        wave_sync();
        dv = lane < n_d ? CL.d[lane] : NONE;
    }
    // ---- perm / perm_d (random_perm = Fisher-Yates over Philox words, blocks sub = 2, 3, ...)
    u32 ws[4];
    philox4x32_10((u32)step, (u32)(step >> 32), gchain, 2u + (u32)lane, k0, k1, ws);  // lane l holds words 4l..4l+3
    // pd[pos]: c part keeps to c; the a positions receive b indices and vice versa (perm_d = perm_c ++ perm_b ++ perm_a)
    u32 pd = (u32)lane;
    if (lane >= n_c && lane < n_c + n_a) pd = (u32)(lane + n_a);
    else if (lane >= n_c + n_a) pd = (u32)(lane - n_a);
    int q = 0;
    for (int seg = 0; seg < 3; ++seg) {
        // word order as in the reference: perm_c, then perm_a (seated at the b positions' slots), then perm_b
        const int len = seg == 0 ? n_c : n_a;
        const int pos0 = seg == 0 ? 0 : (seg == 1 ? n_c + n_a : n_c);
        for (int i = len - 1; i >= 1; --i, ++q) {
            const int wl = q >> 2, wr = q & 3;
            const u32 word = wr == 0 ? rdlane(ws[0], wl) : (wr == 1 ? rdlane(ws[1], wl) : (wr == 2 ? rdlane(ws[2], wl) : rdlane(ws[3], wl)));
            const int j = (int)mulhi32(word, (u32)(i + 1));
            const u32 a = rdlane(pd, pos0 + i), b = rdlane(pd, pos0 + j);
            pd = wrlane(b, pos0 + i, pd);
            pd = wrlane(a, pos0 + j, pd);
        }
        if (move == 2) break;  // clique_permute: one permutation of the whole clique
    }
    CLQ_STAMP(1);                                                      // clique pick, d, permutations
    // ---- OLD: adjacency among d over the touched pairs
    const bool act = lane < n_d;
    const bool in_a = lane >= n_c && lane < n_c + n_a, in_b = lane >= n_c + n_a && lane < n_d;
    u32 oldr = 0u;
    {
        const u32 *myrow = rows + (size_t)(act ? dv : 0u) * stride32;
#pragma unroll
        for (int j = 0; j < 32; ++j) {  // unrolled so that the gathers are all in flight together
            if (j < n_d) {
                const u32 dj = rdlane(dv, j);
                const bool j_a = j >= n_c && j < n_c + n_a, j_b = j >= n_c + n_a;
                const bool valid = act && j != lane && !((in_a && j_b) || (in_b && j_a));
                const u32 wv = valid ? myrow[dj >> 5] : 0u;
                oldr |= ((wv >> (dj & 31u)) & 1u) << j;
            }
        }
    }
    // ---- NEW[pd[i]] bit pd[j] = OLD[i] bit j
    u32 t = 0u;
    for (int j = 0; j < n_d; ++j) {
        const u32 pj = rdlane(pd, j);
        t |= ((oldr >> j) & 1u) << pj;
    }
    if (act) { CL.rowbuf[pd] = t; CL.oldm[lane] = oldr; CL.d[lane] = dv; }
    wave_sync();
    const u32 newr = act ? CL.rowbuf[lane] : 0u;
    if (act) CL.newm[lane] = newr;
    // ---- change_edges (src/lib.rs:226-228, 277-287) = OLD ^ NEW, grouped per vertex pair
    const int nchg = wave_sum_i32(__popc(oldr ^ newr));
    res.nchg = nchg;
    if (nchg == 0) return res;
    wave_sync();
    u32 oldt = 0u, newt = 0u;  // transposes: bit j = row j has bit `lane`
    for (int j = 0; j < n_d; ++j) {
        oldt |= ((CL.oldm[j] >> lane) & 1u) << j;
        newt |= ((CL.newm[j] >> lane) & 1u) << j;
    }
    res.n_d = n_d; res.oldt = oldt; res.newt = newt;
    const u32 upper = act ? ~((2u << lane) - 1u) : 0u;  // each pair once: j > lane
    const u32 chm = ((oldr ^ newr) | (oldt ^ newt)) & upper;
    const int mine = __popc(chm);
    int inc = mine;
#pragma unroll
    for (int sft = 1; sft < WAVE; sft <<= 1) {
        const int y = __shfl_up(inc, sft, WAVE);
        if (lane >= sft) inc += y;
    }
    const int npairs = (int)rdlane((u32)inc, WAVE - 1);
    if (4u * (u32)npairs > 2u * p.chg_cap) { res.status = 32u; res.nchg = 0; return res; }
    res.npairs = npairs;
    {
        int pos = inc - mine;
        for (u32 m = chm; m; m &= m - 1, ++pos) {
            const int j = __ffs((int)m) - 1;
            const u32 o2 = ((oldr >> j) & 1u) | (((oldt >> j) & 1u) << 1), n2 = ((newr >> j) & 1u) | (((newt >> j) & 1u) << 1);
            CL.chg[4 * pos] = (u32)lane | ((u32)j << 8) | (o2 << 16) | (n2 << 20);
        }
    }
    wave_sync();
    CLQ_STAMP(2);                                                      // OLD gather, NEW, pair list
    // ---- table entries of all changed pairs, in parallel
    {
        const u32 *ptab = p.clq_pairs + uni64(p.clp_base[oi]);
        const u32 npo = (u32)(o * (o - 1) / 2);
        bool bad = false;
        for (int x = lane; x < npairs; x += WAVE) {
            const u32 w0 = CL.chg[4 * x];
            const u32 di = w0 & 0xFFu, dj = (w0 >> 8) & 0xFFu;
            u32 pa = CL.p1[di], pb = CL.p1[dj];
            u64 ci = c1;
            if (pa == NONE || pb == NONE) { pa = CL.p2[di]; pb = CL.p2[dj]; ci = c2; }
            const u32 lo = pa < pb ? pa : pb, hi = pa < pb ? pb : pa;
            u32 e = 0u;
            if (hi < (u32)o && lo != hi) e = ptab[ci * npo + lo * (u32)o - lo * (lo + 1u) / 2u + (hi - lo - 1u)];
            else bad = true;
            const FcmEdgeEntry ent = p.etab[e];
            const u32 a = CL.d[di], b = CL.d[dj];
            bad = bad || ent.big != (a > b ? a : b) || ent.small != (a > b ? b : a);
            CL.chg[4 * x + 1] = e;
            CL.chg[4 * x + 2] = ent.k;
            CL.chg[4 * x + 3] = ent.nb_off;
        }
        if (ballot(bad)) { res.status = 1u; res.nchg = 0; res.npairs = 0; return res; }
    }
    wave_sync();
    CLQ_STAMP(3);                                                      // pair ids + table entries
    return res;
}

// Builds the changed-pair list of a clique move and applies it to the bitmap,
// adding the simplex-count change to `delta` (fast evaluations) and res.wide_d
// (wide).  move == 2: clique_permute, 3: clique_swap.  (The one-wave kernel: a pair's change is stored before the next
// pair's rows are read, and a rejected move is put back; the cooperative kernel, fcm_step_cq.hpp, evaluates every pair
// on the pre-move bitmap instead.)
template <int MAXT, bool XW>
__device__ __forceinline__ CliqueResult clique_propose(const FcmStepParams &p, u32 *rows, const rsrc_t rrows, u64 *smem, const CliqueLds CL, int move,
                                                       u32 w1, u64 x64, u64 step, u32 gchain, u32 k0, u32 k1, int lane, int tmax,
                                                       int maxnw, fcm_acc_t<MAXT> (&delta)[MAXT + 1], EvScal &es, u64 *sacc, u64 *stt, FcmGuard *guard = nullptr)
{
    const CliqueTables T = clique_tables(p);
    CliqueResult res = clique_setup(T, rows, CL, move, w1, x64, step, gchain, k0, k1, lane, sacc, stt);
    if (res.nchg == 0) return res;
    const int npairs = res.npairs;
#if FCM_NB_KARG
    const u32 *const nbk = (const u32 *)fcm_karg64<offsetof(FcmStepParams, nb)>();   // (re-read once per move: see fcm_karg64)
#else
    const u32 *const nbk = p.nb;
#endif
    u64 *Hp = smem + WAVE;
    const u32 stride32 = p.stride32;
    // ---- apply one pair at a time, counting each directed change
    for (int x = 0; x < npairs; ++x) {
        const u32 w0 = CL.chg[4 * x], off = CL.chg[4 * x + 3];
        const int k = (int)CL.chg[4 * x + 2];
        const u32 a = CL.d[w0 & 0xFFu], b = CL.d[(w0 >> 8) & 0xFFu];
        const u32 o2 = (w0 >> 16) & 3u, n2 = (w0 >> 20) & 3u;
        const bool agb = a > b;
        const u32 big = agb ? a : b, small = agb ? b : a;
        // directions in (big, small) terms
        const u32 o_bs = agb ? (o2 & 1u) : (o2 >> 1), o_sb = agb ? (o2 >> 1) : (o2 & 1u);
        const u32 n_bs = agb ? (n2 & 1u) : (n2 >> 1), n_sb = agb ? (n2 >> 1) : (n2 & 1u);
        const bool need_bs = o_bs != n_bs, need_sb = o_sb != n_sb;
        u32 *wbs = rows + (size_t)big * stride32 + (small >> 5), *wsb = rows + (size_t)small * stride32 + (big >> 5);
        const u32 bit_s = 1u << (small & 31u), bit_b = 1u << (big & 31u);
        const u32 vbs = *wbs, vsb = *wsb;  // in flight with the list and the rows
        res.sum_k += (u64)k * (u64)((need_bs ? 1 : 0) + (need_sb ? 1 : 0));
        bool done = false;
        const int s = k + 2;
        if (s <= WAVE) {
            const u32 Lv = load_list(nbk, off, k, big, small, lane);
            const u64 myH = build_local(rrows, stride32, Lv, s, lane);
            CLQ_STAMP(4);                                              // per pair: list + build
            // in-masks: run on the transposed graph (see build_local)
            const u64 inB = rdlane64(myH, k), inS = rdlane64(myH, k + 1);
            if ((u32)((inS >> k) & 1ull) != o_bs || (u32)((inB >> (k + 1)) & 1ull) != o_sb) res.status |= 1u;
            const u64 outB = ballot((myH >> k) & 1ull), outS = ballot((myH >> (k + 1)) & 1ull);
            const u64 nbm = ~(3ull << k);
            Cls cbs, csb;  // classes around big->small and around small->big, as classify(.., k+1, k) / (.., k, k+1) give them
            cbs.P = csb.P = outB & outS & nbm;
            cbs.S = csb.S = inB & inS & nbm;
            cbs.M = inS & outB & nbm;
            csb.M = inB & outS & nbm;
            bool merged = false;
            if constexpr (MAXT <= 6) {   // both directions change: the pair is flipped -- one signed evaluation for the two (eval_flip_merged)
                if (tmax == MAXT && need_bs && need_sb && (o_bs ^ o_sb) != 0u) {
                    const u64 MA = o_bs ? cbs.M : csb.M, MB = o_bs ? csb.M : cbs.M;
                    merged = flip_merged_fits(cbs.P, MA, MB, cbs.S, k) && eval_flip_merged<MAXT>(myH, Hp, cbs.P, MA, MB, cbs.S, k, lane, delta, es);
                }
            }
            if (merged) {
                if (lane == 0) {
                    *wbs = n_bs ? (vbs | bit_s) : (vbs & ~bit_s);
                    *wsb = n_sb ? (vsb | bit_b) : (vsb & ~bit_b);
                }
                wave_sync();
                done = true;
            } else if ((!need_bs || extras_fit(cbs, s)) && (!need_sb || extras_fit(csb, s))) {
                if (need_bs) eval_nodes<MAXT>(myH, Hp, cbs, k, tmax, n_bs ? +1 : -1, lane, delta, es, nullptr, nullptr, guard);
                if constexpr (MAXT >= 7) {   // the bound on the 32-bit counts was passed: this direction on the wide path (64-bit counts)
                    if (need_bs && guard && guard->tripped) {
                        guard->tripped = 0u;
                        clique_dir_wide(smem, maxnw, rows, stride32, p.nb, off, k, big, small, 1u, n_bs != 0u, wbs, bit_s, lane, tmax, res);
                    }
                }
                if (need_sb) eval_nodes<MAXT>(myH, Hp, csb, k, tmax, n_sb ? +1 : -1, lane, delta, es, nullptr, nullptr, guard);
                if constexpr (MAXT >= 7) {
                    if (need_sb && guard && guard->tripped) {
                        guard->tripped = 0u;
                        clique_dir_wide(smem, maxnw, rows, stride32, p.nb, off, k, big, small, 0u, n_sb != 0u, wsb, bit_b, lane, tmax, res);
                    }
                }
                if (lane == 0) {
                    if (need_bs) *wbs = n_bs ? (vbs | bit_s) : (vbs & ~bit_s);
                    if (need_sb) *wsb = n_sb ? (vsb | bit_b) : (vsb & ~bit_b);
                }
                wave_sync();
                done = true;
            }
            CLQ_STAMP(5);                                              // per pair: evaluations + stores
        }
        // (XW: kernel variants 6_2 / 14_2, chosen when the graph has such a pair: inlined into every clique kernel this rare
        //  path costs 3 % in allocation quality, as a call 9 %)
        if constexpr (XW) if (!done && s > 64 * maxnw) {
            // 257 .. 1024 local vertices (a hub pair): the evaluator with its masks in the chain's HBM workspace (fcm_xwide.hpp)
            u64 *xw = p.xw_ws ? (u64 *)p.xw_ws + (size_t)blockIdx.x * FCM_XW_WORDS : nullptr;
            if (!xw || s > 64 * FCM_XW_MAXNW) { res.status |= 1u; continue; }
            res.n_wide += 1u;
            bool first = true;
            for (int dir = 0; dir < 2; ++dir) {
                if (!(dir == 0 ? need_bs : need_sb)) continue;
                const bool add = (dir == 0 ? n_bs : n_sb) != 0u;
                u32 *word = dir == 0 ? wbs : wsb;
                const u32 bit = dir == 0 ? bit_s : bit_b;
                if (add) {
                    if (lane == 0) *word |= bit;
                    wave_sync();
                }
                if (!xw_edge(xw, rows, stride32, p.nb, off, k, big, small, dir == 0 ? 1u : 0u, add ? +1 : -1, lane, tmax, !first)) res.status |= 1u;
                first = false;
                if (!add) {
                    if (lane == 0) *word &= ~bit;
                    wave_sync();
                }
            }
            if (lane >= 2 && lane < 16 && lane - 1 <= tmax) res.wide_d += xw_count(xw, lane - 1);
            wave_sync();
            continue;
        }
        if (!done) {
            if (s > 64 * maxnw) { res.status |= 1u; continue; }
            res.n_wide += 1u;
            const Wide W = wide_carve(smem, maxnw);
            for (int dir = 0; dir < 2; ++dir) {
                if (!(dir == 0 ? need_bs : need_sb)) continue;
                const bool add = (dir == 0 ? n_bs : n_sb) != 0u;
                u32 *word = dir == 0 ? wbs : wsb;
                const u32 bit = dir == 0 ? bit_s : bit_b;
                if (add) {
                    if (lane == 0) *word |= bit;
                    wave_sync();
                }
                wide_zero_counts(W, lane);
                if (!wide_edge(W, rows, stride32, p.nb, off, k, big, small, dir == 0 ? 1u : 0u, add ? +1 : -1, lane, tmax)) res.status |= 1u;
                if (lane >= 2 && lane < 16 && lane - 1 <= tmax) res.wide_d += W.cnt[lane - 1];
                wave_sync();
                if (!add) {
                    if (lane == 0) *word &= ~bit;
                    wave_sync();
                }
            }
        }
    }
    return res;
}

// Put the bitmap back after a rejected clique move.
__device__ __forceinline__ void clique_revert(u32 *rows, u32 stride32, const CliqueLds CL, int npairs, int lane)
{
    for (int x = lane; x < npairs; x += WAVE) {
        const u32 w0 = CL.chg[4 * x];
        const u32 a = CL.d[w0 & 0xFFu], b = CL.d[(w0 >> 8) & 0xFFu];
        const u32 o2 = (w0 >> 16) & 3u, ch = o2 ^ ((w0 >> 20) & 3u);
        if (ch & 1u) {
            u32 *word = rows + (size_t)a * stride32 + (b >> 5);
            const u32 bit = 1u << (b & 31u);
            if (o2 & 1u) atomicOr(word, bit); else atomicAnd(word, ~bit);
        }
        if (ch & 2u) {
            u32 *word = rows + (size_t)b * stride32 + (a >> 5);
            const u32 bit = 1u << (a & 31u);
            if (o2 & 2u) atomicOr(word, bit); else atomicAnd(word, ~bit);
        }
    }
    wave_sync();
}

// After an accepted clique move: the reciprocal-pair slot list.  The i-th pair
// (ascending pair id) that stopped being reciprocal hands its slot to the i-th
// pair that became reciprocal (same rule in the oracle).  Returns a status bit.
__device__ __forceinline__ u32 clique_update_slots(u32 *dbl, u32 *slot_of, const CliqueLds CL, int npairs, int lane)
{
    u32 *lostv = CL.rowbuf, *gainv = CL.newm;  // 64 entries each (rowbuf+oldm, newm+p1: no longer needed)
    int nl = 0, ng = 0;
    wave_sync();
    for (int base = 0; base < npairs; base += WAVE) {
        const int x = base + lane;
        const u32 w0 = x < npairs ? CL.chg[4 * x] : 0u;
        const u32 o2 = (w0 >> 16) & 3u, n2 = (w0 >> 20) & 3u;
        const bool lost = o2 == 3u && n2 != 3u, gained = n2 == 3u && o2 != 3u;
        const u64 ml = ballot(lost), mg = ballot(gained);
        const u64 below = (1ull << lane) - 1ull;
        const int pl = nl + __popcll(ml & below), pg = ng + __popcll(mg & below);
        nl += __popcll(ml);
        ng += __popcll(mg);
        if (nl > WAVE || ng > WAVE) return 128u;
        if (lost) lostv[pl] = CL.chg[4 * x + 1];
        if (gained) gainv[pg] = CL.chg[4 * x + 1];
    }
    if (nl != ng) return 64u;
    if (nl == 0) return 0u;
    wave_sync();
    // rank within each list (ascending id), then hand over the slots
    const u32 ml = lane < nl ? lostv[lane] : 0u, mg = lane < nl ? gainv[lane] : 0u;
    int rl = 0, rg = 0;
    for (int y = 0; y < nl; ++y) {
        rl += (lostv[y] < ml) ? 1 : 0;
        rg += (gainv[y] < mg) ? 1 : 0;
    }
    wave_sync();
    if (lane < nl) gainv[rg] = mg;  // sorted in place (every lane holds its entry)
    wave_sync();
    if (lane < nl) {
        const u32 ge = gainv[rl];
        const u32 slot = slot_of[ml];
        dbl[slot] = ge;
        slot_of[ge] = slot;
        slot_of[ml] = FCM_NOSLOT;
    }
    wave_sync();
    return 0u;
}
