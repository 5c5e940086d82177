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
// permutation one LDS scatter; ADD = NEW & ~OLD and REM = OLD & ~NEW are the
// reference's change_edges (as a set).  The changes are then applied to the
// bitmap one at a time, each counted exactly like a simple move's edge
// (subtract E before a removal, add E after an addition); on a rejection the
// bits are put back (State::revert_transition, src/lib.rs:81-95).
#pragma once

#define FCM_CHG_ADD 0x80000000u
#define FCM_NOSLOT 0xFFFFFFFFu

struct CliqueLds {
    u32 *d;       // [32] vertices involved
    u32 *rowbuf;  // [32] scatter buffer for the row permutation
    u32 *oldm;    // [32] OLD rows
    u32 *newm;    // [32] NEW rows
    u32 *chg;     // [chg_cap][2] change list: from, to | FCM_CHG_ADD
};
// u64 words of LDS behind the evaluator region
__host__ __device__ inline unsigned fcm_clique_lds_words(unsigned chg_cap) { return 64u + chg_cap; }
__device__ __forceinline__ CliqueLds clique_carve(u64 *base)
{
    CliqueLds L;
    L.d = (u32 *)base;
    L.rowbuf = L.d + 32;
    L.oldm = L.d + 64;
    L.newm = L.d + 96;
    L.chg = L.d + 128;
    return L;
}

// one evaluation of an edge present in the bitmap: fast path, FCM_NEEDS_WIDE if it does not fit
template <int MAXT>
__device__ __forceinline__ int edge_eval(const rsrc_t rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big, u32 small,
                                         u32 fwd, int sign, u64 *Hs, u64 *Hp, int lane, int tmax, int (&delta)[MAXT + 1])
{
    const int s = k + 2;
    const u32 Lv = lane < k ? nb[off + lane] : (lane == k ? big : small);
    const u64 myH = build_local(rows, stride32, Lv, s, lane);
    Hs[lane] = myH;
    wave_sync();
    const int iu = fwd ? k : k + 1, iv = fwd ? k + 1 : k;
    const u32 present = (u32)((Hs[iu] >> iv) & 1ull);
    const Cls c = classify(myH, Hs, iu, iv);
    if (!extras_fit(c, s)) return FCM_NEEDS_WIDE;
    eval_nodes<MAXT>(myH, Hp, c, k, tmax, sign, lane, delta);
    return (int)present;
}
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
    int n_d;          // vertices involved
    u64 sum_k;
    long long wide_d; // this lane's share of deltas that came through the wide path
    u32 status;
};

// Builds the change list of a clique move and applies it to the bitmap, adding
// the simplex-count change to `delta` (fast evaluations) and res.wide_d (wide).
// move == 2: clique_permute, 3: clique_swap.
template <int MAXT>
__device__ __forceinline__ CliqueResult clique_propose(const FcmStepParams &p, u32 *rows, const rsrc_t rrows, u64 *smem, const CliqueLds CL, int move,
                                                       u32 w1, u64 x64, u64 step, u32 gchain, u32 k0, u32 k1, int lane, int tmax,
                                                       int maxnw, int (&delta)[MAXT + 1])
{
    CliqueResult res = {0, 0, 0ull, 0ll, 0u};
    u64 *Hs = smem, *Hp = smem + WAVE;
    const u32 stride32 = p.stride32;
    // ---- clique_order_distribution.sample, cliques.choose (src/lib.rs:215-216, 235-237)
    int oi = 0;
    while (oi < p.cl_orders - 1 && (u64)w1 >= p.cumo[oi]) ++oi;
    const int o = oi + 1;
    const u64 cnt = p.cl_count[oi];
    if (cnt == 0) { res.status = 16u; return res; }
    const u32 *bucket = p.clq + p.cl_base[oi];
    const u32 NONE = 0xFFFFFFFFu;
    const u32 m1 = lane < o ? bucket[__umul64hi(x64, cnt) * (u64)o + lane] : NONE;
    int n_c = o, n_a = 0, n_d = o;
    u32 dv = m1;  // d[lane]
    if (move == 3) {
        u32 v[4];
        philox4x32_10((u32)step, (u32)(step >> 32), gchain, 1u, k0, k1, v);
        const u32 m2 = lane < o ? bucket[__umul64hi((u64)v[0] | ((u64)v[1] << 32), cnt) * (u64)o + lane] : NONE;
        bool in2 = false, in1 = false;  // m1[lane] in m2, m2[lane] in m1
        for (int j = 0; j < o; ++j) {
            in2 = in2 || (m1 == rdlane(m2, j));
            in1 = in1 || (m2 == rdlane(m1, j));
        }
        const u64 omask = (1ull << o) - 1ull;
        const u64 cm1 = ballot(lane < o && in2), cm2 = ballot(lane < o && in1);
        n_c = __popcll(cm1);
        n_a = o - n_c;
        n_d = o + n_a;
        const u64 below = (1ull << lane) - 1ull;
        // d = c ++ (m1 - c) ++ (m2 - c), each part in its clique's order (vec_intersect / vec_setminus, src/util.rs:34-50)
        if (lane < o) {
            const int pos1 = in2 ? __popcll(cm1 & below) : n_c + __popcll(~cm1 & omask & below);
            CL.d[pos1] = m1;
            if (!in1) CL.d[n_c + n_a + __popcll(~cm2 & omask & below)] = m2;
        }
        wave_sync();
        dv = lane < n_d ? CL.d[lane] : NONE;
    }
    res.n_d = n_d;
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
    if (act) { CL.rowbuf[pd] = t; CL.oldm[lane] = oldr; }
    wave_sync();
    const u32 newr = act ? CL.rowbuf[lane] : 0u;
    if (act) CL.newm[lane] = newr;
    const u32 addm = newr & ~oldr, remm = oldr & ~newr;
    // ---- change list (src/lib.rs:226-228, 277-287), order immaterial
    const int mine = __popc(addm) + __popc(remm);
    int inc = mine;
#pragma unroll
    for (int sft = 1; sft < WAVE; sft <<= 1) {
        const int y = __shfl_up(inc, sft, WAVE);
        if (lane >= sft) inc += y;
    }
    const int nchg = (int)rdlane((u32)inc, WAVE - 1);
    res.nchg = nchg;
    if (nchg == 0) return res;
    if ((u32)nchg > p.chg_cap) { res.status = 32u; res.nchg = 0; return res; }
    // d[] in LDS for both moves (clique_permute has not written it yet)
    if (act) CL.d[lane] = dv;
    wave_sync();
    {
        int pos = inc - mine;
        for (u32 m = addm; m; m &= m - 1, ++pos) {
            CL.chg[2 * pos] = dv;
            CL.chg[2 * pos + 1] = CL.d[__ffs((int)m) - 1] | FCM_CHG_ADD;
        }
        for (u32 m = remm; m; m &= m - 1, ++pos) {
            CL.chg[2 * pos] = dv;
            CL.chg[2 * pos + 1] = CL.d[__ffs((int)m) - 1];
        }
    }
    wave_sync();
    // ---- apply one change at a time, counting each
    for (int c = 0; c < nchg; ++c) {
        const u32 from = CL.chg[2 * c], tw = CL.chg[2 * c + 1];
        const u32 to = tw & ~FCM_CHG_ADD;
        const bool add = (tw & FCM_CHG_ADD) != 0u;
        const u32 big = from > to ? from : to, small = from > to ? to : from;
        const u32 fwd = from > to ? 1u : 0u;
        const u32 e = find_pair(p.etab, p.efirst, big, small, lane);
        if (e == FCM_NOSLOT) { res.status |= 1u; continue; }
        const FcmEdgeEntry de = p.etab[e];
        const int k = (int)de.k;
        u32 *word = rows + (size_t)from * stride32 + (to >> 5);
        const u32 bit = 1u << (to & 31u);
        if (add) {
            if (lane == 0) *word |= bit;
            wave_sync();
        }
        int r = FCM_NEEDS_WIDE;
        if (k + 2 <= WAVE) r = edge_eval<MAXT>(rrows, stride32, p.nb, de.nb_off, k, big, small, fwd, add ? +1 : -1, Hs, Hp, lane, tmax, delta);
        if (r == FCM_NEEDS_WIDE) {
            if (k + 2 <= 64 * maxnw) {
                const Wide W = wide_carve(smem, maxnw);
                wide_zero_counts(W, lane);
                r = wide_edge(W, rows, stride32, p.nb, de.nb_off, k, big, small, fwd, add ? +1 : -1, lane, tmax) ? 1 : 0;
                if (lane >= 2 && lane < 16 && lane - 1 <= tmax) res.wide_d += W.cnt[lane - 1];
                wave_sync();
            } else {
                r = 0;
            }
        }
        if (r == 0) res.status |= 1u;  // the edge to count was not in the bitmap
        if (!add) {
            if (lane == 0) *word &= ~bit;
            wave_sync();
        }
        res.sum_k += (u64)k;
    }
    return res;
}

// Put the bitmap back after a rejected clique move.
__device__ __forceinline__ void clique_revert(u32 *rows, u32 stride32, const CliqueLds CL, int nchg, int lane)
{
    for (int c = lane; c < nchg; c += WAVE) {
        const u32 from = CL.chg[2 * c], tw = CL.chg[2 * c + 1];
        const u32 to = tw & ~FCM_CHG_ADD;
        u32 *word = rows + (size_t)from * stride32 + (to >> 5);
        const u32 bit = 1u << (to & 31u);
        if (tw & FCM_CHG_ADD) atomicAnd(word, ~bit); else atomicOr(word, bit);
    }
    wave_sync();
}

// After an accepted clique move: the reciprocal-pair slot list.  The i-th pair
// (ascending pair id) that stopped being reciprocal hands its slot to the i-th
// pair that became reciprocal (same rule in the oracle).  Returns a status bit.
__device__ __forceinline__ u32 clique_update_slots(const FcmStepParams &p, u32 *dbl, u32 *slot_of, const CliqueLds CL, int n_d, int lane)
{
    const bool act = lane < n_d;
    const u32 oldr = act ? CL.oldm[lane] : 0u, newr = act ? CL.newm[lane] : 0u;
    u32 oldt = 0u, newt = 0u;  // transposes: bit j = row j has bit `lane`
    for (int j = 0; j < n_d; ++j) {
        oldt |= ((CL.oldm[j] >> lane) & 1u) << j;
        newt |= ((CL.newm[j] >> lane) & 1u) << j;
    }
    const u32 upper = act ? ~((2u << lane) - 1u) : 0u;  // pairs once: j > lane
    const u32 was = oldr & oldt & upper, is = newr & newt & upper;
    const u32 lost = was & ~is, gained = is & ~was;
    const int nl_mine = __popc(lost), ng_mine = __popc(gained);
    int incl = nl_mine, incg = ng_mine;
#pragma unroll
    for (int sft = 1; sft < WAVE; sft <<= 1) {
        const int y = __shfl_up(incl, sft, WAVE), z = __shfl_up(incg, sft, WAVE);
        if (lane >= sft) { incl += y; incg += z; }
    }
    const int nl = (int)rdlane((u32)incl, WAVE - 1), ng = (int)rdlane((u32)incg, WAVE - 1);
    if (nl != ng) return 64u;
    if (nl == 0) return 0u;
    if (nl > WAVE || 2 * (u32)nl > 2 * p.chg_cap) return 128u;
    // pair ids into LDS (the change list is no longer needed): lostv[0..nl), gainv[0..nl)
    u32 *lostv = CL.chg, *gainv = CL.chg + nl;
    wave_sync();
    {
        int pl = incl - nl_mine, pg = incg - ng_mine;
        for (u32 m = lost; m; m &= m - 1) lostv[pl++] = ((u32)lane << 8) | (u32)(__ffs((int)m) - 1);   // (i, j) packed
        for (u32 m = gained; m; m &= m - 1) gainv[pg++] = ((u32)lane << 8) | (u32)(__ffs((int)m) - 1);
    }
    wave_sync();
    // resolve (i,j) -> pair id, uniformly
    for (int x = 0; x < 2 * nl; ++x) {
        const u32 ij = CL.chg[x];
        const u32 a = CL.d[ij >> 8], b = CL.d[ij & 0xFFu];
        const u32 e = find_pair(p.etab, p.efirst, a > b ? a : b, a > b ? b : a, lane);
        wave_sync();
        if (lane == 0) CL.chg[x] = e;
        wave_sync();
    }
    // rank within each list (ascending id), then hand over the slots
    const u32 ml = lane < nl ? lostv[lane] : 0u, mg = lane < nl ? gainv[lane] : 0u;
    int rl = 0, rg = 0;
    for (int y = 0; y < nl; ++y) {
        rl += (lostv[y] < ml) ? 1 : 0;
        rg += (gainv[y] < mg) ? 1 : 0;
    }
    wave_sync();
    u32 *sortl = CL.rowbuf, *sortg = CL.rowbuf + 32;  // rowbuf + oldm: 64 slots
    if (nl > 32) return 128u;
    if (lane < nl) { sortl[rl] = ml; sortg[rg] = mg; }
    wave_sync();
    if (lane < nl) {
        const u32 le = sortl[lane], ge = sortg[lane];
        const u32 slot = slot_of[le];
        dbl[slot] = ge;
        slot_of[ge] = slot;
        slot_of[le] = FCM_NOSLOT;
    }
    wave_sync();
    return 0u;
}
