// fcm_step_mw.hpp — the step kernel with W waves per chain (simple moves;
// included by fcm_step_variant.hip for the m<depth> tags).
//
// Proposals of one chain are strictly ordered (reference MCMCSampler::next,
// src/lib.rs:181-194), but nearly always independent: proposal q reads the
// orientation bits among the ~40 vertices of its local set(s), and the commit
// of an earlier proposal matters to it only if that commit's pair lies inside
// one of those sets (or touches the same slot of the reciprocal list / one of
// the candidate pairs it looked at).  So the W waves of a workgroup run W
// consecutive proposals of the chain at once -- wave w takes q = w, w+W, ... --
// each from the list load to the per-dimension count changes, on whatever state
// is committed when it starts, and the decisions are then taken strictly in
// order:
//
//   head (LDS)   number of proposals decided so far.  A wave notes head before
//                its first read of mutable state (snap) and, when its counts
//                are ready, waits until head == q: it then holds the token.
//   log  (LDS)   ring of the last W decisions: accepted?, the pair(s) changed,
//                the slot of the reciprocal list rewritten.
//   token holder checks the log entries snap..q-1 against its own reads.  No
//                hit (the rule, > 99 %): its counts are those of the exact
//                sequential state; bounds check against the chain's counts (LDS),
//                commit (read-modify-write of two bitmap words, the slot list),
//                log entry, head = q+1 (release).  A hit, or a proposal that
//                needs the wide evaluator or a long candidate search: run it
//                again now -- every earlier commit is visible, nobody else can
//                commit -- and decide on that (tallied as n_redo).
//
// Nothing is ever decided on stale data, so trajectories are those of the
// one-wave kernel and the oracle bit for bit, whatever W is.  W = 2 fills the
// chip at 4096 chains (8 waves per SIMD at <= 64 VGPRs); fewer chains take more
// waves each (W = 4, 8, 16), which is what keeps the GPU busy on the per-GPU
// shares of the 8-GPU configs (1024 and 256 chains).
//
// Memory ordering: head is read with acquire and written with release at
// workgroup scope; all waves of a workgroup run on one CU and share its vector
// L1, which is what workgroup scope means on gfx950.
#pragma once

#define MW_NONE 0xFFFFFFFFu
// log entry, u32 words: what a decision changed (flags = 0: nothing)
enum { ML_FLAGS = 0, ML_BIG1, ML_SMALL1, ML_ID1, ML_BIG2, ML_SMALL2, ML_ID2, ML_DSLOT, ML_WCLR, ML_WSET, ML_WORDS = 12 };
#define ML_ACCEPTED 1u
#define ML_DMOVE 2u

// LDS map in u64 words:
//   shared    cnt[16] | bmin[16] | bmax[16] | ctl[4] | log[W][6]
//   per wave  Hp[64] | arc list[64] | draw table: 32 entries of 14 u32 [224]
//   wide evaluator (one: only the token holder runs it)
#define MW_SHARED_WORDS 52u
#define MW_TBL_WORDS 14u
#define MW_WAVE_WORDS (128u + 16u * MW_TBL_WORDS)
__host__ __device__ inline unsigned fcm_mw_lds_words(int NW, int W)
{
    return MW_SHARED_WORDS + 6u * W + (unsigned)W * MW_WAVE_WORDS + fcm_lds_words(NW < 2 ? 2 : NW);
}

__device__ __forceinline__ void mw_barrier()   // orders LDS only
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ u32 mw_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

template <bool ROWS128>
__device__ __forceinline__ u64 mw_build(const rsrc_t rr, u32 stride32, u32 Lv, int s, int lane)
{
    if constexpr (ROWS128) return build_local_rows128(rr, Lv, s, lane);
    else return build_local_loop16(rr, stride32, Lv, s, lane);
}

// both endpoints of the pair (big, small) in the local list Lv?  (lanes beyond the list repeat its last vertex)
__device__ __forceinline__ bool mw_inside(u32 Lv, u32 big, u32 small) { return ballot(Lv == big) != 0ull && ballot(Lv == small) != 0ull; }

template <int MAXT, bool ROWS128>
__device__ __forceinline__ void mw_wave(const FcmStepParams &p, u64 *smem)
{
    const u32 W = p.mw_waves;                              // waves per chain: 2, 4, 8 or 16 (blockDim.x / 64)
    const int lane = threadIdx.x & (WAVE - 1);
    const u32 wv = mw_uni(threadIdx.x >> 6);
    const u32 chain = blockIdx.x;
    const int tmax = MAXT;
    const u32 N = (u32)p.nprop;                            // <= FCM_LAUNCH_CHUNK per launch
    if (N == 0) return;

    u64 *cntL = smem;
    u64 *bminL = smem + 16, *bmaxL = smem + 32;
    u32 *ctl = (u32 *)(smem + 48);                         // [0] head  [1] state inside the bounds?
    u32 *logL = (u32 *)(smem + MW_SHARED_WORDS);
    const int maxnw = p.maxnw < 2 ? 2 : p.maxnw;
    u64 *mine_lds = smem + MW_SHARED_WORDS + 6u * W + (size_t)wv * MW_WAVE_WORDS;
    u64 *Hp = mine_lds;                                    // split graph + arc list (eval_nodes / walk_nodes)
    u32 *T = (u32 *)(mine_lds + 128);                      // this wave's next 32 proposals
    u64 *wide_lds = smem + MW_SHARED_WORDS + 6u * W + (size_t)W * MW_WAVE_WORDS;

    u32 *rows = p.rows + (size_t)chain * p.rows_per_chain;
    const u64 rows_bytes = p.rows_per_chain * 4ull;
    u32 *dbl = p.dbl + (size_t)chain * p.dbl_stride;
    u64 *st_g = (u64 *)p.stats + (size_t)chain * FCM_DEV_NSTATS;
    const u32 *nb = p.nb;
    const FcmEdgeEntry *etab = p.etab;
    const u32 U = p.U, D = p.D, stride32 = p.stride32;
    const u64 Mtot = (u64)U + D;
    const u32 gchain = p.first_chain + chain;
    const u64 sampled0 = st_g[0];                          // Philox step index of proposal 0 of this launch

    if (wv == 0) {
        const int NC = p.ncounts;
        const bool cl = lane < NC;
        u64 *cnt_g = (u64 *)p.counts + (size_t)chain * FCM_DEV_MAX_COUNTS;
        const u64 c0 = cl ? cnt_g[lane] : 0ull;
        const u64 mn = cl ? p.bmin[lane] : 0ull, mx = cl ? p.bmax[lane] : ~0ull;   // zero-padded (src/util.rs:53-57)
        if (lane < 16) { cntL[lane] = c0; bminL[lane] = mn; bmaxL[lane] = mx; }
        const bool inb = ballot(cl && (c0 < mn || c0 > mx)) == 0ull;
        if (lane == 0) { ctl[0] = 0u; ctl[1] = inb ? 1u : 0u; }
    }
    mw_barrier();

    // this wave's share of the counters (added to the stats row at the end)
    u32 accepted = 0, n_empty = 0, n_flip = 0, n_dmove = 0, sum_k = 0, n_redo = 0, n_wide = 0, n_big = 0, mine = 0;
    u32 count_len = 0u, status = 0u;
    u32 ti = 32u;                                          // next table entry; 32 = refill

    for (u32 q = wv; q < N; q += W) {
        // ---- draw table: lane j draws this wave's j-th proposal from here ("Philox per lane"), with the static data it names
        if (ti >= 32u) {
            const int j = lane & 31;
            const u64 tj = sampled0 + (u64)q + (u64)j * W;
            const u32 k0 = (u32)p.seed, k1 = (u32)(p.seed >> 32);
            u32 w[4];
            philox4x32_10((u32)tj, (u32)(tj >> 32), gchain, 0u, k0, k1, w);
            const u64 cum0 = p.cum0, cum1 = p.cum1, cum2 = p.cum2;
            const int mv = ((u64)w[0] < cum0) ? 0 : (((u64)w[0] < cum1) ? 1 : (((u64)w[0] < cum2) ? 2 : 3));
            const u64 x64 = (u64)w[2] | ((u64)w[3] << 32);
            const u64 ix = mv >= 2 ? x64 : __umul64hi(x64, mv == 0 ? Mtot : (u64)D);
            FcmEdgeEntry e = {0u, 0u, 0u, 0u}, de = {0u, 0u, 0u, 0u};
            u32 ed = 0u, c0 = MW_NONE, c1 = MW_NONE;
            if (mv == 0 && ix < U) e = etab[ix];
            if (mv == 1 && D > 0) {
                ed = dbl[(u32)ix];           // a guess (the list is mutable): verified when the proposal runs
                de = etab[ed];
                u32 v[4];
                philox4x32_10((u32)tj, (u32)(tj >> 32), gchain, 1u, k0, k1, v);
                const u64 x0 = __umul64hi((u64)v[0] | ((u64)v[1] << 32), Mtot), x1 = __umul64hi((u64)v[2] | ((u64)v[3] << 32), Mtot);
                if (x0 < U) { c0 = (u32)x0; e = etab[x0]; }
                if (x1 < U) c1 = (u32)x1;
            }
            if (lane < 32) {
                u32 *t = T + j * MW_TBL_WORDS;
                t[0] = (u32)mv | ((w[1] & 1u) << 8); t[1] = ed; t[2] = (u32)ix; t[3] = (u32)(ix >> 32);
                t[4] = e.big; t[5] = e.small; t[6] = e.nb_off; t[7] = e.k;
                t[8] = de.big; t[9] = de.small; t[10] = de.nb_off; t[11] = de.k;
                t[12] = c0; t[13] = c1;
            }
            wave_sync();
            ti = 0u;
        }
        const u32 tv = lane < (int)MW_TBL_WORDS ? T[ti * MW_TBL_WORDS + lane] : 0u;
        ++ti;
        const int move = (int)(rdlane(tv, 0) & 0xFFu);
        const u32 coin = (rdlane(tv, 0) >> 8) & 1u;
        const u64 idx = (u64)rdlane(tv, 2) | ((u64)rdlane(tv, 3) << 32);
        if (move >= 2) status |= 4u;   // this kernel has no clique moves

        // ---- the proposal: first on the state as committed now, again under the token if that was not good enough
        bool exact = false;
        u32 nonempty, is_dmove, used_wide, big_set;   // 0 / 1 (wave-uniform integers: they end up in SGPRs, not in lane masks)
        u32 wid_clr, wid_set, bit_clr, bit_set, w_clr, w_set, dslot, dnew, add_k;
        u32 id1, big1, small1, id2, big2, small2, cx0, cx1, sus;
        u32 Lv1, Lv2;
        long long myd;
        for (;;) {
            nonempty = 0u; is_dmove = 0u; used_wide = 0u; big_set = 0u;
            wid_clr = wid_set = MW_NONE; bit_clr = bit_set = 0u; w_clr = w_set = 0u; dslot = MW_NONE; dnew = 0u; add_k = 0u;
            id1 = big1 = small1 = id2 = big2 = small2 = cx0 = cx1 = MW_NONE; sus = 0u;
            Lv1 = MW_NONE; Lv2 = MW_NONE;
            myd = 0;
            bool need_exact = false;
            // commits below snap are visible to every load from here on; those from snap on are checked under the token
            const u32 snap = exact ? q : mw_uni(__hip_atomic_load(&ctl[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
            const rsrc_t rr = make_rows_rsrc(rows, rows_bytes);

            // what is to be evaluated: up to two (masks, classes, size, sign) on the fast path
            int nev = 0;
            u64 HA = 0ull, HB = 0ull;
            Cls cA = {0ull, 0ull, 0ull}, cB = {0ull, 0ull, 0ull};
            int kA = 0, kB = 0;
            bool go_wide = false;
            // the pairs of the move: (big1, small1) with local set 1, (big2, small2) with local set 2
            FcmEdgeEntry e1 = {0u, 0u, 0u, 0u}, e2 = {0u, 0u, 0u, 0u};
            u32 rfwd = 0u;
            int fres = 0;          // flip: 1 = big->small flipped, 2 = small->big

            if (move == 0) {
                // ---- single_edge_flip (src/lib.rs:292-299)
                if (Mtot > 0 && idx < U) {
                    e1 = FcmEdgeEntry{rdlane(tv, 4), rdlane(tv, 5), rdlane(tv, 6), rdlane(tv, 7)};
                    const int k = (int)e1.k;
                    id1 = (u32)idx; big1 = e1.big; small1 = e1.small;
                    if (k + 2 <= WAVE) {
                        Lv1 = load_list(nb, e1.nb_off, k, e1.big, e1.small, lane);
                        const u64 myH = mw_build<ROWS128>(rr, stride32, Lv1, k + 2, lane);
                        const u64 hk = rdlane64(myH, k), hk1 = rdlane64(myH, k + 1);
                        const u32 ab = (u32)((hk1 >> k) & 1ull), ba = (u32)((hk >> (k + 1)) & 1ull);  // big->small, small->big
                        if (ab == ba) {
                            if (!ab) sus |= 1u;   // table says adjacent, bitmap says not
                        } else {
                            const int iu = ab ? k : k + 1, iv = ab ? k + 1 : k;
                            cA = classify(myH, iv, iu);
                            cB.P = cA.P; cB.S = cA.S;   // after the flip P and S are the same sets, M becomes {v->w, w->u}
                            cB.M = (ab ? hk : hk1) & ballot((myH >> iv) & 1ull) & ~(3ull << k);
                            fres = ab ? 1 : 2;
                            if (extras_fit(cA, k + 2) && extras_fit(cB, k + 2)) { HA = HB = myH; kA = kB = k; nev = 2; }
                            else go_wide = true;
                        }
                    } else {
                        go_wide = true;   // direction unknown yet: the wide run finds it
                    }
                    if (go_wide) {
                        if (!exact) {
                            need_exact = true;
                        } else if (k + 2 <= 64 * maxnw) {
                            const Wide Wd = wide_carve(wide_lds, maxnw);
                            wide_zero_counts(Wd, lane);
                            const int res = wide_flip(Wd, rows, stride32, nb, e1.nb_off, k, e1.big, e1.small, lane, tmax);
                            if (lane >= 2 && lane < 16 && lane - 1 <= tmax) myd = Wd.cnt[lane - 1];
                            wave_sync();
                            used_wide = 1u;
                            if (res < 0) sus |= 1u;
                            fres = res > 0 ? res : 0;
                        } else {
                            sus |= 1u;
                        }
                    }
                    if (fres > 0 && !need_exact) {
                        nonempty = 1u;
                        const u32 cf = fres == 1 ? e1.big : e1.small, ct = fres == 1 ? e1.small : e1.big;
                        wid_clr = cf * stride32 + (ct >> 5); bit_clr = 1u << (ct & 31u);
                        wid_set = ct * stride32 + (cf >> 5); bit_set = 1u << (cf & 31u);
                        add_k = (u32)k; big_set = k + 2 > 48 ? 1u : 0u;
                    }
                }
            } else if (move == 1 && D > 0) {
                // ---- double_edge_move (src/lib.rs:304-325)
                dslot = (u32)idx;
                const u32 ed = mw_uni(dbl[dslot]);                                             // the live entry ...
                e1 = FcmEdgeEntry{rdlane(tv, 8), rdlane(tv, 9), rdlane(tv, 10), rdlane(tv, 11)};   // ... and the table's guess of its pair
                if (ed != rdlane(tv, 1)) {
                    const FcmEdgeEntry t = etab[ed];
                    e1 = FcmEdgeEntry{mw_uni(t.big), mw_uni(t.small), mw_uni(t.nb_off), mw_uni(t.k)};
                }
                id1 = ed; big1 = e1.big; small1 = e1.small;
                // single-edge candidates (:308-313): candidates 0 and 1 come from the table; a longer search, or a
                // candidate that needs the wide path, is left to the exact run
                e2 = FcmEdgeEntry{rdlane(tv, 4), rdlane(tv, 5), rdlane(tv, 6), rdlane(tv, 7)};
                u64 cand = 0ull, cand_next = 0ull;
                bool found = false;
                const u64 tt = sampled0 + q;
#pragma nounroll
                for (int ci = 0; ci < WAVE && !found && !need_exact; ++ci) {
                    if (ci < 2) {
                        const u32 cv = rdlane(tv, 12 + ci);
                        cand = cv == MW_NONE ? ~0ull : (u64)cv;
                        if (ci == 0) cx0 = cv; else cx1 = cv;
                    } else if (!exact) {
                        need_exact = true;
                        break;
                    } else if ((ci & 1) == 0) {  // Philox block sub = ci/2 + 1: two candidates
                        u32 v[4];
                        philox4x32_10((u32)tt, (u32)(tt >> 32), gchain, (u32)(ci >> 1) + 1u, (u32)p.seed, (u32)(p.seed >> 32), v);
                        cand = __umul64hi((u64)v[0] | ((u64)v[1] << 32), Mtot);
                        cand_next = __umul64hi((u64)v[2] | ((u64)v[3] << 32), Mtot);
                    } else {
                        cand = cand_next;
                    }
                    if (cand < U) {
                        if (ci > 0) {
                            const FcmEdgeEntry t = etab[cand];
                            e2 = FcmEdgeEntry{mw_uni(t.big), mw_uni(t.small), mw_uni(t.nb_off), mw_uni(t.k)};
                        }
                        const int ck = (int)e2.k;
                        u32 f, bwd;
                        if (ck + 2 <= WAVE) {
                            Lv2 = load_list(nb, e2.nb_off, ck, e2.big, e2.small, lane);
                            HB = mw_build<ROWS128>(rr, stride32, Lv2, ck + 2, lane);
                            f = (u32)(rdlane64(HB, ck + 1) >> ck) & 1u;
                            bwd = (u32)(rdlane64(HB, ck) >> (ck + 1)) & 1u;
                        } else if (!exact) {
                            need_exact = true;
                            break;
                        } else {  // wide candidate: look at its two words directly
                            const u32 wf = mw_uni(rows[(size_t)e2.big * stride32 + (e2.small >> 5)]);
                            const u32 wb = mw_uni(rows[(size_t)e2.small * stride32 + (e2.big >> 5)]);
                            f = (wf >> (e2.small & 31u)) & 1u;
                            bwd = (wb >> (e2.big & 31u)) & 1u;
                            Lv2 = MW_NONE;
                        }
                        if (!(f | bwd)) sus |= 1u;
                        found = (f ^ bwd) != 0u;
                        rfwd = f;
                    }
                }
                if (found) {
                    const int dk = (int)e1.k, rk = (int)e2.k;
                    id2 = (u32)cand; big2 = e2.big; small2 = e2.small;
                    const u32 ea = rfwd ? e2.big : e2.small, eb = rfwd ? e2.small : e2.big;  // ea->eb is the single edge
                    const u32 dfrom = coin ? e1.big : e1.small, dto = coin ? e1.small : e1.big;  // delme (:316-320)
                    go_wide = dk + 2 > WAVE || rk + 2 > WAVE;
                    bool okd = true;
                    if (!go_wide) {
                        Lv1 = load_list(nb, e1.nb_off, dk, e1.big, e1.small, lane);
                        HA = mw_build<ROWS128>(rr, stride32, Lv1, dk + 2, lane);
                        // (1) remove the direction the coin picks from the reciprocal pair
                        const u32 ab = (u32)((rdlane64(HA, dk + 1) >> dk) & 1ull), ba = (u32)((rdlane64(HA, dk) >> (dk + 1)) & 1ull);
                        okd = (ab & ba) != 0u;
                        const int iu = coin ? dk : dk + 1, iv = coin ? dk + 1 : dk;
                        cA = classify(HA, iv, iu);
                        // (2) add the reverse of the single edge on the graph without the removed one
                        const u64 mf = ballot(lane < rk + 2 && Lv2 == dfrom), mt = ballot(lane < rk + 2 && Lv2 == dto);
                        if (mf && mt) {
                            const int fi = __ffsll((long long)mf) - 1, tix = __ffsll((long long)mt) - 1;
                            if (lane == tix) HB &= ~(1ull << fi);
                        }
                        const int ia = rfwd ? rk : rk + 1, ib = rfwd ? rk + 1 : rk;  // a->b exists, add b->a
                        if (lane == ia) HB |= 1ull << ib;
                        cB = classify(HB, ia, ib);
                        if (extras_fit(cA, dk + 2) && extras_fit(cB, rk + 2)) { kA = dk; kB = rk; nev = 2; }
                        else go_wide = true;
                    }
                    if (go_wide) {
                        if (!exact) {
                            need_exact = true;
                        } else if (dk + 2 > 64 * maxnw || rk + 2 > 64 * maxnw) {
                            sus |= 1u;
                        } else {
                            const Wide Wd = wide_carve(wide_lds, maxnw);
                            wide_zero_counts(Wd, lane);
                            okd = wide_del(Wd, rows, stride32, nb, e1.nb_off, dk, e1.big, e1.small, coin, lane, tmax);
                            wide_add(Wd, rows, stride32, nb, e2.nb_off, rk, e2.big, e2.small, rfwd, dfrom, dto, lane, tmax);
                            if (lane >= 2 && lane < 16 && lane - 1 <= tmax) myd = Wd.cnt[lane - 1];
                            wave_sync();
                            used_wide = 1u;
                        }
                    }
                    if (!okd) sus |= 2u;  // slot list says reciprocal, bitmap says not
                    nonempty = 1u; is_dmove = 1u;
                    wid_clr = dfrom * stride32 + (dto >> 5); bit_clr = 1u << (dto & 31u);
                    wid_set = eb * stride32 + (ea >> 5); bit_set = 1u << (ea & 31u);
                    dnew = (u32)cand;
                    add_k = (u32)(dk + rk); big_set = (dk + 2 > 48 || rk + 2 > 48) ? 1u : 0u;
                }
            }
            if (nonempty) {   // the two bitmap words a commit rewrites, read now: the commit is then two plain stores
                w_clr = rows[wid_clr];
                w_set = rows[wid_set];
            }
            if (nev) {
                int delta[MAXT + 1];
#pragma unroll
                for (int t = 0; t <= MAXT; ++t) delta[t] = 0;
#pragma nounroll
                for (int ev = 0; ev < 2; ++ev) {
                    const Cls c = ev ? cB : cA;
                    eval_nodes<MAXT>(ev ? HB : HA, Hp, c, ev ? kB : kA, tmax, ev ? +1 : -1, lane, delta);
                }
#pragma unroll
                for (int tq = 1; tq <= MAXT; ++tq) {
                    const int sum = wave_sum_i32(delta[tq]);
                    if (lane == tq + 1) myd = (long long)sum;
                }
            }
            if (exact) break;

            // ---- in-order decision.  While waiting for the token, hold the decisions taken since `snap` against this
            // proposal's reads as they are published: by the time head == q only the last of them is left to look at.
            bool hit = need_exact;
            u32 c = snap;
            for (;;) {
                const u32 h = mw_uni(__hip_atomic_load(&ctl[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));   // h <= q
                while (c < h && !hit) {
                    const u32 ev = lane < ML_WORDS ? logL[(c & (W - 1u)) * ML_WORDS + lane] : 0u;
                    ++c;
                    const u32 fl = rdlane(ev, ML_FLAGS);
                    if (!(fl & ML_ACCEPTED)) continue;
                    const u32 b1 = rdlane(ev, ML_BIG1), s1 = rdlane(ev, ML_SMALL1), i1 = rdlane(ev, ML_ID1);
                    const u32 wc = rdlane(ev, ML_WCLR), ws = rdlane(ev, ML_WSET);
                    hit = mw_inside(Lv1, b1, s1) || mw_inside(Lv2, b1, s1) || i1 == cx0 || i1 == cx1
                          || wc == wid_clr || wc == wid_set || ws == wid_clr || ws == wid_set;
                    if (fl & ML_DMOVE) {
                        const u32 b2 = rdlane(ev, ML_BIG2), s2 = rdlane(ev, ML_SMALL2), i2 = rdlane(ev, ML_ID2);
                        hit = hit || mw_inside(Lv1, b2, s2) || mw_inside(Lv2, b2, s2) || i2 == cx0 || i2 == cx1 || rdlane(ev, ML_DSLOT) == dslot;
                    }
                }
                if (h == q) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!hit) break;
            exact = true;   // under the token: every earlier commit is visible, nobody else can commit
            n_redo += 1;
        }

        // ---- Bounds::check; accept or drop (src/lib.rs:185-191) -- under the token, as little as possible
        const bool l16 = lane < 16;
        const u64 cnt = l16 ? cntL[lane] : 0ull, bmin = l16 ? bminL[lane] : 0ull, bmax = l16 ? bmaxL[lane] : ~0ull;
        const u32 in_bounds = mw_uni(ctl[1]);
        const u64 ncnt = cnt + (u64)myd;
        const u32 within = ballot(ncnt < bmin || ncnt > bmax) == 0ull ? 1u : 0u;
        const u32 commit = nonempty & within;
        if (commit) {
            if (l16) cntL[lane] = ncnt;
            if (lane == 0) {
                if (!in_bounds) ctl[1] = 1u;
                if (wid_clr == wid_set) {   // both changes in one word (double-edge move only)
                    rows[wid_clr] = (w_clr & ~bit_clr) | bit_set;
                } else {
                    rows[wid_clr] = w_clr & ~bit_clr;
                    rows[wid_set] = w_set | bit_set;
                }
                if (is_dmove) dbl[dslot] = dnew;
            }
        }
        if (lane < ML_WORDS) {
            u32 v = commit | (is_dmove << 1);
            v = lane == ML_BIG1 ? big1 : v; v = lane == ML_SMALL1 ? small1 : v; v = lane == ML_ID1 ? id1 : v;
            v = lane == ML_BIG2 ? big2 : v; v = lane == ML_SMALL2 ? small2 : v; v = lane == ML_ID2 ? id2 : v;
            v = lane == ML_DSLOT ? dslot : v; v = lane == ML_WCLR ? wid_clr : v; v = lane == ML_WSET ? wid_set : v;
            logL[(q & (W - 1u)) * ML_WORDS + lane] = v;
        }
        __hip_atomic_store(&ctl[0], q + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);

        // ---- the token is gone: this wave's counters (sampled += 1, src/lib.rs:185)
        status |= sus;
        mine += 1u;
        accepted += commit | ((nonempty ^ 1u) & in_bounds);   // an empty transition is accepted iff the state is inside the bounds
        n_empty += nonempty ^ 1u;
        n_dmove += nonempty & is_dmove;
        n_flip += nonempty & (is_dmove ^ 1u);
        sum_k += add_k;
        n_wide += used_wide;
        n_big += big_set;
        if (nonempty) {
            if (ballot(myd < 0 && cnt < (u64)(-myd))) status |= 8u;  // reference assert, src/lib.rs:65
            const u64 nz = ballot(l16 && ncnt != 0ull);              // flag_count never shrinks in length (src/lib.rs:72-74)
            const u32 nlen = nz ? (u32)(64 - __clzll((long long)nz)) : 0u;
            if (nlen > count_len) count_len = nlen;
        }
    }

    mw_barrier();   // every proposal decided
    if (wv == 0 && lane < p.ncounts) ((u64 *)p.counts + (size_t)chain * FCM_DEV_MAX_COUNTS)[lane] = cntL[lane];
    if (lane == 0) {
        atomicAdd((unsigned long long *)&st_g[0], (unsigned long long)mine);
        atomicAdd((unsigned long long *)&st_g[1], (unsigned long long)accepted);
        atomicAdd((unsigned long long *)&st_g[2], (unsigned long long)n_empty);
        atomicAdd((unsigned long long *)&st_g[3], (unsigned long long)n_flip);
        atomicAdd((unsigned long long *)&st_g[4], (unsigned long long)n_dmove);
        atomicAdd((unsigned long long *)&st_g[5], (unsigned long long)sum_k);
        atomicMax((unsigned long long *)&st_g[6], (unsigned long long)count_len);
        if (status) atomicOr((unsigned long long *)&st_g[7], (unsigned long long)status);
        if (n_redo) atomicAdd((unsigned long long *)&st_g[11], (unsigned long long)n_redo);
        if (n_wide) atomicAdd((unsigned long long *)&st_g[12], (unsigned long long)n_wide);
        if (n_big) atomicAdd((unsigned long long *)&st_g[13], (unsigned long long)n_big);
    }
}

// (W is a launch parameter: the block is W x 64 threads.  8 waves per SIMD whatever W is: at most 64 VGPRs.)
template <int MAXT, bool ROWS128>
__global__ __launch_bounds__(16 * WAVE, 8) void fcm_step_mw_kernel(const FcmStepParams p)
{
    extern __shared__ u64 smem[];
    if (blockIdx.x >= p.nchains) return;
    mw_wave<MAXT, ROWS128>(p, smem);
}
