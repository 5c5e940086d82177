// fcm_step_pc.hpp — the step kernel as a producer/consumer pair of waves per
// chain (simple moves, n <= 1024; included by fcm_step_variant.hip for the
// p<depth> tags).
//
// The one-wave kernel spends about half of each wave's time waiting on the two
// or three dependent memory round trips of a proposal (vertex list, bitmap
// rows) and the other half in the evaluations, with four waves per SIMD to
// overlap them.  Here the two halves are separate waves of one workgroup:
//
//   wave 0 (producer)  owns global memory: draws the proposals, loads the
//                      vertex lists and bitmap rows, builds the local in-masks
//                      of proposal q+1 into an LDS slot -- and writes the
//                      commit of proposal q-1 to the bitmap and the slot list;
//   wave 1 (consumer)  owns the chain state (counts, bounds, statistics): reads
//                      the masks of proposal q from the other LDS slot,
//                      evaluates, decides, and posts its decision in LDS.
//
// One workgroup barrier per proposal; neither wave waits for the other's
// memory traffic (the barrier orders LDS only).  Each needs fewer than 64
// VGPRs, so 8192 waves (4096 chains) are resident at 8 per SIMD.
//
// Staleness.  The slot of proposal q is built on the bitmap after the commits
// <= q-2; the consumer patches the commit of q-1 -- its own last decision --
// into the masks: the producer tells it where the endpoints of the pairs of
// proposal q-1 sit in each local set of q, the consumer knows what became of
// them.  The few decisions the producer takes from bits such a patch could
// change (which pair a slot of the reciprocal list names, whether a candidate
// pair is single) are never taken on stale data: when proposal q-1 could
// interfere -- same slot, a candidate on one of its pairs -- or when the local
// set needs the wide evaluator, the producer marks the proposal SERIAL, the
// consumer answers REDO, and the producer, which by then has written every
// commit, runs that proposal itself on the exact state (wide evaluator) while
// the consumer sits out one phase; the consumer then only does the bounds check.
#pragma once

#define PC_KIND_EMPTY 0u   // nothing to evaluate (the index names no single edge, D == 0, no candidate)
#define PC_KIND_FLIP 1u
#define PC_KIND_DMOVE 2u
#define PC_KIND_DELTA 3u   // the producer ran it itself on the exact state: count changes are in the slot
#define PC_KIND_SERIAL 4u  // cannot be prepared ahead: the consumer answers REDO at once

#define PC_DEC_NONE 0u     // no change to write (empty or rejected)
#define PC_DEC_COMMIT 1u
#define PC_DEC_REDO 2u

#define PC_NONE 0xFFFFFFFFu

// slot header, u32 words
enum {
    PH_KIND = 0, PH_BIG1, PH_SMALL1, PH_K1, PH_PAIR1,      // flip: its pair; double move: the reciprocal pair
    PH_BIG2, PH_SMALL2, PH_K2, PH_PAIR2,                   // double move: the single pair
    PH_FLAGS,                                              // coin | rfwd << 1
    PH_DSLOT,                                              // double move: index into the slot list
    PH_MEM11, PH_MEM12, PH_MEM21, PH_MEM22,                // where pair 1 / 2 of the PREVIOUS proposal sit in local set 1 / 2
    PH_PEND,                                               // where the pending removal (pair 1) sits in local set 2
    PH_PREV1, PH_PREV2,                                    // pair ids of the previous proposal the MEM words refer to
    // PC_KIND_DELTA: the proposal as the producer ran it
    PH_D_NONEMPTY, PH_D_ISDMOVE, PH_D_SUMK, PH_D_CLRF, PH_D_CLRT, PH_D_SETF, PH_D_SETT, PH_D_DSLOT, PH_D_DNEW,
    PH_D_BITS,                                             // new direction bits of pair 1 and pair 2: bs1 | sb1<<1 | bs2<<2 | sb2<<3
    PH_WORDS = 32
};
// decision, u32 words
enum { PD_STATUS = 0, PD_CLRF, PD_CLRT, PD_SETF, PD_SETT, PD_DSLOT, PD_DNEW, PD_WORDS = 8 };

// LDS map in u64 words: two slots {H1[64], H2[64], hdr[16]}, two decisions [2 x 4], Hp[64] + arc list [64], the producer's wide evaluator
#define PC_SLOT_WORDS (64u + 64u + 16u)
#define PC_FIXED_WORDS (2u * PC_SLOT_WORDS + 8u + 128u)
__host__ __device__ inline unsigned fcm_pc_lds_words(int NW) { return PC_FIXED_WORDS + fcm_lds_words(NW < 2 ? 2 : NW); }

// workgroup barrier that orders LDS only: the producer's global stores and loads are its own business
__device__ __forceinline__ void pc_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ u32 pc_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

// where the vertices (big, small) sit in the local list Lv (s entries): ibig | ismall << 8 | 1 << 16, or 0
__device__ __forceinline__ u32 pc_member(u32 Lv, int s, int lane, u32 big, u32 small)
{
    if (big == PC_NONE) return 0u;   // (wave-uniform)
    const u64 mb = ballot(lane < s && Lv == big), ms = ballot(lane < s && Lv == small);
    if (!mb || !ms) return 0u;
    return (u32)(__ffsll((long long)mb) - 1) | ((u32)(__ffsll((long long)ms) - 1) << 8) | (1u << 16);
}
// in-masks: x -> y present <=> bit ix of lane iy.  The pair at (ibig, ismall) now has big->small = bs, small->big = sb.
__device__ __forceinline__ u64 pc_patch(u64 myH, u32 mem, u32 bs, u32 sb, int lane)
{
    if (mem >> 16) {
        const int ibig = (int)(mem & 0xFFu), ismall = (int)((mem >> 8) & 0xFFu);
        if (lane == ismall) myH = (myH & ~(1ull << ibig)) | ((u64)bs << ibig);
        if (lane == ibig) myH = (myH & ~(1ull << ismall)) | ((u64)sb << ismall);
    }
    return myH;
}

struct PcCand { u64 id; FcmEdgeEntry e; u32 Lv; u64 H; u32 fwd; bool found, serial; };

// the producer's view of a chain (wave-uniform values)
struct PcChain {
    const FcmEdgeEntry *etab;
    const u32 *nb;
    u32 *rows;
    u32 *dbl;
    u64 rows_bytes, Mtot;
    u32 U, D, stride32, k0, k1, gchain;
};

// ROWS128: rows of one cache line (n <= 1024) are read whole, two per load; longer rows one dword per
// lane and row, 16 at a time
template <bool ROWS128>
__device__ __forceinline__ u64 pc_build(const u32 *rows, u64 rows_bytes, u32 stride32, u32 Lv, int s, int lane)
{
    if constexpr (ROWS128) return build_local_rows128(make_rows_rsrc(rows, rows_bytes), Lv, s, lane);
    else return build_local_loop16(make_rows_rsrc(rows, rows_bytes), stride32, Lv, s, lane);
}

// Single-edge candidate search of a double move (src/lib.rs:308-313) on the bitmap as it is now.
// exact == false: a candidate on one of the pairs (pv1, pv2) of the previous proposal, or one that
// needs the wide path, stops the search (serial).  have01: candidates 0 and 1 (and the entry of 0)
// come from the producer's table.  The winner's list and masks are returned when its set is narrow.
template <bool ROWS128>
__device__ __forceinline__ PcCand pc_find_candidate(const PcChain &C, u64 tt, bool exact, u32 pv1, u32 pv2, bool have01, u32 c0, u32 c1,
                                                    FcmEdgeEntry e0, int lane, u32 &status)
{
    PcCand c;
    c.id = 0ull; c.e = FcmEdgeEntry{0u, 0u, 0u, 0u}; c.Lv = 0u; c.H = 0ull; c.fwd = 0u;
    c.found = false; c.serial = false;
    u64 cand = 0ull, cand_next = 0ull;
#pragma nounroll
    for (int ci = 0; ci < WAVE && !c.found && !c.serial; ++ci) {
        if (have01 && ci < 2) {   // PC_NONE: not a pair index
            cand = ci == 0 ? (c0 == PC_NONE ? ~0ull : (u64)c0) : (c1 == PC_NONE ? ~0ull : (u64)c1);
        } else if ((ci & 1) == 0) {  // Philox block sub = ci/2 + 1: two candidates
            u32 v[4];
            philox4x32_10((u32)tt, (u32)(tt >> 32), C.gchain, (u32)(ci >> 1) + 1u, C.k0, C.k1, v);
            cand = __umul64hi((u64)v[0] | ((u64)v[1] << 32), C.Mtot);
            cand_next = __umul64hi((u64)v[2] | ((u64)v[3] << 32), C.Mtot);
        } else {
            cand = cand_next;
        }
        if (cand < C.U) {
            if (!exact && ((u32)cand == pv1 || (u32)cand == pv2)) { c.serial = true; break; }
            const FcmEdgeEntry ce = (have01 && ci == 0) ? e0 : C.etab[cand];
            const int ck = (int)ce.k;
            u32 f, bwd;
            if (ck + 2 <= WAVE) {
                c.Lv = load_list(C.nb, ce.nb_off, ck, ce.big, ce.small, lane);
                c.H = pc_build<ROWS128>(C.rows, C.rows_bytes, C.stride32, c.Lv, ck + 2, lane);
                f = (u32)(rdlane64(c.H, ck + 1) >> ck) & 1u;
                bwd = (u32)(rdlane64(c.H, ck) >> (ck + 1)) & 1u;
            } else {
                if (!exact) { c.serial = true; break; }
                const u32 wf = C.rows[(size_t)ce.big * C.stride32 + (ce.small >> 5)];
                const u32 wb = C.rows[(size_t)ce.small * C.stride32 + (ce.big >> 5)];
                f = (wf >> (ce.small & 31u)) & 1u;
                bwd = (wb >> (ce.big & 31u)) & 1u;
            }
            if (!(f | bwd)) status |= 1u;
            c.found = (f ^ bwd) != 0u;
            c.fwd = f;
            c.id = cand;
            c.e = ce;
        }
    }
    return c;
}

// Run one proposal on the exact state with the wide evaluator and leave the count changes, the
// change list and the pairs in the slot (kind DELTA).  A few per ten thousand proposals.  Returns
// status bits.  (Inlined: as a real call it costs the producer's loop more -- stack traffic at
// every use of the chain descriptor -- than its registers do.)
template <bool ROWS128>
__device__ __forceinline__ u32 pc_run_exact(const PcChain &C, u64 *wsm, int maxnw, int tmax, u64 *H1, u32 *hdr, int move, u32 coin,
                                                      u64 idx, u64 tt, int lane)
{
    u32 status = 0u;
    const Wide W = wide_carve(wsm, maxnw);
    wide_zero_counts(W, lane);
    u32 nonempty = 0u, isd = 0u, sumk = 0u, clrf = 0u, clrt = 0u, setf = 0u, sett = 0u, dslot = PC_NONE, dnew = 0u, bits = 0u;
    u32 pair1 = PC_NONE, big1 = PC_NONE, small1 = PC_NONE, pair2 = PC_NONE, big2 = PC_NONE, small2 = PC_NONE;
    if (move == 0) {
        if (C.Mtot > 0 && idx < C.U) {
            const FcmEdgeEntry e = C.etab[idx];
            int res = -1;
            if ((int)e.k + 2 <= 64 * maxnw) res = wide_flip(W, C.rows, C.stride32, C.nb, e.nb_off, (int)e.k, e.big, e.small, lane, tmax);
            if (res < 0) status |= 1u;
            if (res > 0) {
                nonempty = 1u; sumk = e.k;
                clrf = res == 1 ? e.big : e.small; clrt = res == 1 ? e.small : e.big;
                setf = clrt; sett = clrf;
                pair1 = (u32)idx; big1 = e.big; small1 = e.small;
                bits = res == 1 ? 2u : 1u;   // big->small went (bs = 0, sb = 1) or the other way round
            }
        }
    } else if (move == 1 && C.D > 0) {
        dslot = (u32)idx;
        const u32 ed = C.dbl[dslot];
        const FcmEdgeEntry de = C.etab[ed];
        const PcCand c = pc_find_candidate<ROWS128>(C, tt, true, PC_NONE, PC_NONE, false, PC_NONE, PC_NONE, FcmEdgeEntry{0u, 0u, 0u, 0u}, lane, status);
        if (c.found) {
            nonempty = 1u; isd = 1u;
            const u32 ea = c.fwd ? c.e.big : c.e.small, eb = c.fwd ? c.e.small : c.e.big;  // ea->eb is the single edge
            const u32 dfrom = coin ? de.big : de.small, dto = coin ? de.small : de.big;
            if ((int)de.k + 2 > 64 * maxnw || (int)c.e.k + 2 > 64 * maxnw) {
                status |= 1u;
            } else {
                if (!wide_del(W, C.rows, C.stride32, C.nb, de.nb_off, (int)de.k, de.big, de.small, coin, lane, tmax)) status |= 2u;
                wide_add(W, C.rows, C.stride32, C.nb, c.e.nb_off, (int)c.e.k, c.e.big, c.e.small, c.fwd, dfrom, dto, lane, tmax);
            }
            clrf = dfrom; clrt = dto; setf = eb; sett = ea;
            dnew = (u32)c.id;
            sumk = de.k + c.e.k;
            pair1 = ed; big1 = de.big; small1 = de.small;
            pair2 = (u32)c.id; big2 = c.e.big; small2 = c.e.small;
            bits = (coin ? 2u : 1u) | (3u << 2);   // the direction the coin picked went; pair 2 is reciprocal now
        } else {
            dslot = PC_NONE;
        }
    }
    // count changes for t = 1..tmax: lane t+1 holds dimension t+1, as in the consumer
    if (lane >= 2 && lane < 16 && lane - 1 <= tmax) H1[lane] = (u64)W.cnt[lane - 1];
    if (lane == 0) {
        hdr[PH_KIND] = PC_KIND_DELTA;
        hdr[PH_D_NONEMPTY] = nonempty; hdr[PH_D_ISDMOVE] = isd; hdr[PH_D_SUMK] = sumk;
        hdr[PH_D_CLRF] = clrf; hdr[PH_D_CLRT] = clrt; hdr[PH_D_SETF] = setf; hdr[PH_D_SETT] = sett;
        hdr[PH_D_DSLOT] = isd ? dslot : PC_NONE; hdr[PH_D_DNEW] = dnew; hdr[PH_D_BITS] = bits;
        hdr[PH_PAIR1] = pair1; hdr[PH_PAIR2] = pair2;
        hdr[PH_BIG1] = big1; hdr[PH_SMALL1] = small1; hdr[PH_BIG2] = big2; hdr[PH_SMALL2] = small2;
    }
    wave_sync();
    return status;
}

template <int MAXT, bool ROWS128>
__device__ __forceinline__ void pc_producer(const FcmStepParams &p, u64 *smem)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const u32 chain = blockIdx.x;
    auto slotH1 = [&](u64 q) -> u64 * { return smem + (q & 1ull) * PC_SLOT_WORDS; };
    auto slotH2 = [&](u64 q) -> u64 * { return smem + (q & 1ull) * PC_SLOT_WORDS + 64; };
    auto slotHdr = [&](u64 q) -> u32 * { return (u32 *)(smem + (q & 1ull) * PC_SLOT_WORDS + 128); };
    auto decOf = [&](u64 q) -> u32 * { return (u32 *)(smem + 2 * PC_SLOT_WORDS) + (q & 1ull) * PD_WORDS; };
    u64 *wsm = smem + PC_FIXED_WORDS;         // wide evaluator / the table

    const u64 N = p.nprop;
    if (N == 0) return;
    const int tmax = MAXT;
    u64 *st_g = (u64 *)p.stats + (size_t)chain * FCM_DEV_NSTATS;
    PcChain C;
    C.etab = p.etab; C.nb = p.nb;
    C.rows = p.rows + (size_t)chain * p.rows_per_chain;
    C.dbl = p.dbl + (size_t)chain * p.dbl_stride;
    C.rows_bytes = p.rows_per_chain * 4ull;
    C.U = p.U; C.D = p.D; C.Mtot = (u64)p.U + p.D; C.stride32 = p.stride32;
    C.k0 = (u32)p.seed; C.k1 = (u32)(p.seed >> 32); C.gchain = p.first_chain + chain;
    const u32 U = C.U, D = C.D;
    const u64 Mtot = C.Mtot;
    const u64 cum0 = p.cum0, cum1 = p.cum1, cum2 = p.cum2;
    const u64 sampled0 = st_g[0];
    const int maxnw = p.maxnw < 2 ? 2 : p.maxnw;
    u32 pstatus = 0u;

    // the proposal prepared (or run) last: its pairs, its slot if it is a double move
    u32 pv_pair1 = PC_NONE, pv_big1 = PC_NONE, pv_small1 = PC_NONE, pv_pair2 = PC_NONE, pv_big2 = PC_NONE, pv_small2 = PC_NONE;
    u32 pv_dslot = PC_NONE;
    // Table of the next 32 proposals (in the LDS region of the wide evaluator, which wipes it:
    // table_dirty), 16 words each: the draw (move, coin word, index), the pair entry of a flip or of
    // a double move's first candidate, the entry of the pair its slot names now (a guess, verified
    // when the proposal is prepared), candidates 0 and 1.  Lane j fills entry j: "Philox per lane".
    u32 *T = (u32 *)wsm;
    bool table_dirty = true;

    enum { S_PRO, S_NORMAL, S_EXACT, S_AFTER };
    int st = S_PRO;
    u64 cq = 0;
    for (;;) {
        const bool do_apply = st == S_NORMAL && cq > 0;
        const bool do_exact = st == S_EXACT;
        const bool do_prep = st == S_PRO || ((st == S_NORMAL || st == S_AFTER) && cq + 1 < N);
        const u64 q = st == S_PRO ? 0ull : cq + 1;   // the proposal to prepare

        // ---- write the commit of proposal cq-1 to the bitmap and the slot list
        if (do_apply) {
            const u32 *d = decOf(cq - 1);
            const u32 dv = lane < PD_WORDS ? d[lane] : 0u;
            if (rdlane(dv, PD_STATUS) == PC_DEC_COMMIT && lane == 0) {
                const u32 cf = rdlane(dv, PD_CLRF), ct = rdlane(dv, PD_CLRT), sf = rdlane(dv, PD_SETF), stt = rdlane(dv, PD_SETT);
                u32 *pc = C.rows + (size_t)cf * C.stride32 + (ct >> 5);
                u32 *ps = C.rows + (size_t)sf * C.stride32 + (stt >> 5);
                const u32 bc = 1u << (ct & 31u), bs = 1u << (stt & 31u);
                if (pc == ps) {
                    *pc = (*pc & ~bc) | bs;
                } else {
                    const u32 vc = *pc, vs = *ps;
                    *pc = vc & ~bc;
                    *ps = vs | bs;
                }
                if (rdlane(dv, PD_DSLOT) != PC_NONE) C.dbl[rdlane(dv, PD_DSLOT)] = rdlane(dv, PD_DNEW);
            }
        }

        // ---- run proposal cq on the exact state (after a REDO; the consumer idles)
        if (do_exact) {
            int x_move;
            u32 x_coin;
            u64 x_idx;
            {   // the draw of proposal cq again (its table entry may be gone)
                const u64 t = sampled0 + cq;
                u32 w[4];
                philox4x32_10((u32)t, (u32)(t >> 32), C.gchain, 0u, C.k0, C.k1, w);
                x_move = ((u64)w[0] < cum0) ? 0 : (((u64)w[0] < cum1) ? 1 : (((u64)w[0] < cum2) ? 2 : 3));
                x_coin = w[1] & 1u;
                const u64 x64 = (u64)w[2] | ((u64)w[3] << 32);
                x_idx = x_move >= 2 ? x64 : __umul64hi(x64, x_move == 0 ? Mtot : (u64)D);
            }
            u32 *hdr = slotHdr(cq);
            pstatus |= pc_run_exact<ROWS128>(C, wsm, maxnw, tmax, slotH1(cq), hdr, x_move, x_coin, x_idx, sampled0 + cq, lane);
            table_dirty = true;   // the wide evaluator's LDS region holds the table
            const u32 hv = lane < PH_WORDS ? hdr[lane] : 0u;
            pv_pair1 = rdlane(hv, PH_PAIR1); pv_big1 = rdlane(hv, PH_BIG1); pv_small1 = rdlane(hv, PH_SMALL1);
            pv_pair2 = rdlane(hv, PH_PAIR2); pv_big2 = rdlane(hv, PH_BIG2); pv_small2 = rdlane(hv, PH_SMALL2);
            pv_dslot = rdlane(hv, PH_D_DSLOT);
        }

        // ---- prepare proposal q (ahead of time) into its slot
        if (do_prep) {
            if ((q & 31ull) == 0 || table_dirty) {
                const u64 qbase = q & ~31ull;
                const int j = lane & 31;
                const u64 tj = sampled0 + qbase + (u64)j;
                u32 w[4];
                philox4x32_10((u32)tj, (u32)(tj >> 32), C.gchain, 0u, C.k0, C.k1, w);
                const int mv = ((u64)w[0] < cum0) ? 0 : (((u64)w[0] < cum1) ? 1 : (((u64)w[0] < cum2) ? 2 : 3));
                const u64 x64 = (u64)w[2] | ((u64)w[3] << 32);
                const u64 ix = mv >= 2 ? x64 : __umul64hi(x64, mv == 0 ? Mtot : (u64)D);
                FcmEdgeEntry e = {0u, 0u, 0u, 0u}, de = {0u, 0u, 0u, 0u};
                u32 ed = 0u, c0 = PC_NONE, c1 = PC_NONE;
                if (mv == 0 && ix < U) e = C.etab[ix];
                if (mv == 1 && D > 0) {
                    ed = C.dbl[(u32)ix];
                    de = C.etab[ed];
                    u32 v[4];
                    philox4x32_10((u32)tj, (u32)(tj >> 32), C.gchain, 1u, C.k0, C.k1, v);
                    const u64 x0 = __umul64hi((u64)v[0] | ((u64)v[1] << 32), Mtot), x1 = __umul64hi((u64)v[2] | ((u64)v[3] << 32), Mtot);
                    if (x0 < U) { c0 = (u32)x0; e = C.etab[x0]; }
                    if (x1 < U) c1 = (u32)x1;
                }
                if (lane < 32) {
                    u32 *t = T + j * 16;
                    t[0] = (u32)mv; t[1] = w[1]; t[2] = (u32)ix; t[3] = (u32)(ix >> 32);
                    t[4] = e.big; t[5] = e.small; t[6] = e.nb_off; t[7] = e.k;
                    t[8] = de.big; t[9] = de.small; t[10] = de.nb_off; t[11] = de.k;
                    t[12] = ed; t[13] = c0; t[14] = c1;
                }
                wave_sync();
                table_dirty = false;
            }
            const u32 tv = lane < 16 ? T[(q & 31ull) * 16 + lane] : 0u;   // this proposal's table entry in one read
            u64 *H1 = slotH1(q), *H2 = slotH2(q);
            u32 *hdr = slotHdr(q);
            const int move = (int)rdlane(tv, 0);
            const u32 coin = rdlane(tv, 1) & 1u;
            const u64 idx = (u64)rdlane(tv, 2) | ((u64)rdlane(tv, 3) << 32);
            u32 kind = PC_KIND_EMPTY;
            u32 big1 = PC_NONE, small1 = PC_NONE, pair1 = PC_NONE, big2 = PC_NONE, small2 = PC_NONE, pair2 = PC_NONE;
            u32 kk1 = 0u, kk2 = 0u, flags = 0u, dslot = PC_NONE, m11 = 0u, m12 = 0u, m21 = 0u, m22 = 0u, pend = 0u;
            if (move == 0) {
                if (Mtot > 0 && idx < U) {
                    const FcmEdgeEntry e = {rdlane(tv, 4), rdlane(tv, 5), rdlane(tv, 6), rdlane(tv, 7)};
                    big1 = e.big; small1 = e.small; pair1 = (u32)idx; kk1 = e.k;
                    if ((int)e.k + 2 <= WAVE) {
                        const u32 Lv = load_list(C.nb, e.nb_off, (int)e.k, e.big, e.small, lane);
                        const u64 myH = pc_build<ROWS128>(C.rows, C.rows_bytes, C.stride32, Lv, (int)e.k + 2, lane);
                        m11 = pc_member(Lv, (int)e.k + 2, lane, pv_big1, pv_small1);
                        m12 = pc_member(Lv, (int)e.k + 2, lane, pv_big2, pv_small2);
                        H1[lane] = myH;
                        kind = PC_KIND_FLIP;
                    } else {
                        kind = PC_KIND_SERIAL;
                    }
                }
            } else if (move == 1) {
                if (D > 0) {
                    dslot = (u32)idx;
                    flags = coin;
                    if (dslot == pv_dslot) {
                        kind = PC_KIND_SERIAL;  // the previous double move may rewrite this very slot
                    } else {
                        const u32 ed = C.dbl[dslot];                       // the live entry ...
                        FcmEdgeEntry de = {rdlane(tv, 8), rdlane(tv, 9), rdlane(tv, 10), rdlane(tv, 11)};  // ... and the table's guess of its pair
                        const FcmEdgeEntry e0 = {rdlane(tv, 4), rdlane(tv, 5), rdlane(tv, 6), rdlane(tv, 7)};
                        const PcCand c = pc_find_candidate<ROWS128>(C, sampled0 + q, false, pv_pair1, pv_pair2, true, rdlane(tv, 13), rdlane(tv, 14), e0,
                                                           lane, pstatus);
                        if (ed != rdlane(tv, 12)) de = C.etab[ed];         // the slot was rewritten since the table was filled
                        big1 = de.big; small1 = de.small; pair1 = ed; kk1 = de.k;
                        if (c.serial || (int)de.k + 2 > WAVE) {
                            kind = PC_KIND_SERIAL;
                        } else if (c.found) {
                            big2 = c.e.big; small2 = c.e.small; pair2 = (u32)c.id; kk2 = c.e.k;
                            flags |= c.fwd << 1;
                            const int s2 = (int)c.e.k + 2, s1 = (int)de.k + 2;
                            const u32 Lv1 = load_list(C.nb, de.nb_off, (int)de.k, de.big, de.small, lane);
                            const u64 myH1 = pc_build<ROWS128>(C.rows, C.rows_bytes, C.stride32, Lv1, s1, lane);
                            m11 = pc_member(Lv1, s1, lane, pv_big1, pv_small1);
                            m12 = pc_member(Lv1, s1, lane, pv_big2, pv_small2);
                            m21 = pc_member(c.Lv, s2, lane, pv_big1, pv_small1);
                            m22 = pc_member(c.Lv, s2, lane, pv_big2, pv_small2);
                            pend = pc_member(c.Lv, s2, lane, de.big, de.small);
                            H1[lane] = myH1;
                            H2[lane] = c.H;
                            kind = PC_KIND_DMOVE;
                        }
                    }
                }
            } else {
                pstatus |= 4u;  // this kernel has no clique moves
            }
            if (lane == 0) {
                hdr[PH_KIND] = kind;
                hdr[PH_BIG1] = big1; hdr[PH_SMALL1] = small1; hdr[PH_K1] = kk1; hdr[PH_PAIR1] = pair1;
                hdr[PH_BIG2] = big2; hdr[PH_SMALL2] = small2; hdr[PH_K2] = kk2; hdr[PH_PAIR2] = pair2;
                hdr[PH_FLAGS] = flags; hdr[PH_DSLOT] = dslot;
                hdr[PH_MEM11] = m11; hdr[PH_MEM12] = m12; hdr[PH_MEM21] = m21; hdr[PH_MEM22] = m22; hdr[PH_PEND] = pend;
                hdr[PH_PREV1] = pv_pair1; hdr[PH_PREV2] = pv_pair2;
            }
            const bool known = kind == PC_KIND_FLIP || kind == PC_KIND_DMOVE;
            pv_pair1 = known ? pair1 : PC_NONE; pv_big1 = known ? big1 : PC_NONE; pv_small1 = known ? small1 : PC_NONE;
            pv_pair2 = kind == PC_KIND_DMOVE ? pair2 : PC_NONE; pv_big2 = kind == PC_KIND_DMOVE ? big2 : PC_NONE;
            pv_small2 = kind == PC_KIND_DMOVE ? small2 : PC_NONE;
            pv_dslot = kind == PC_KIND_DMOVE ? dslot : PC_NONE;
        }

        pc_barrier();
        // ---- next phase
        if (st == S_PRO) {
            st = S_NORMAL;
        } else if (st == S_NORMAL) {
            if (__builtin_expect(pc_uni(decOf(cq)[PD_STATUS]) == PC_DEC_REDO, 0)) st = S_EXACT;
            else if (++cq == N) break;
        } else if (st == S_EXACT) {
            st = S_AFTER;
        } else {
            st = S_NORMAL;
            if (++cq == N) break;
        }
    }
    // the last commit
    {
        const u32 *d = decOf(N - 1);
        const u32 dv = lane < PD_WORDS ? d[lane] : 0u;
        if (rdlane(dv, PD_STATUS) == PC_DEC_COMMIT && lane == 0) {
            const u32 cf = rdlane(dv, PD_CLRF), ct = rdlane(dv, PD_CLRT), sf = rdlane(dv, PD_SETF), stt = rdlane(dv, PD_SETT);
            u32 *pc = C.rows + (size_t)cf * C.stride32 + (ct >> 5);
            u32 *ps = C.rows + (size_t)sf * C.stride32 + (stt >> 5);
            const u32 bc = 1u << (ct & 31u), bs = 1u << (stt & 31u);
            if (pc == ps) {
                *pc = (*pc & ~bc) | bs;
            } else {
                const u32 vc = *pc, vs = *ps;
                *pc = vc & ~bc;
                *ps = vs | bs;
            }
            if (rdlane(dv, PD_DSLOT) != PC_NONE) C.dbl[rdlane(dv, PD_DSLOT)] = rdlane(dv, PD_DNEW);
        }
    }
    if (lane == 0 && pstatus) atomicOr((unsigned long long *)&st_g[7], (unsigned long long)pstatus);
}

template <int MAXT>
__device__ __forceinline__ void pc_consumer(const FcmStepParams &p, u64 *smem)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const u32 chain = blockIdx.x;

    auto slotH1 = [&](u64 q) -> u64 * { return smem + (q & 1ull) * PC_SLOT_WORDS; };
    auto slotH2 = [&](u64 q) -> u64 * { return smem + (q & 1ull) * PC_SLOT_WORDS + 64; };
    auto slotHdr = [&](u64 q) -> u32 * { return (u32 *)(smem + (q & 1ull) * PC_SLOT_WORDS + 128); };
    auto decOf = [&](u64 q) -> u32 * { return (u32 *)(smem + 2 * PC_SLOT_WORDS) + (q & 1ull) * PD_WORDS; };
    u64 *Hp = smem + 2 * PC_SLOT_WORDS + 8;   // + arc list behind it (consumer)
    u64 *wsm = smem + PC_FIXED_WORDS;         // wide evaluator (producer)

    const u64 N = p.nprop;
    const int tmax = MAXT;
    const u32 U = p.U, D = p.D;
    const u64 Mtot = (u64)U + D;
    const u32 stride32 = p.stride32;
    u64 *st_g = (u64 *)p.stats + (size_t)chain * FCM_DEV_NSTATS;

    {
        // ================================ consumer ================================
        if (N == 0) return;
        u64 *cnt_g = (u64 *)p.counts + (size_t)chain * FCM_DEV_MAX_COUNTS;
        const int NC = p.ncounts;
        const bool cl = lane < NC;
        u64 cnt = cl ? cnt_g[lane] : 0ull;
        const u64 bmin = cl ? p.bmin[lane] : 0ull;
        const u64 bmax = cl ? p.bmax[lane] : ~0ull;
        u64 sampled = st_g[0], accepted = st_g[1], n_empty = st_g[2], n_flip = st_g[3], n_dmove = st_g[4], sum_k = st_g[5];
        u32 count_len = (u32)st_g[6];
        u64 n_redo = st_g[11], n_wide = st_g[12], n_big = st_g[13];   // diagnostics
        u32 status = 0u;
        bool in_bounds = ballot(cl && (cnt < bmin || cnt > bmax)) == 0ull;
        // the last commit: its pairs and what became of their two directions
        bool last_commit = false;
        u32 lc_pair1 = PC_NONE, lc_pair2 = PC_NONE, lc_bits = 0u;

        pc_barrier();
        u64 cq = 0;
        while (cq < N) {
            const u64 *H1 = slotH1(cq), *H2 = slotH2(cq);
            const u32 *hdr = slotHdr(cq);
            u32 *d = decOf(cq);
            const u32 hv = lane < PH_WORDS ? hdr[lane] : 0u;   // the whole header in one LDS read; fields by v_readlane
            const u64 slot_h1 = H1[lane], slot_h2 = H2[lane];   // (in flight with the header: one LDS round trip, not two)
            const u32 kind = rdlane(hv, PH_KIND);
            bool nonempty = false, is_dmove = false, redo = false;
            u32 clr_from = 0u, clr_to = 0u, set_from = 0u, set_to = 0u, dslot = PC_NONE, dnew = 0u;
            u32 my_pair1 = PC_NONE, my_pair2 = PC_NONE, my_bits = 0u, add_k = 0u;
            long long myd = 0;
            int delta[MAXT + 1];
#pragma unroll
            for (int q = 0; q <= MAXT; ++q) delta[q] = 0;
            bool have_delta = false;

            // the masks of a stale slot get the last commit patched in
            const bool patch = last_commit && (kind == PC_KIND_FLIP || kind == PC_KIND_DMOVE);
            if (patch && (rdlane(hv, PH_PREV1) != lc_pair1 || rdlane(hv, PH_PREV2) != lc_pair2)) status |= 32u;  // (cannot happen)

            if (kind == PC_KIND_SERIAL) {
                redo = true;
            } else if (kind == PC_KIND_FLIP) {
                const u32 a = rdlane(hv, PH_BIG1), b = rdlane(hv, PH_SMALL1);
                const int k = (int)rdlane(hv, PH_K1);
                u64 myH = slot_h1;
                if (patch) {
                    myH = pc_patch(myH, rdlane(hv, PH_MEM11), lc_bits & 1u, (lc_bits >> 1) & 1u, lane);
                    myH = pc_patch(myH, rdlane(hv, PH_MEM12), (lc_bits >> 2) & 1u, (lc_bits >> 3) & 1u, lane);
                }
                const u64 hk = rdlane64(myH, k), hk1 = rdlane64(myH, k + 1);
                const u32 ab = (u32)((hk1 >> k) & 1ull), ba = (u32)((hk >> (k + 1)) & 1ull);  // big->small, small->big
                if (ab == ba) {
                    if (!ab) status |= 1u;
                } else {
                    const int iu = ab ? k : k + 1, iv = ab ? k + 1 : k;
                    Cls c = classify(myH, iv, iu);
                    Cls c2;
                    c2.P = c.P; c2.S = c.S;
                    c2.M = (ab ? hk : hk1) & ballot((myH >> iv) & 1ull) & ~(3ull << k);
                    if (!extras_fit(c, k + 2) || !extras_fit(c2, k + 2)) {
                        redo = true;
                    } else {
                        eval_nodes<MAXT>(myH, Hp, c, k, tmax, -1, lane, delta);
                        eval_nodes<MAXT>(myH, Hp, c2, k, tmax, +1, lane, delta);
                        nonempty = true; have_delta = true;
                        clr_from = ab ? a : b; clr_to = ab ? b : a;
                        set_from = clr_to; set_to = clr_from;
                        add_k = (u32)k;
                        if (k + 2 > 48) n_big += 1;
                        my_pair1 = rdlane(hv, PH_PAIR1);
                        my_bits = ab ? 2u : 1u;
                    }
                }
            } else if (kind == PC_KIND_DMOVE) {
                const u32 big1 = rdlane(hv, PH_BIG1), small1 = rdlane(hv, PH_SMALL1);
                const u32 big2 = rdlane(hv, PH_BIG2), small2 = rdlane(hv, PH_SMALL2);
                const int dk = (int)rdlane(hv, PH_K1), rk = (int)rdlane(hv, PH_K2);
                const u32 flags = rdlane(hv, PH_FLAGS);
                const u32 coin = flags & 1u, rfwd = (flags >> 1) & 1u;
                u64 myH1 = slot_h1, myH2 = slot_h2;
                if (patch) {
                    const u32 b1 = lc_bits & 1u, s1 = (lc_bits >> 1) & 1u, b2 = (lc_bits >> 2) & 1u, s2 = (lc_bits >> 3) & 1u;
                    myH1 = pc_patch(myH1, rdlane(hv, PH_MEM11), b1, s1, lane);
                    myH1 = pc_patch(myH1, rdlane(hv, PH_MEM12), b2, s2, lane);
                    myH2 = pc_patch(myH2, rdlane(hv, PH_MEM21), b1, s1, lane);
                    myH2 = pc_patch(myH2, rdlane(hv, PH_MEM22), b2, s2, lane);
                }
                // (1) remove the direction the coin picks from the reciprocal pair
                const u32 ab = (u32)((rdlane64(myH1, dk + 1) >> dk) & 1ull), ba = (u32)((rdlane64(myH1, dk) >> (dk + 1)) & 1ull);
                if (!(ab & ba)) status |= 2u;
                const int iu = coin ? dk : dk + 1, iv = coin ? dk + 1 : dk;
                const Cls c = classify(myH1, iv, iu);
                // (2) add the reverse of the single edge on the graph without the removed one
                const u32 f = (u32)((rdlane64(myH2, rk + 1) >> rk) & 1ull), bw = (u32)((rdlane64(myH2, rk) >> (rk + 1)) & 1ull);
                if ((f ^ bw) == 0u || f != rfwd) status |= 1u;
                const u32 pend = rdlane(hv, PH_PEND);
                if (pend >> 16) {   // the removed direction, if both its endpoints are in local set 2
                    const int ibig = (int)(pend & 0xFFu), ismall = (int)((pend >> 8) & 0xFFu);
                    const int fi = coin ? ibig : ismall, ti = coin ? ismall : ibig;
                    if (lane == ti) myH2 &= ~(1ull << fi);
                }
                const int ia = rfwd ? rk : rk + 1, ib = rfwd ? rk + 1 : rk;  // a->b exists, add b->a
                if (lane == ia) myH2 |= 1ull << ib;
                const Cls c2 = classify(myH2, ia, ib);
                if (!extras_fit(c, dk + 2) || !extras_fit(c2, rk + 2)) {
                    redo = true;
                } else {
                    eval_nodes<MAXT>(myH1, Hp, c, dk, tmax, -1, lane, delta);
                    eval_nodes<MAXT>(myH2, Hp, c2, rk, tmax, +1, lane, delta);
                    nonempty = true; is_dmove = true; have_delta = true;
                    clr_from = coin ? big1 : small1; clr_to = coin ? small1 : big1;
                    set_from = rfwd ? small2 : big2; set_to = rfwd ? big2 : small2;
                    dslot = rdlane(hv, PH_DSLOT); dnew = rdlane(hv, PH_PAIR2);
                    add_k = (u32)(dk + rk);
                    if (dk + 2 > 48 || rk + 2 > 48) n_big += 1;
                    my_pair1 = rdlane(hv, PH_PAIR1); my_pair2 = dnew;
                    my_bits = (coin ? 2u : 1u) | (3u << 2);
                }
            } else if (kind == PC_KIND_DELTA) {
                if (rdlane(hv, PH_D_NONEMPTY)) {
                    nonempty = true;
                    n_wide += 1;   // the producer's exact runs go through the wide evaluator
                    is_dmove = rdlane(hv, PH_D_ISDMOVE) != 0u;
                    if (lane >= 2 && lane < 16 && lane - 1 <= tmax) myd = (long long)slot_h1;
                    clr_from = rdlane(hv, PH_D_CLRF); clr_to = rdlane(hv, PH_D_CLRT);
                    set_from = rdlane(hv, PH_D_SETF); set_to = rdlane(hv, PH_D_SETT);
                    dslot = rdlane(hv, PH_D_DSLOT); dnew = rdlane(hv, PH_D_DNEW);
                    add_k = rdlane(hv, PH_D_SUMK);
                    my_pair1 = rdlane(hv, PH_PAIR1); my_pair2 = rdlane(hv, PH_PAIR2);
                    my_bits = rdlane(hv, PH_D_BITS);
                }
            }

            if (__builtin_expect(redo, 0)) {
                n_redo += 1;
                if (lane == 0) d[PD_STATUS] = PC_DEC_REDO;
                pc_barrier();   // end of this phase
                pc_barrier();   // the producer runs the proposal; next phase the slot is a DELTA
                continue;
            }

            // ---- sampled += 1; Bounds::check; accept or drop (src/lib.rs:181-194)
            sampled += 1;
            bool commit = false;
            if (!nonempty) {
                n_empty += 1;
                if (in_bounds) accepted += 1;
            } else {
                if (is_dmove) n_dmove += 1; else n_flip += 1;
                sum_k += (u64)add_k;
                if (have_delta) {
#pragma unroll
                    for (int tq = 1; tq <= MAXT; ++tq) {
                        const int sum = wave_sum_i32(delta[tq]);
                        if (lane == tq + 1) myd = (long long)sum;
                    }
                }
                const u64 ncnt = cnt + (u64)myd;
                if (ballot(cl && myd < 0 && cnt < (u64)(-myd))) status |= 8u;  // reference assert, src/lib.rs:65
                const u64 nz = ballot(cl && ncnt != 0ull);
                const u32 nlen = nz ? (u32)(64 - __clzll((long long)nz)) : 0u;
                if (nlen > count_len) count_len = nlen;
                const bool ok = ballot(cl && (ncnt < bmin || ncnt > bmax)) == 0ull;
                if (ok) {
                    accepted += 1;
                    in_bounds = true;
                    cnt = ncnt;
                    commit = true;
                }
            }
            if (lane == 0) {
                d[PD_STATUS] = commit ? PC_DEC_COMMIT : PC_DEC_NONE;
                d[PD_CLRF] = clr_from; d[PD_CLRT] = clr_to; d[PD_SETF] = set_from; d[PD_SETT] = set_to;
                d[PD_DSLOT] = is_dmove ? dslot : PC_NONE; d[PD_DNEW] = dnew;
            }
            last_commit = commit;
            lc_pair1 = commit ? my_pair1 : PC_NONE; lc_pair2 = commit ? my_pair2 : PC_NONE; lc_bits = my_bits;
            pc_barrier();
            ++cq;
        }
        if (cl) cnt_g[lane] = cnt;
        if (lane == 0) {
            st_g[0] = sampled; st_g[1] = accepted; st_g[2] = n_empty; st_g[3] = n_flip; st_g[4] = n_dmove; st_g[5] = sum_k;
            st_g[6] = count_len;
            st_g[11] = n_redo; st_g[12] = n_wide; st_g[13] = n_big;
            if (status) atomicOr((unsigned long long *)&st_g[7], (unsigned long long)status);
        }
    }
}

template <int MAXT, bool ROWS128>
__global__ __launch_bounds__(2 * WAVE, 8) void fcm_step_pc_kernel(const FcmStepParams p)
{
    extern __shared__ u64 smem[];
    if (blockIdx.x >= p.nchains) return;
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) pc_producer<MAXT, ROWS128>(p, smem);
    else pc_consumer<MAXT>(p, smem);
}
