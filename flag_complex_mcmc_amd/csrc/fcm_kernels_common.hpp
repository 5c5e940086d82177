// fcm_kernels_common.hpp — gfx950 (MI355X, CDNA4) kernels for the edge-flip MCMC hot
// path of flag-complex-mcmc.  64-wide wavefronts throughout; integer bitset
// work only (no MFMA).
//
//   fcm_step_kernel   one persistent 64-lane workgroup per chain.  Runs
//                     `nprop` iterations of the reference loop
//                     MCMCSampler::next (src/lib.rs:182-192): propose
//                     (src/lib.rs:292-325), count the change
//                     (State::apply_transition, :61-79), integer bounds check
//                     (Bounds::check, :157-160), commit or drop (:187-191).
//   fcm_count_kernel  flagser_count (src/lib.rs:51,130; src/flagser.rs:9):
//                     one wave per directed edge, simplices that start with
//                     that edge.
//
// How a proposal is counted.  The reference recounts the whole induced
// subgraph on N(a) cap N(b) + {a,b} before and after (src/lib.rs:63,71); only
// post - pre matters.  Every simplex that differs contains
// the changed directed edge, so the kernel counts exactly those:
//   E(G, u->v)[d] = #d-simplices of G that contain the edge u->v.
// removing an edge subtracts E before removal, adding one adds E after.
// All vertices of such a simplex lie in L = N(u) cap N(v) + {u,v} (static,
// src/lib.rs:330).  The wave builds the induced out-adjacency of L as one
// 64-bit mask per local vertex (lane j tests bit L[j] of row L[i]; the
// v_cmp result *is* the ballot), stages the masks in LDS, classifies each
// w in L by where it can sit relative to u->v (P: w->u,w->v  M: u->w,w->v
// S: u->w,v->w) and runs a per-lane DFS (lane = first vertex) over mask
// intersections with popcounts at the leaves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
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

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// Single-wave workgroup: orders LDS/global traffic between lanes of the wave.
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// ---------------------------------------------------------------------------
// Local vertex sets and masks.  A local set of s vertices uses NW = ceil(s/64)
// 64-bit words per mask; lane l owns local vertices l, l+64, ... (one per
// "group" g < NW).  NW is a template parameter, every loop over it is
// unrolled and every index into a per-lane array is a compile-time constant
// (runtime-indexed register arrays would go to scratch).
// ---------------------------------------------------------------------------
template <int NW> struct Mask { u64 w[NW]; };

template <int NW> __device__ __forceinline__ Mask<NW> m_zero()
{
    Mask<NW> r;
#pragma unroll
    for (int g = 0; g < NW; ++g) r.w[g] = 0ull;
    return r;
}
template <int NW> __device__ __forceinline__ Mask<NW> m_and(const Mask<NW> &a, const Mask<NW> &b)
{
    Mask<NW> r;
#pragma unroll
    for (int g = 0; g < NW; ++g) r.w[g] = a.w[g] & b.w[g];
    return r;
}
template <int NW> __device__ __forceinline__ Mask<NW> m_or(const Mask<NW> &a, const Mask<NW> &b)
{
    Mask<NW> r;
#pragma unroll
    for (int g = 0; g < NW; ++g) r.w[g] = a.w[g] | b.w[g];
    return r;
}
template <int NW> __device__ __forceinline__ bool m_any(const Mask<NW> &a)
{
    u64 r = 0;
#pragma unroll
    for (int g = 0; g < NW; ++g) r |= a.w[g];
    return r != 0ull;
}
template <int NW> __device__ __forceinline__ int m_popc(const Mask<NW> &a)
{
    int r = 0;
#pragma unroll
    for (int g = 0; g < NW; ++g) r += __popcll(a.w[g]);
    return r;
}
// bit `idx` (runtime) of a
template <int NW> __device__ __forceinline__ bool m_test(const Mask<NW> &a, int idx)
{
    u64 w = a.w[0];
#pragma unroll
    for (int g = 1; g < NW; ++g) w = ((idx >> 6) == g) ? a.w[g] : w;
    return (w >> (idx & 63)) & 1ull;
}
template <int NW> __device__ __forceinline__ void m_assign_bit(Mask<NW> &a, int idx, bool val)
{
    const u64 bit = 1ull << (idx & 63);
#pragma unroll
    for (int g = 0; g < NW; ++g)
        if ((idx >> 6) == g) a.w[g] = val ? (a.w[g] | bit) : (a.w[g] & ~bit);
}
// remove and return the lowest set bit (a must be non-empty); one call site
// regardless of NW, straight-line selects
template <int NW> __device__ __forceinline__ int m_pop_lowest(Mask<NW> &a)
{
    int x = 0;
    bool done = false;
#pragma unroll
    for (int g = 0; g < NW; ++g) {
        const bool take = !done && a.w[g] != 0ull;
        x = take ? (g * 64 + __ffsll((long long)a.w[g]) - 1) : x;
        a.w[g] = take ? (a.w[g] & (a.w[g] - 1ull)) : a.w[g];
        done = done || take;
    }
    return x;
}
template <int NW> __device__ __forceinline__ Mask<NW> lds_row(const u64 *Hs, int x)
{
    Mask<NW> r;
#pragma unroll
    for (int g = 0; g < NW; ++g) r.w[g] = Hs[x * NW + g];
    return r;
}

// ---------------------------------------------------------------------------
// Induced out-adjacency of the local vertex set.  Lane l holds local vertices
// Lv[g] (index g*64+l, valid while < s).  On return myH[g] is the out-mask of
// local vertex g*64+l over local indices 0..s-1.  One dword per lane per row
// and word: bit L[j] of row L[i]; the v_cmp result *is* the ballot.  Rows are
// 128-B multiples, so one row-read is one or few cache lines shared by the wave.
// ---------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ void build_local(const u32 *rows, u32 stride32, const u32 (&Lv)[NW], int s, int lane,
                                            Mask<NW> (&myH)[NW])
{
    constexpr int HB = 16 / NW;  // rows per batch: 16 loads in flight
    bool act[NW];
    u32 woff[NW], bit[NW];
#pragma unroll
    for (int g = 0; g < NW; ++g) {
        act[g] = g * 64 + lane < s;
        woff[g] = act[g] ? (Lv[g] >> 5) : 0u;
        bit[g] = Lv[g] & 31u;
        myH[g] = m_zero<NW>();
    }
#pragma unroll
    for (int gi = 0; gi < NW; ++gi) {
        const int cnt = min(64, s - 64 * gi);
        for (int i0 = 0; i0 < cnt; i0 += HB) {
            u32 w[HB][NW];
#pragma unroll
            for (int q = 0; q < HB; ++q) {
                const int i = min(i0 + q, cnt - 1);
                const u32 vi = rdlane(Lv[gi], i);
                const u32 *row = rows + (size_t)vi * stride32;
#pragma unroll
                for (int gj = 0; gj < NW; ++gj) w[q][gj] = row[woff[gj]];
            }
#pragma unroll
            for (int q = 0; q < HB; ++q) {
#pragma unroll
                for (int gj = 0; gj < NW; ++gj) {
                    const u64 m = ballot(act[gj] && ((w[q][gj] >> bit[gj]) & 1u));
                    if (lane == i0 + q) myH[gi].w[gj] = m;
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < NW; ++g)
        if (!act[g]) myH[g] = m_zero<NW>();
}

template <int NW>
__device__ __forceinline__ void store_local(u64 *Hs, const Mask<NW> (&myH)[NW], int lane)
{
#pragma unroll
    for (int g = 0; g < NW; ++g)
#pragma unroll
        for (int q = 0; q < NW; ++q) Hs[(g * 64 + lane) * NW + q] = myH[g].w[q];
}

// set / clear the local edge i -> j in the lane copy (the LDS copy is rewritten by store_local)
template <int NW>
__device__ __forceinline__ void local_set_edge(Mask<NW> (&myH)[NW], int lane, int i, int j, bool present)
{
#pragma unroll
    for (int g = 0; g < NW; ++g)
        if (lane == (i & 63) && (i >> 6) == g) m_assign_bit<NW>(myH[g], j, present);
}

// ---------------------------------------------------------------------------
// Per-lane DFS.  A node has T vertices of K chosen, `cand` = common
// out-neighbours still allowed (non-empty, already restricted to classes
// >= ph).  Children with class ph2 >= ph each add one simplex with T+1
// K-vertices.  delta[t] accumulates sign * (#simplices with t K-vertices).
// ---------------------------------------------------------------------------
template <int NW> struct Classes { Mask<NW> P, M, S; };

template <int NW> __device__ __forceinline__ Mask<NW> cls_mask(const Classes<NW> &c, int ph)
{
    Mask<NW> r;
#pragma unroll
    for (int g = 0; g < NW; ++g) r.w[g] = ph == 0 ? c.P.w[g] : (ph == 1 ? c.M.w[g] : c.S.w[g]);
    return r;
}
template <int NW> __device__ __forceinline__ Mask<NW> cls_ge(const Classes<NW> &c, int ph)
{
    Mask<NW> r;
#pragma unroll
    for (int g = 0; g < NW; ++g)
        r.w[g] = ph == 0 ? (c.P.w[g] | c.M.w[g] | c.S.w[g]) : (ph == 1 ? (c.M.w[g] | c.S.w[g]) : c.S.w[g]);
    return r;
}

template <int T, int MAXT, int NW, bool DETECT>
__device__ __forceinline__ void dfs_level(const Mask<NW> &cand, int ph, const u64 *Hs, const Classes<NW> &cls,
                                          int tmax, int sign, int (&delta)[MAXT + 1], u32 &overflow)
{
    if constexpr (T < MAXT) {
        if (T + 1 <= tmax) {
            const bool deeper = DETECT || (T + 2 <= tmax);
#pragma nounroll
            for (int ph2 = ph; ph2 < 3; ++ph2) {
                Mask<NW> c = m_and<NW>(cand, cls_mask<NW>(cls, ph2));
                delta[T + 1] += sign * m_popc<NW>(c);
                if (deeper) {
                    const Mask<NW> ge = cls_ge<NW>(cls, ph2);
                    while (m_any<NW>(c)) {
                        const int x = m_pop_lowest<NW>(c);
                        const Mask<NW> nc = m_and<NW>(m_and<NW>(cand, lds_row<NW>(Hs, x)), ge);
                        if (m_any<NW>(nc))
                            dfs_level<T + 1, MAXT, NW, DETECT>(nc, ph2, Hs, cls, tmax, sign, delta, overflow);
                    }
                }
            }
        } else if (DETECT) {
            overflow = 1u;  // simplices deeper than the tracked dimensions exist
        }
    } else if (DETECT) {
        overflow = 1u;
    }
}

// Count simplices through the classified local set.  Lane = first K-vertex.
template <int MAXT, int NW, bool DETECT>
__device__ __forceinline__ void eval_classes(const Mask<NW> (&myH)[NW], const u64 *Hs, const Classes<NW> &cls,
                                             int tmax, int sign, int lane, int (&delta)[MAXT + 1], u32 &overflow)
{
#pragma unroll
    for (int g = 0; g < NW; ++g) {
        const u32 cb = (u32)((cls.P.w[g] >> lane) & 1ull) | ((u32)((cls.M.w[g] >> lane) & 1ull) << 1) |
                       ((u32)((cls.S.w[g] >> lane) & 1ull) << 2);
        if (tmax >= 1) delta[1] += sign * __popc(cb);
        if (cb && tmax >= 2) {
#pragma nounroll
            for (int ph = 0; ph < 3; ++ph) {
                if ((cb >> ph) & 1u) {
                    const Mask<NW> nc = m_and<NW>(myH[g], cls_ge<NW>(cls, ph));
                    if (m_any<NW>(nc)) dfs_level<1, MAXT, NW, DETECT>(nc, ph, Hs, cls, tmax, sign, delta, overflow);
                }
            }
        }
    }
}

// E(G, u->v) on the local set: iu, iv = local indices of u and v; the edge
// u->v must be present in Hs / myH.
template <int MAXT, int NW>
__device__ __forceinline__ void eval_edge(const Mask<NW> (&myH)[NW], const u64 *Hs, int iu, int iv, int tmax, int sign,
                                          int lane, int (&delta)[MAXT + 1])
{
    const Mask<NW> outU = lds_row<NW>(Hs, iu), outV = lds_row<NW>(Hs, iv);
    Classes<NW> cls;
#pragma unroll
    for (int g = 0; g < NW; ++g) {
        const u64 inU = ballot(m_test<NW>(myH[g], iu)), inV = ballot(m_test<NW>(myH[g], iv));
        u64 nbm = ~0ull;
        if ((iu >> 6) == g) nbm &= ~(1ull << (iu & 63));
        if ((iv >> 6) == g) nbm &= ~(1ull << (iv & 63));
        cls.P.w[g] = inU & inV & nbm;            // w->u, w->v : before u
        cls.M.w[g] = outU.w[g] & inV & nbm;      // u->w, w->v : between
        cls.S.w[g] = outU.w[g] & outV.w[g] & nbm;  // u->w, v->w : after v
    }
    u32 dummy = 0;
    eval_classes<MAXT, NW, false>(myH, Hs, cls, tmax, sign, lane, delta, dummy);
}

// local vertex list of undirected edge (big, small): K then big, small
template <int NW>
__device__ __forceinline__ void load_local_list(const u32 *nb, u32 off, int k, u32 big, u32 small, int lane, u32 (&Lv)[NW])
{
#pragma unroll
    for (int g = 0; g < NW; ++g) {
        const int j = g * 64 + lane;
        Lv[g] = j < k ? nb[off + j] : (j == k ? big : small);
    }
}

// ---- the three evaluations a simple move is made of ------------------------
// single_edge_flip on undirected edge e=(big,small): returns 0 if the pair is
// reciprocal (empty transition), 1 if big->small was flipped, 2 if small->big,
// -1 if the bitmap disagrees with the static table.
template <int MAXT, int NW>
__device__ __forceinline__ int flip_eval(const u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big, u32 small,
                                         u64 *Hs, int lane, int tmax, int (&delta)[MAXT + 1])
{
    const int s = k + 2;
    u32 Lv[NW];
    load_local_list<NW>(nb, off, k, big, small, lane, Lv);
    Mask<NW> myH[NW];
    build_local<NW>(rows, stride32, Lv, s, lane, myH);
    store_local<NW>(Hs, myH, lane);
    wave_sync();
    const bool ab = m_test<NW>(lds_row<NW>(Hs, k), k + 1), ba = m_test<NW>(lds_row<NW>(Hs, k + 1), k);
    int res;
    if (ab == ba) {
        res = ab ? 0 : -1;
    } else {
        const int iu = ab ? k : k + 1, iv = ab ? k + 1 : k;  // u->v present
        eval_edge<MAXT, NW>(myH, Hs, iu, iv, tmax, -1, lane, delta);
        local_set_edge<NW>(myH, lane, iu, iv, false);
        local_set_edge<NW>(myH, lane, iv, iu, true);
        wave_sync();
        store_local<NW>(Hs, myH, lane);
        wave_sync();
        eval_edge<MAXT, NW>(myH, Hs, iv, iu, tmax, +1, lane, delta);
        res = ab ? 1 : 2;
    }
    wave_sync();
    return res;
}

// double_edge_move step 1: subtract the simplices through one direction of the
// reciprocal pair (big,small).  coin=1 removes big->small.  Returns false if
// the pair is not reciprocal in the bitmap.
template <int MAXT, int NW>
__device__ __forceinline__ bool del_eval(const u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big, u32 small,
                                         u32 coin, u64 *Hs, int lane, int tmax, int (&delta)[MAXT + 1])
{
    const int s = k + 2;
    u32 Lv[NW];
    load_local_list<NW>(nb, off, k, big, small, lane, Lv);
    Mask<NW> myH[NW];
    build_local<NW>(rows, stride32, Lv, s, lane, myH);
    store_local<NW>(Hs, myH, lane);
    wave_sync();
    const bool ab = m_test<NW>(lds_row<NW>(Hs, k), k + 1), ba = m_test<NW>(lds_row<NW>(Hs, k + 1), k);
    const int iu = coin ? k : k + 1, iv = coin ? k + 1 : k;
    eval_edge<MAXT, NW>(myH, Hs, iu, iv, tmax, -1, lane, delta);
    wave_sync();
    return ab && ba;
}

// double_edge_move step 2: on the graph without dfrom->dto, add the reverse of
// the single edge of (big,small) and add the simplices through it.  fwd=1
// means big->small is the existing direction.
template <int MAXT, int NW>
__device__ __forceinline__ void add_eval(const u32 *rows, u32 stride32, const u32 *nb, u32 off, int k, u32 big, u32 small,
                                         u32 fwd, u32 dfrom, u32 dto, u64 *Hs, int lane, int tmax, int (&delta)[MAXT + 1])
{
    const int s = k + 2;
    u32 Lv[NW];
    load_local_list<NW>(nb, off, k, big, small, lane, Lv);
    Mask<NW> myH[NW];
    build_local<NW>(rows, stride32, Lv, s, lane, myH);
    // the pending removal, if both its endpoints are local
    int fi = -1, ti = -1;
#pragma unroll
    for (int g = 0; g < NW; ++g) {
        const bool act = g * 64 + lane < s;
        const u64 mf = ballot(act && Lv[g] == dfrom), mt = ballot(act && Lv[g] == dto);
        if (mf) fi = g * 64 + __ffsll((long long)mf) - 1;
        if (mt) ti = g * 64 + __ffsll((long long)mt) - 1;
    }
    if (fi >= 0 && ti >= 0) local_set_edge<NW>(myH, lane, fi, ti, false);
    const int ia = fwd ? k : k + 1, ib = fwd ? k + 1 : k;  // a->b exists, add b->a
    local_set_edge<NW>(myH, lane, ib, ia, true);
    store_local<NW>(Hs, myH, lane);
    wave_sync();
    eval_edge<MAXT, NW>(myH, Hs, ib, ia, tmax, +1, lane, delta);
    wave_sync();
}

// dispatch on the local set size (wave-uniform)
#define FCM_DISPATCH_NW(S_, CALL1, CALL2, CALL4)                 \
    do {                                                         \
        if ((S_) <= 64) { CALL1; }                               \
        else if constexpr (MAXNW >= 2) {                         \
            if ((S_) <= 128) { CALL2; }                          \
            else if constexpr (MAXNW >= 4) { CALL4; }            \
        }                                                        \
    } while (0)

// ---------------------------------------------------------------------------
// Step kernel
// ---------------------------------------------------------------------------
// MINW = minimum waves per SIMD the register allocator must leave room for
// (4 => at most 128 VGPRs; the rare wide-neighbourhood paths then spill instead
// of costing every chain its residency).
template <int MAXT, int MAXNW, int MINW>
__global__ __launch_bounds__(WAVE, MINW) void fcm_step_kernel(const FcmStepParams p)
{
    __shared__ u64 Hs[WAVE * MAXNW * MAXNW];
    const int lane = threadIdx.x;
    const u32 chain = blockIdx.x;
    if (chain >= p.nchains) return;

    u32 *rows = p.rows + (size_t)chain * p.rows_per_chain;
    u32 *dbl = p.dbl + (size_t)chain * p.dbl_stride;
    u64 *cnt_g = (u64 *)p.counts + (size_t)chain * FCM_DEV_MAX_COUNTS;
    u64 *st_g = (u64 *)p.stats + (size_t)chain * FCM_DEV_NSTATS;
    const u32 *nb = p.nb;

    const int NC = p.ncounts;
    const int tmax = NC - 2;
    const bool cl = lane < NC;
    // lane d holds count[d] and its bounds (zero-padded, src/util.rs:53-57)
    u64 cnt = cl ? cnt_g[lane] : 0ull;
    const u64 bmin = cl ? p.bmin[lane] : 0ull;
    const u64 bmax = cl ? p.bmax[lane] : ~0ull;

    u64 sampled = st_g[0], accepted = st_g[1], n_empty = st_g[2], n_flip = st_g[3], n_dmove = st_g[4], sum_k = st_g[5];
    u32 count_len = (u32)st_g[6];
    u32 status = (u32)st_g[7];

    const u32 U = p.U, D = p.D;
    const u64 Mtot = (u64)U + D;
    const u32 k0 = (u32)p.seed, k1 = (u32)(p.seed >> 32);
    const u32 gchain = p.first_chain + chain;
    const u32 stride32 = p.stride32;

    // is the current state inside the bounds?  (decides whether an empty
    // transition is "accepted", src/lib.rs:186-187)
    bool in_bounds = ballot(cl && (cnt < bmin || cnt > bmax)) == 0ull;

    for (u64 done = 0; done < p.nprop; done += WAVE) {
        // ---- batch: lane s draws proposal `sampled + s` ------------------
        const u64 t = sampled + (u64)lane;
        u32 w[4];
        philox4x32_10((u32)t, (u32)(t >> 32), gchain, 0u, k0, k1, w);
        const int l_move = ((u64)w[0] < p.cum0) ? 0 : (((u64)w[0] < p.cum1) ? 1 : 2);
        const u32 l_coin = w[1] & 1u;
        const u64 x64 = (u64)w[2] | ((u64)w[3] << 32);
        const u64 l_idx = __umul64hi(x64, l_move == 0 ? Mtot : (u64)D);
        FcmEdgeEntry l_e = {0u, 0u, 0u, 0u};
        if (l_move == 0 && l_idx < U) l_e = p.etab[l_idx];

        const int nbatch = (int)min((u64)WAVE, p.nprop - done);
        for (int sidx = 0; sidx < nbatch; ++sidx) {
            const int move = (int)rdlane((u32)l_move, sidx);
            const u32 coin = rdlane(l_coin, sidx);
            const u64 idx = rdlane64(l_idx, sidx);

            int delta[MAXT + 1];
#pragma unroll
            for (int q = 0; q <= MAXT; ++q) delta[q] = 0;

            bool nonempty = false;
            // pending commit (uniform)
            u32 c_clr_from = 0, c_clr_to = 0, c_set_from = 0, c_set_to = 0;
            u32 c_slot = 0, c_newdbl = 0;
            bool is_dmove = false;

            if (move == 0) {
                // ---- single_edge_flip (src/lib.rs:292-299) -----------------
                if (Mtot > 0 && idx < U) {
                    const u32 a = rdlane(l_e.big, sidx), b = rdlane(l_e.small, sidx);
                    const u32 off = rdlane(l_e.nb_off, sidx);
                    const int k = (int)rdlane(l_e.k, sidx);
                    int res = 0;
                    FCM_DISPATCH_NW(k + 2,
                        (res = flip_eval<MAXT, 1>(rows, stride32, nb, off, k, a, b, Hs, lane, tmax, delta)),
                        (res = flip_eval<MAXT, 2>(rows, stride32, nb, off, k, a, b, Hs, lane, tmax, delta)),
                        (res = flip_eval<MAXT, 4>(rows, stride32, nb, off, k, a, b, Hs, lane, tmax, delta)));
                    if (res < 0) status |= 1u;  // table says adjacent, bitmap says not
                    if (res > 0) {
                        nonempty = true;
                        c_clr_from = res == 1 ? a : b; c_clr_to = res == 1 ? b : a;
                        c_set_from = c_clr_to; c_set_to = c_clr_from;
                        sum_k += (u64)k;
                    }
                }
            } else if (move == 1) {
                // ---- double_edge_move (src/lib.rs:304-325) -----------------
                if (D > 0) {
                    const u32 slot = (u32)idx;
                    const u32 ed = dbl[slot];
                    const FcmEdgeEntry de = p.etab[ed];
                    // 64 candidate draws for the single edge, first valid wins
                    // (uniform directed edge, retry while reciprocal: :308-313)
                    const u64 tt = sampled;  // this proposal's step index
                    u32 v[4];
                    philox4x32_10((u32)tt, (u32)(tt >> 32), gchain, (u32)(lane >> 1) + 1u, k0, k1, v);
                    const u64 y64 = (lane & 1) ? ((u64)v[2] | ((u64)v[3] << 32)) : ((u64)v[0] | ((u64)v[1] << 32));
                    const u64 rr = __umul64hi(y64, Mtot);
                    bool valid = rr < U;
                    FcmEdgeEntry ce = {0u, 0u, 0u, 0u};
                    u32 fwd = 0;
                    if (valid) {
                        ce = p.etab[rr];
                        const u32 wf = rows[(size_t)ce.big * stride32 + (ce.small >> 5)];
                        const u32 wb = rows[(size_t)ce.small * stride32 + (ce.big >> 5)];
                        fwd = (wf >> (ce.small & 31u)) & 1u;
                        const u32 bwd = (wb >> (ce.big & 31u)) & 1u;
                        valid = (fwd ^ bwd) != 0u;
                    }
                    const u64 vm = ballot(valid);
                    if (vm) {
                        const int first = __ffsll((long long)vm) - 1;
                        const u32 r = (u32)rdlane64(rr, first);
                        const u32 rbig = rdlane(ce.big, first), rsmall = rdlane(ce.small, first);
                        const u32 roff = rdlane(ce.nb_off, first);
                        const int rk = (int)rdlane(ce.k, first);
                        const u32 rfwd = rdlane(fwd, first);
                        const u32 ea = rfwd ? rbig : rsmall, eb = rfwd ? rsmall : rbig;  // ea->eb is the single edge
                        // delme: coin ? (big->small) : (small->big) of the reciprocal pair (:316-320)
                        const u32 dfrom = coin ? de.big : de.small, dto = coin ? de.small : de.big;
                        nonempty = true; is_dmove = true;
                        // (1) remove delme: subtract simplices through it
                        bool okd = true;
                        const int dk = (int)de.k;
                        FCM_DISPATCH_NW(dk + 2,
                            (okd = del_eval<MAXT, 1>(rows, stride32, nb, de.nb_off, dk, de.big, de.small, coin, Hs, lane, tmax, delta)),
                            (okd = del_eval<MAXT, 2>(rows, stride32, nb, de.nb_off, dk, de.big, de.small, coin, Hs, lane, tmax, delta)),
                            (okd = del_eval<MAXT, 4>(rows, stride32, nb, de.nb_off, dk, de.big, de.small, coin, Hs, lane, tmax, delta)));
                        if (!okd) status |= 2u;  // slot list says reciprocal, bitmap says not
                        // (2) add eb->ea on the graph without delme: add simplices through it
                        FCM_DISPATCH_NW(rk + 2,
                            (add_eval<MAXT, 1>(rows, stride32, nb, roff, rk, rbig, rsmall, rfwd, dfrom, dto, Hs, lane, tmax, delta)),
                            (add_eval<MAXT, 2>(rows, stride32, nb, roff, rk, rbig, rsmall, rfwd, dfrom, dto, Hs, lane, tmax, delta)),
                            (add_eval<MAXT, 4>(rows, stride32, nb, roff, rk, rbig, rsmall, rfwd, dfrom, dto, Hs, lane, tmax, delta)));
                        c_clr_from = dfrom; c_clr_to = dto;
                        c_set_from = eb; c_set_to = ea;
                        c_slot = slot; c_newdbl = r;
                        sum_k += (u64)de.k + (u64)rk;
                    }
                }
            } else {
                status |= 4u;  // clique moves are not built
            }

            // ---- sampled += 1; Bounds::check; accept or drop ---------------
            sampled += 1;
            if (!nonempty) {
                n_empty += 1;
                if (in_bounds) accepted += 1;
            } else {
                if (is_dmove) n_dmove += 1; else n_flip += 1;
                long long myd = 0;
#pragma unroll
                for (int tq = 1; tq <= MAXT; ++tq) {
                    if (tq <= tmax) {
                        const long long sum = wave_sum_i64((long long)delta[tq]);
                        if (lane == tq + 1) myd = sum;
                    }
                }
                const u64 ncnt = cnt + (u64)myd;
                if (ballot(cl && myd < 0 && cnt < (u64)(-myd))) status |= 8u;  // reference assert, src/lib.rs:65
                // flag_count never shrinks in length (src/lib.rs:72-74)
                const u64 nz = ballot(cl && ncnt != 0ull);
                const u32 nlen = nz ? (u32)(64 - __clzll((long long)nz)) : 0u;
                if (nlen > count_len) count_len = nlen;
                const bool ok = ballot(cl && (ncnt < bmin || ncnt > bmax)) == 0ull;
                if (ok) {
                    accepted += 1;
                    in_bounds = true;
                    cnt = ncnt;
                    if (lane == 0) {
                        u32 *pc = rows + (size_t)c_clr_from * stride32 + (c_clr_to >> 5);
                        *pc &= ~(1u << (c_clr_to & 31u));
                        u32 *ps = rows + (size_t)c_set_from * stride32 + (c_set_to >> 5);
                        *ps |= (1u << (c_set_to & 31u));
                        if (is_dmove) dbl[c_slot] = c_newdbl;
                    }
                    wave_sync();
                }
            }
        }
    }

    if (cl) cnt_g[lane] = cnt;
    if (lane == 0) {
        st_g[0] = sampled; st_g[1] = accepted; st_g[2] = n_empty; st_g[3] = n_flip;
        st_g[4] = n_dmove; st_g[5] = sum_k; st_g[6] = count_len; st_g[7] = status;
    }
}

