// fcm_kernels.hip — gfx950 (MI355X, CDNA4) kernels for the edge-flip MCMC hot
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
// Induced out-adjacency of the local vertex set.  Lane j (< s) holds local
// vertex Lv.  Returns this lane's out-mask over local indices 0..s-1.
// One dword per lane per row: bit L[j] of row L[i].  Rows are 128-B multiples,
// so one row-read is one or few cache lines, shared by the 64 lanes.
// ---------------------------------------------------------------------------
#define FCM_HB 16  // rows in flight per batch
__device__ __forceinline__ u64 build_local(const u32 *rows, u32 stride32, u32 Lv, int s, int lane)
{
    const bool act = lane < s;
    const u32 woff = act ? (Lv >> 5) : 0u;
    const u32 bit = Lv & 31u;
    u64 myH = 0;
    for (int i0 = 0; i0 < s; i0 += FCM_HB) {
        u32 w[FCM_HB];
#pragma unroll
        for (int q = 0; q < FCM_HB; ++q) {
            const int i = min(i0 + q, s - 1);
            const u32 vi = rdlane(Lv, i);
            w[q] = rows[(size_t)vi * stride32 + woff];
        }
#pragma unroll
        for (int q = 0; q < FCM_HB; ++q) {
            const u64 m = ballot(act && ((w[q] >> bit) & 1u));
            if (lane == i0 + q) myH = m;
        }
    }
    return act ? myH : 0ull;
}

// ---------------------------------------------------------------------------
// Per-lane DFS.  A node has T vertices of K chosen, `cand` = common
// out-neighbours still allowed (non-empty, already restricted to classes
// >= ph).  Children with class ph2 >= ph each add one simplex with T+1
// K-vertices.  delta[t] accumulates sign * (#simplices with t K-vertices).
// ---------------------------------------------------------------------------
template <int T, int MAXT, bool DETECT>
__device__ __forceinline__ void dfs_level(u64 cand, int ph, const u64 *Hs, u64 P, u64 M, u64 S,
                                          int tmax, int sign, int (&delta)[MAXT + 1], u32 &overflow)
{
    if constexpr (T < MAXT) {
        if (T + 1 <= tmax) {
            const bool deeper = DETECT || (T + 2 <= tmax);
#pragma nounroll
            for (int ph2 = ph; ph2 < 3; ++ph2) {
                const u64 cm = ph2 == 0 ? P : (ph2 == 1 ? M : S);
                u64 c = cand & cm;
                delta[T + 1] += sign * __popcll(c);
                if (deeper) {
                    const u64 ge = ph2 == 0 ? (P | M | S) : (ph2 == 1 ? (M | S) : S);
                    while (c) {
                        const int x = __ffsll((long long)c) - 1;
                        c &= c - 1;
                        const u64 nc = cand & Hs[x] & ge;
                        if (nc) dfs_level<T + 1, MAXT, DETECT>(nc, ph2, Hs, P, M, S, tmax, sign, delta, overflow);
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
template <int MAXT, bool DETECT>
__device__ __forceinline__ void eval_classes(u64 myH, const u64 *Hs, u64 P, u64 M, u64 S, int tmax, int sign,
                                             int lane, int (&delta)[MAXT + 1], u32 &overflow)
{
    const u32 cb = (u32)((P >> lane) & 1ull) | ((u32)((M >> lane) & 1ull) << 1) | ((u32)((S >> lane) & 1ull) << 2);
    if (tmax >= 1) delta[1] += sign * __popc(cb);
    if (cb && tmax >= 2) {
#pragma nounroll
        for (int ph = 0; ph < 3; ++ph) {
            if ((cb >> ph) & 1u) {
                const u64 ge = ph == 0 ? (P | M | S) : (ph == 1 ? (M | S) : S);
                const u64 nc = myH & ge;
                if (nc) dfs_level<1, MAXT, DETECT>(nc, ph, Hs, P, M, S, tmax, sign, delta, overflow);
            }
        }
    }
}

// E(G, u->v) on the local set: iu, iv = local indices of u and v; the edge
// u->v must be present in Hs.
template <int MAXT>
__device__ __forceinline__ void eval_edge(u64 myH, const u64 *Hs, int iu, int iv, int tmax, int sign, int lane,
                                          int (&delta)[MAXT + 1])
{
    const u64 outU = Hs[iu], outV = Hs[iv];
    const u64 inU = ballot((myH >> iu) & 1ull), inV = ballot((myH >> iv) & 1ull);
    const u64 nbm = ~((1ull << iu) | (1ull << iv));
    const u64 P = inU & inV & nbm;   // w->u, w->v : before u
    const u64 M = outU & inV & nbm;  // u->w, w->v : between
    const u64 S = outU & outV & nbm; // u->w, v->w : after v
    u32 dummy = 0;
    eval_classes<MAXT, false>(myH, Hs, P, M, S, tmax, sign, lane, delta, dummy);
}

// ---------------------------------------------------------------------------
// Step kernel
// ---------------------------------------------------------------------------
template <int MAXT>
__global__ __launch_bounds__(WAVE) void fcm_step_kernel(const FcmStepParams p)
{
    __shared__ u64 Hs[WAVE];
    const int lane = threadIdx.x;
    const u32 chain = blockIdx.x;
    if (chain >= p.nchains) return;

    u32 *rows = p.rows + (size_t)chain * p.rows_per_chain;
    u32 *dbl = p.dbl + (size_t)chain * p.dbl_stride;
    u64 *cnt_g = (u64 *)p.counts + (size_t)chain * FCM_DEV_MAX_COUNTS;
    u64 *st_g = (u64 *)p.stats + (size_t)chain * FCM_DEV_NSTATS;

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

        const int nb = (int)min((u64)WAVE, p.nprop - done);
        for (int sidx = 0; sidx < nb; ++sidx) {
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
                    const int s = k + 2;
                    const u32 Lv = lane < k ? p.nb[off + lane] : (lane == k ? a : b);
                    u64 myH = build_local(rows, stride32, Lv, s, lane);
                    Hs[lane] = myH;
                    wave_sync();
                    const u32 ab = (u32)((Hs[k] >> (k + 1)) & 1ull), ba = (u32)((Hs[k + 1] >> k) & 1ull);
                    if (!(ab | ba)) status |= 1u;  // table says adjacent, bitmap says not
                    if ((ab ^ ba) != 0u) {
                        nonempty = true;
                        const int iu = ab ? k : k + 1, iv = ab ? k + 1 : k;  // u->v present
                        eval_edge<MAXT>(myH, Hs, iu, iv, tmax, -1, lane, delta);
                        wave_sync();
                        if (lane == iu) { myH &= ~(1ull << iv); Hs[lane] = myH; }
                        if (lane == iv) { myH |= (1ull << iu); Hs[lane] = myH; }
                        wave_sync();
                        eval_edge<MAXT>(myH, Hs, iv, iu, tmax, +1, lane, delta);
                        wave_sync();
                        c_clr_from = ab ? a : b; c_clr_to = ab ? b : a;
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
                        {
                            const int k = (int)de.k, s = k + 2;
                            const u32 Lv = lane < k ? p.nb[de.nb_off + lane] : (lane == k ? de.big : de.small);
                            const u64 myH = build_local(rows, stride32, Lv, s, lane);
                            Hs[lane] = myH;
                            wave_sync();
                            const u32 ab = (u32)((Hs[k] >> (k + 1)) & 1ull), ba = (u32)((Hs[k + 1] >> k) & 1ull);
                            if (!(ab & ba)) status |= 2u;  // slot list says reciprocal, bitmap says not
                            const int iu = coin ? k : k + 1, iv = coin ? k + 1 : k;
                            eval_edge<MAXT>(myH, Hs, iu, iv, tmax, -1, lane, delta);
                            wave_sync();
                        }
                        // (2) add eb->ea on the graph without delme: add simplices through it
                        {
                            const int k = rk, s = k + 2;
                            const u32 Lv = lane < k ? p.nb[roff + lane] : (lane == k ? rbig : rsmall);
                            u64 myH = build_local(rows, stride32, Lv, s, lane);
                            const bool act = lane < s;
                            const u64 mf = ballot(act && Lv == dfrom), mt = ballot(act && Lv == dto);
                            if (mf && mt) {
                                const int fi = __ffsll((long long)mf) - 1, ti = __ffsll((long long)mt) - 1;
                                if (lane == fi) myH &= ~(1ull << ti);
                            }
                            const int ia = rfwd ? k : k + 1, ib = rfwd ? k + 1 : k;
                            if (lane == ib) myH |= (1ull << ia);
                            Hs[lane] = myH;
                            wave_sync();
                            eval_edge<MAXT>(myH, Hs, ib, ia, tmax, +1, lane, delta);
                            wave_sync();
                        }
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

// ---------------------------------------------------------------------------
// Global count: flagser_count.  One wave per directed edge u->v (grid-stride):
// the simplices whose first two vertices are u, v have their remaining
// vertices in C = out(u) & out(v); count directed simplices inside C.
// ---------------------------------------------------------------------------
#define FCM_COUNT_MAXT 14  // dims up to 15
__global__ __launch_bounds__(WAVE) void fcm_count_kernel(const FcmCountParams p)
{
    __shared__ u64 Hs[WAVE];
    __shared__ u32 Lc[WAVE];
    const int lane = threadIdx.x;
    const u32 nwords = (p.n + 31u) >> 5;
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
            // inclusive scan over the wave
            u32 inc = c;
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
                if (pos < WAVE) Lc[pos] = w * 32u + b;
                ++pos;
            }
            total += tot;
        }
        if (total == 0) continue;
        if (total > WAVE) { toolarge = 1u; continue; }
        wave_sync();
        const u32 Lv = lane < (int)total ? Lc[lane] : 0u;
        const u64 myH = build_local(p.rows, p.stride32, Lv, (int)total, lane);
        Hs[lane] = myH;
        wave_sync();
        int delta[FCM_COUNT_MAXT + 1];
#pragma unroll
        for (int q = 0; q <= FCM_COUNT_MAXT; ++q) delta[q] = 0;
        const u64 S = total >= 64 ? ~0ull : ((1ull << total) - 1ull);
        eval_classes<FCM_COUNT_MAXT, true>(myH, Hs, 0ull, 0ull, S, FCM_COUNT_MAXT, +1, lane, delta, overflow);
#pragma unroll
        for (int q = 1; q <= FCM_COUNT_MAXT; ++q) acc[q] += (u64)(u32)delta[q];
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
// Launchers
// ---------------------------------------------------------------------------
extern "C" int fcm_launch_step(const FcmStepParams *p, int maxt_variant, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(p->nchains), block(WAVE);
    if (maxt_variant <= 6)
        fcm_step_kernel<6><<<grid, block, 0, st>>>(*p);
    else
        fcm_step_kernel<14><<<grid, block, 0, st>>>(*p);
    return (int)hipGetLastError();
}

extern "C" int fcm_launch_count(const FcmCountParams *p, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (p->m == 0) return 0;
    const u64 maxgrid = 256ull * 32ull;
    dim3 grid((unsigned)(p->m < maxgrid ? p->m : maxgrid)), block(WAVE);
    hipLaunchKernelGGL(fcm_count_kernel, grid, block, 0, st, *p);
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
