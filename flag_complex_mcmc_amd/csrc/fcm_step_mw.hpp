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
//   head (LDS)   number of proposals decided so far.  A wave whose counts are
//                ready waits until head == q: it then holds the token.
//   ring (LDS)   the records of the last 4W proposals: the pair(s) a proposal
//                changes if accepted, the slot of the reciprocal list it
//                rewrites -- published as soon as the proposal knows them
//                (before its evaluations: "staged"), and marked accepted or
//                dropped when it is decided.
//   checks       a wave holds the records snap..q-1 against its own reads while
//                it waits for the token, staged ones included: by the time the
//                token arrives, nothing of that is left to do.  A conflict with
//                a record that is only staged is looked at again when that
//                proposal is decided (it may be dropped).  Under the token: one
//                look at the state words of snap..q-1 (a proposal that was run
//                again after being checked carries a mark: its record may have
//                changed, it is checked again), then -- no hit, the rule,
//                > 99 % -- the wave's counts are those of the exact sequential
//                state: bounds check against the chain's counts (LDS), the
//                record's state word, head = q+1, and after that the commit
//                (two bitmap words, the slot list).  A hit, or a proposal that
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
// Memory ordering.  head orders LDS only (records, counts): it is passed on without waiting for the commit's global stores.
// Visibility of those stores is tracked apart: wave w publishes vis[w] = "every proposal of mine below this index is
// in memory" at the start of each proposal, after its own earlier stores have completed (s_waitcnt vmcnt(0), long
// hidden behind the decision's tail).  snap = min over the waves of vis: every proposal below snap is visible to
// whatever the wave loads from then on; the proposals snap..q-1 -- at most 2W-1 of them -- are held against its reads.
// A record's place in the ring is taken again 4W proposals later, by a proposal that starts after q+3W is decided: every
// reader of it (up to q+2W-1) has been decided by then.  LDS executes a wave's operations in order: a record's state word is
// written after its other words, head after the state word, with no wait in between.  All waves of a workgroup run on one CU and share its vector L1: workgroup scope.
#pragma once
#ifndef MW_PROBE
#define MW_PROBE 0
#endif

#ifndef MW_MINW
#define MW_MINW 8
#endif
// Ablation probes (tools/ablate.sh; WRONG RESULTS BY DESIGN, never in the product build): bits of MW_ABL leave a part of the
// proposal out so that the bench shows what the chip spends on it -- 1: the evaluations, 2: the bitmap reads of the builds
// (masks made up from the vertex ids), 4: the commit's global stores, 8: the loads of the two words a commit rewrites, 16: the
// in-order wait (every wave decides when it is ready), 32: the vertex-list loads (lists made up).
#ifndef MW_ABL
#define MW_ABL 0
#endif
// Code-shape knobs.  They change nothing in what the kernel computes; they decide whether hipcc finds a clean
// allocation at 64 VGPRs / 80 SGPRs for a given variant (tools/scratch_census.sh; the defaults below are what
// tools/tune_knobs.sh found clean for every variant; `make EXTRA=-DMW_K_..=..` overrides them for all variants).  ZERO: the count registers are zeroed by instructions of their own.  LANE: the lane id
// is made opaque per proposal, so that comparisons with it are not hoisted out of the loop.  EVLOOP: the two
// evaluations of a proposal run as a loop over one inlined evaluator (1) or as two inlined copies (0).
#ifndef MW_K_ZERO
#define MW_K_ZERO 1
#endif
#ifndef MW_K_LANE2
#define MW_K_LANE2 1
#endif
#ifndef MW_K_LANE
#define MW_K_LANE 1
#endif
#ifndef MW_VEC_MIN
#define MW_VEC_MIN 4u   // records to check from which the lookup maps pay (fewer: each is looked at exactly)
#endif
#ifndef MW_SLEEP_NEAR
#define MW_SLEEP_NEAR 12  // s_sleep argument (x 64 cycles) between polls when the token is one decision away, few-records path
#endif
#ifndef MW_SLEEP_NEAR2
#define MW_SLEEP_NEAR2 24 // ... the same at W = 2, where the partner wave is in the middle of a proposal more often than not (16 .. 32: + 0.3 % on the headline against 12)
#endif
#ifndef MW_PRIO_OLDEST
#define MW_PRIO_OLDEST 2   // s_setprio of the wave that holds its chain's oldest undecided proposal ...
#endif
#ifndef MW_PRIO_TOKEN
#define MW_PRIO_TOKEN 3    // ... and of the token holder
#endif
#ifndef MW_SPIN_NEAR
#define MW_SPIN_NEAR 1     // 1: the wave next in line (W >= 4) polls without dozing (+2 % on configs[4], neutral elsewhere)
#endif
#ifndef MW_SLEEP_FAR
#define MW_SLEEP_FAR 10   // s_sleep argument (x 64 cycles) between polls of a wave (W >= 4) four or more decisions away from the token ...
#endif
#ifndef MW_SLEEP_MID
#define MW_SLEEP_MID 4    // ... two or three away (0: no doze)
#endif
#ifndef MW_K_EVLOOP
#define MW_K_EVLOOP 0
#endif
#ifndef MW_MERGED
#define MW_MERGED 1       // 1: a flip's two evaluations as one signed evaluation (eval_flip_merged, fcm_kernels_common.hpp)
#endif
#define MW_NONE 0xFFFFFFFFu
// state word of a record: (proposal index << 4) | flags | phase.  Phase 0: not there (not staged yet, or being
// written again by its exact run); 1: staged -- what the proposal changes if it is accepted; 2: decided.
#define MS_STAGED 1u
#define MS_DECIDED 2u
#define MS_ACCEPTED 4u
#define MS_REDONE 8u    // decided on an exact run: the record may differ from what was staged

// chain context, u32 words in LDS: what the out-of-line parts (table fill, exact run) need, so that the hot loop
// does not have to keep it in registers
enum { MC_ROWS = 0, MC_DBL = 2, MC_NB = 4, MC_ETAB = 6, MC_ROWS_BYTES = 8, MC_SEED = 10, MC_SAMPLED0 = 12, MC_CUM0 = 14, MC_CUM1 = 16,
       MC_CUM2 = 18, MC_U = 20, MC_D, MC_STRIDE32, MC_GCHAIN, MC_MAXNW, MC_W, MC_GUARD = 26, MC_XW = 28, MC_CLIM = 30 /* u32 words of the chain's
       mutable record: what a commit's word indices are held against */, MC_WORDS = 32 };

// LDS map in u64 words:
//   shared    E[9]: {count, inside-the-bounds flag (entry 0), bmin, bmax} | head | ctx[16] | vis[8] | tallies[2 u32] | ring[4W][7]
//             (<= 8 count entries: tmax <= 6; entry 8 = {0, 0, 0, ~0} is what the lanes without a count read)
//   per wave  Hp[64] | arc list[64] | draw table: 28 entries of 14 u32 [196] | tallies[9]   (the arc list doubles as the
//             exact run's per-lane results; tallies: 17 u32, MA_*)
//   wide evaluator (one: only the token holder runs it)
#define MW_SHARED_WORDS 62u
#define MW_TALLY_OFF 61u                                     // [0] proposals that checked a record again under the token, [1] that waited for a staged record's decision
#define MW_HEAD_OFF 36u
#define MW_CTX_OFF 37u
#define MW_VIS_OFF 53u
static_assert(MW_CTX_OFF + MC_WORDS / 2 == MW_VIS_OFF && MW_VIS_OFF + 8u == MW_TALLY_OFF && MW_TALLY_OFF + 1u == MW_SHARED_WORDS, "shared LDS map");
#define MW_TBL_WORDS 14u
#ifndef MW_TBL_N
#ifndef MW_TBL_N
#define MW_TBL_N 28u
#endif
#endif
#define MW_TALLY_WAVE_OFF (128u + (MW_TBL_N * MW_TBL_WORDS) / 2u)
#define MW_WAVE_WORDS (MW_TALLY_WAVE_OFF + 9u)
#define SR_WORDS_C 14u   // = SR_WORDS (the enum below)
#define MW_REC_WORDS 14u                                  // a record: SR_* words, u32
#define MW_RING_WORDS(W) (4u * (W) * (MW_REC_WORDS / 2u))   // u64 words
static_assert(MW_REC_WORDS % 2u == 0u && SR_WORDS_C == MW_REC_WORDS, "a record is a whole number of 64-bit words (mw_stage stores it as such)");
__host__ __device__ constexpr inline unsigned fcm_mw_lds_words(int NW, int W)
{
    return MW_SHARED_WORDS + MW_RING_WORDS((unsigned)W) + (unsigned)W * MW_WAVE_WORDS + fcm_lds_words(NW < 2 ? 2 : NW);
}
// 4096 chains at W = 2 are all resident only if 16 workgroups fit the 160 KiB of a CU: 10 KiB each
static_assert(MW_TBL_N != 28u || fcm_mw_lds_words(2, 2) * 8u <= 10240u, "the W = 2 workgroup must stay within 10 KiB of LDS (16 per CU)");

__device__ __forceinline__ void mw_barrier()   // orders LDS only
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ u32 mw_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 mw_uni64(u64 v) { return (u64)mw_uni((u32)v) | ((u64)mw_uni((u32)(v >> 32)) << 32); }

template <bool ROWS128>
__device__ __forceinline__ u64 mw_build(const rsrc_t rr, u32 stride32, u32 Lv, int s, int lane)
{
#if MW_ABL & 2   // (no bitmap reads: masks made up from the vertex ids, the pair itself one way)
    {
        u64 h = ((u64)Lv * 0x9E3779B97F4A7C15ull) ^ ((u64)Lv << 21);
        h &= (h >> 7) | (h << 3);
        h = lane < s ? (h & (s >= 64 ? ~0ull : ((1ull << s) - 1ull))) : 0ull;
        const int k = s - 2;
        h &= ~(1ull << lane);
        if (lane == k + 1) h |= 1ull << k;
        if (lane == k) h &= ~(1ull << (k + 1));
        return h;
    }
#endif
    if constexpr (ROWS128) return build_local_rows128(rr, Lv, s, lane);
    else return build_local_loop16(rr, stride32, Lv, s, lane);
}

// ---- sparse state (graphs whose rows are long and whose local sets are tiny: BASELINE configs[4], n = 30000, k ~ 0.1) -----------
// Per chain, instead of row bitmaps: two bits per adjacent pair e of pr(G) -- bit 2e = big -> small, bit 2e+1 = small -> big,
// the reference's edgebits layout (src/io.rs:152-159) -- 250 KB per chain at n = 30000 instead of 115 MB.  The static side
// names, behind a pair's neighbour list in `nb`, the pairs among its local set: for every local pair (i < j, lexicographic;
// local indices as in the list: K ascending, then big, then small; the pair itself last) two words: its pair id (MW_NONE: not
// adjacent) and i | j << 8 | (L[i] is that pair's `big` ? 0 : 1) << 16.  A build is then one coalesced read of those entries
// (none at all for k = 0, nine proposals in ten on configs[4]) and one dword gather from the chain's bits.  Local sets of up to
// 11 vertices (55 local pairs, one per lane); the host selects the layout only for such graphs.
#define MW_SPARSE_MAX_S 11
__device__ __forceinline__ u64 mw_build_sparse(const rsrc_t rbits, const u32 *nb, u32 off, int k, u32 pid, int lane)
{
    const int s = k + 2, T = s * (s - 1) / 2;
    u32 id = MW_NONE, ij = 0u;
    if (k == 0) {                       // the pair alone: nothing to look up
        id = pid; ij = 0u | (1u << 8);
    } else if (lane < T) {
        const u32 base = (off + (u32)k + 1u) & ~1u;   // (the entries start at an even word: 8-byte reads)
        const uint2 e = *(const uint2 *)(nb + base + 2u * (u32)lane);
        id = e.x; ij = e.y;
    }
    const bool have = lane < T && id != MW_NONE;
    const u32 word = have ? (u32)__builtin_amdgcn_raw_buffer_load_b32(rbits, (id >> 4) * 4u, 0, 0) : 0u;   // bits 2 id, 2 id + 1 of the chain's record
    const u32 two = (word >> ((id & 15u) * 2u)) & 3u;
    const u32 f = (ij >> 16) & 1u ? (((two >> 1) & 1u) | ((two & 1u) << 1)) : two;   // bit 0: L[i] -> L[j], bit 1: L[j] -> L[i]
    u64 h = 0ull;
    for (int t = 0; t < T; ++t) {       // (wave-uniform; one trip for k = 0)
        const u32 e = rdlane(ij, t), ft = rdlane(f, t);
        const u32 i = e & 0xFFu, j = (e >> 8) & 0xFFu;
        h |= (lane == (int)j ? (u64)(ft & 1u) << i : 0ull) | (lane == (int)i ? (u64)(ft >> 1) << j : 0ull);
    }
    return h;
}

// The wave's tallies, u32 words in LDS (added to the chain's stats row at the end): counters (the first four are the low
// bits of the staged flags word), then one flag per count entry -- was it ever non-zero after a transition (flag_count
// never shrinks in length, src/lib.rs:72-74) -- and the OR of the proposals' status words.  In LDS rather than in a
// dozen scalars that would live, spilled, across the whole loop.
enum { MA_NONEMPTY = 0, MA_DMOVE = 1, MA_WIDE = 2, MA_BIG = 3, MA_MINE = 4, MA_ACCEPTED = 5, MA_REDO = 6, MA_SUMK = 7, MA_NZ0 = 8, MA_STATUS = 16, MA_WORDS = 17 };

// both endpoints of the pair (big, small) in the local list Lv?  (lanes beyond the list repeat its last vertex)
__device__ __forceinline__ bool mw_inside(u32 Lv, u32 big, u32 small) { return ballot(Lv == big) != 0ull && ballot(Lv == small) != 0ull; }

// LDS stores of the lanes in a literal mask, with no predicate to compute or keep.  Only where all 64 lanes are active
// (the top level of the wave loop: its branches are wave-uniform).  LDS executes a wave's operations in order, so a
// later store of the same wave (the head) is seen by nobody before these are.
#define MW_LDS_ST32(addr, val, MASK) asm volatile("s_mov_b64 exec, " MASK "\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(addr), "v"(val) : "memory")
#define MW_LDS_ST64(addr, val, MASK) asm volatile("s_mov_b64 exec, " MASK "\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(addr), "v"(val) : "memory")
__device__ __forceinline__ u32 mw_lds_addr(const void *p) { return (u32)(size_t)(__attribute__((address_space(3))) const char *)p; }

// min over the waves of vis[]: every proposal below it is visible to what this wave loads from now on (acquire)
__device__ __forceinline__ u32 mw_vis_min(const u32 *vis, u32 W, int lane)
{
    if (W == 2u) {   // (two entries: one 64-bit read and a min -- no reduction over the lanes)
        const u64 v2 = __hip_atomic_load((const u64 *)vis, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return mw_uni(min((u32)v2, (u32)(v2 >> 32)));
    }
    u32 v = __hip_atomic_load(&vis[lane & 15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // every row of 16 lanes reads all of it (entries >= W hold MW_NONE)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)MW_NONE, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)MW_NONE, (int)v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)MW_NONE, (int)v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)MW_NONE, (int)v, 0x118, 0xf, 0xf, false));  // row_shr:8
    return rdlane(v, 15);
}

struct MwProp;
// wave-uniform view of the chain
struct MwChain {
    u32 *rows, *dbl;
    const u32 *nb;
    const FcmEdgeEntry *etab;
    u64 rows_bytes;
    u64 *xw;            // workspace of the evaluator for 257..1024 local vertices, or null
    u32 U, D, stride32;
};
__device__ __forceinline__ MwChain mw_chain_from_lds(const u32 *ctx, int lane)
{
    const u32 cv = lane < MC_WORDS ? ctx[lane] : 0u;
    MwChain C;
    C.rows = (u32 *)((u64)rdlane(cv, MC_ROWS) | ((u64)rdlane(cv, MC_ROWS + 1) << 32));
    C.dbl = (u32 *)((u64)rdlane(cv, MC_DBL) | ((u64)rdlane(cv, MC_DBL + 1) << 32));
    C.nb = (const u32 *)((u64)rdlane(cv, MC_NB) | ((u64)rdlane(cv, MC_NB + 1) << 32));
    C.etab = (const FcmEdgeEntry *)((u64)rdlane(cv, MC_ETAB) | ((u64)rdlane(cv, MC_ETAB + 1) << 32));
    C.rows_bytes = (u64)rdlane(cv, MC_ROWS_BYTES) | ((u64)rdlane(cv, MC_ROWS_BYTES + 1) << 32);
    C.xw = (u64 *)((u64)rdlane(cv, MC_XW) | ((u64)rdlane(cv, MC_XW + 1) << 32));
    C.U = rdlane(cv, MC_U); C.D = rdlane(cv, MC_D); C.stride32 = rdlane(cv, MC_STRIDE32);
    return C;
}

// What a run of a proposal leaves behind.  The wave-uniform part (MwRec) is written to the proposal's record in the ring
// *before* the evaluations and read back after them (SR_* words; words 0..9 are what the other waves hold against their
// reads, SR_STATE says whose record it is and whether it is staged or decided): kept in SGPRs it would be live across
// the evaluations, where at 80 SGPRs it is spilled and reloaded.
struct MwRec {
    u32 nonempty, is_dmove, used_wide, big_set;   // 0 / 1
    u32 wid_clr, wid_set, bit_clr, bit_set, dslot, dnew, add_k;
    u32 id1, big1, small1, id2, big2, small2, cx0, cx1, sus;
};
enum { SR_FLAGS = 0, SR_BIG1, SR_SMALL1, SR_ID1, SR_BIG2, SR_SMALL2, SR_ID2, SR_DSLOT, SR_WCLR, SR_WSET, SR_CX0, SR_CX1, SR_SUS, SR_STATE, SR_WORDS };
static_assert(SR_WORDS == SR_WORDS_C && SR_STATE == SR_SUS + 1 && SR_SUS % 2 == 0, "record layout");
// SR_FLAGS: nonempty<<0 | dmove<<1 | used_wide<<2 | big_set<<3 | add_k<<8 (12 bits) | clr bit index<<20 | set bit index<<25
#define SRF_NONEMPTY 1u
#define SRF_DMOVE 2u
#define SRF_WIDE 4u
#define SRF_BIG 8u
struct MwOut {
    u32 need_exact, snap;   // wave-uniform
    u32 Lv1, Lv2;           // per lane: the local vertex lists the run read
    u32 w_clr, w_set;       // per lane (all lanes alike): the two bitmap words the commit rewrites, as read beside the builds
    int myd;                // per lane: lane d holds the change of count[d] (32 bits: fcm_lane_guard / the wide evaluator's own check)
};
// (the state word last: LDS takes the stores in order, a reader that finds the state finds the record)
__device__ __forceinline__ void mw_stage(u32 *stage, const MwRec &R, int lane, u32 state)
{
    if (lane == 0) {
        const u32 fl = R.nonempty | (R.is_dmove << 1) | (R.used_wide << 2) | (R.big_set << 3) | ((R.add_k & 0xFFFu) << 8)
                       | (R.bit_clr << 20) | (R.bit_set << 25);   // (bit indices, 0..31)
        // (64-bit stores: the ring starts at an 8-byte boundary and a record is 56 bytes, so a record is 8-byte aligned only;
        //  hipcc pairs them into ds_write2_b64, which asks for no more)
        uint2 *st2 = (uint2 *)stage;
        st2[0] = make_uint2(fl, R.big1); st2[1] = make_uint2(R.small1, R.id1);
        st2[2] = make_uint2(R.big2, R.small2); st2[3] = make_uint2(R.id2, R.dslot);
        st2[4] = make_uint2(R.wid_clr, R.wid_set); st2[5] = make_uint2(R.cx0, R.cx1);
        st2[6] = make_uint2(R.sus, state);
    }
    wave_sync();
}

// One proposal on the bitmap as it is (reference Transition::random_move + State::apply_transition,
// src/lib.rs:207-212, 292-325, 61-79).  tv = the proposal's draw-table entry (lane i = word i).
// EXACT = false: the hot path -- two single-edge candidates, local sets of <= 64 vertices whose split graph fits; anything
// else sets need_exact and is left to the exact run.  EXACT = true: the whole of it (candidate search to the end, wide
// evaluator), on a state nobody else changes meanwhile.
template <int MAXT, bool ROWS128, bool EXACT, bool SPARSE = false>
__device__ __forceinline__ void mw_run(const MwChain &C, u64 *Hp, u64 *wide_lds, int maxnw, u32 tv, u64 tt, u32 gchain, u64 seed, int lane, MwOut &O,
                                       u32 *stage, u32 rec_q, u32 *vis = nullptr, u32 wv = 0u, u32 q = 0u, u32 W = 0u, u64 guard_limit = 0x7FFFFFFFull)
{
    MwRec R;
    const int tmax = MAXT;
    // The snap point (hot path only): between the static loads of the proposal (table entry, vertex lists) and its first
    // load of mutable state.  The wave's earlier commit stores have completed by then (s_waitcnt vmcnt(0), which the
    // vertex list needs anyway): publish that, and note from where on decisions have to be held against this proposal.
    bool snapped = false;
    auto snap_point = [&]() {
        if constexpr (!EXACT) {
            if (!snapped) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(&vis[wv], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                O.snap = mw_vis_min(vis, W, lane);
                snapped = true;
            }
        }
    };
    R.nonempty = R.is_dmove = R.used_wide = R.big_set = O.need_exact = 0u;
    R.wid_clr = R.wid_set = MW_NONE; R.bit_clr = R.bit_set = 0u; O.w_clr = O.w_set = 0u; R.dslot = MW_NONE; R.dnew = 0u; R.add_k = 0u;
    R.id1 = R.big1 = R.small1 = R.id2 = R.big2 = R.small2 = R.cx0 = R.cx1 = MW_NONE; R.sus = 0u;
    O.Lv1 = MW_NONE; O.Lv2 = MW_NONE;
    O.myd = 0; O.snap = 0u;
    FcmGuard guard = {guard_limit, 0u};
    const int move = (int)(rdlane(tv, 0) & 0xFFu);
    const u32 coin = (rdlane(tv, 0) >> 8) & 1u;
    const u64 idx = (u64)rdlane(tv, 2) | ((u64)rdlane(tv, 3) << 32);
    const u32 U = C.U, D = C.D, stride32 = ROWS128 ? 32u : C.stride32;   // (cache-line rows: a constant, word ids by shifts)
    const rsrc_t rr = make_rows_rsrc(C.rows, C.rows_bytes);
    const u32 *const nbk = EXACT ? C.nb : (const u32 *)fcm_karg64<offsetof(FcmStepParams, nb)>();   // (the hot path: re-read, see fcm_karg64)
    const u32 *const dblk = C.dbl;
    const FcmEdgeEntry *const etabk = C.etab;
    const u64 Mtot = (u64)U + D;

    // what is to be evaluated on the fast path: two (masks, classes, size), signs -1 and +1
    int nev = 0;
    u64 HA = 0ull, HB = 0ull;
    Cls cA = {0ull, 0ull, 0ull}, cB = {0ull, 0ull, 0ull};
    int kA = 0, kB = 0;
    bool go_wide = false;

    if (move == 0) {
        // ---- single_edge_flip (src/lib.rs:292-299)
        if (Mtot > 0 && idx < U) {
            const FcmEdgeEntry e1 = {rdlane(tv, 4), rdlane(tv, 5), rdlane(tv, 6), rdlane(tv, 7)};
            const int k = (int)e1.k;
            int fres = 0;          // 1 = big->small is flipped, 2 = small->big
            R.id1 = (u32)idx; R.big1 = e1.big; R.small1 = e1.small;
            // the words (and bits) of big->small, small->big: of the rows, or of the pair's two bits in the chain's sparse record
            const u32 wid_bs = SPARSE ? (u32)idx >> 4 : e1.big * stride32 + (e1.small >> 5), wid_sb = SPARSE ? wid_bs : e1.small * stride32 + (e1.big >> 5);
            const u32 bit_bs = SPARSE ? ((u32)idx & 15u) * 2u : e1.small & 31u, bit_sb = SPARSE ? bit_bs + 1u : e1.big & 31u;
            if (k + 2 <= WAVE) {
                O.Lv1 = load_list(nbk, e1.nb_off, k, e1.big, e1.small, lane);
                snap_point();
#if MW_PROBE == 1   // (instruction-cost probe, tools/probe_costs.sh: a flip builds its masks twice)
                u64 myH = mw_build<ROWS128>(rr, stride32, O.Lv1, k + 2, lane);
                { u32 z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); u32 Lz = O.Lv1; asm volatile("" : "+v"(Lz)); myH |= mw_build<ROWS128>(rr, stride32, Lz, k + 2, lane) & (u64)z; }
#else
                const u64 myH = SPARSE ? mw_build_sparse(rr, nbk, e1.nb_off, k, (u32)idx, lane) : mw_build<ROWS128>(rr, stride32, O.Lv1, k + 2, lane);
#endif
                const u64 hk = rdlane64(myH, k), hk1 = rdlane64(myH, k + 1);
                const u32 ab = (u32)((hk1 >> k) & 1ull), ba = (u32)((hk >> (k + 1)) & 1ull);  // big->small, small->big
                if (ab == ba) {
                    if (!ab) R.sus |= 1u;   // table says adjacent, bitmap says not
                } else {
                    const int iu = ab ? k : k + 1, iv = ab ? k + 1 : k;
                    cA = classify(myH, iv, iu);
                    cB.P = cA.P; cB.S = cA.S;   // after the flip P and S are the same sets, M becomes {v->w, w->u}
                    cB.M = (ab ? hk : hk1) & ballot((myH >> iv) & 1ull) & ~(3ull << k);
                    fres = ab ? 1 : 2;
                    // a flip's two evaluations as one (eval_flip_merged: the P*S* cliques, which cancel, are never walked); if its
                    // nodes do not fit 64, the two evaluations one after the other; if theirs do not either, the wide evaluator
                    if (MW_MERGED && flip_merged_fits(cA.P, cA.M, cB.M, cA.S, k)) { HA = HB = myH; kA = kB = k; nev = 3; }
                    else if (extras_fit(cA, k + 2) && extras_fit(cB, k + 2)) { HA = HB = myH; kA = kB = k; nev = 2; }
                    else go_wide = true;
                }
            } else {
                go_wide = true;   // direction unknown yet: the wide run finds it
            }
            if (go_wide) {
                if constexpr (!EXACT) {
                    O.need_exact = 1u;
                } else if constexpr (SPARSE) {
                    fres = 0; R.sus |= 1u;   // (the sparse layout is only selected for graphs whose local sets always fit the fast evaluator)
                } else {
                    fres = 0;
                    if (k + 2 > 64 * maxnw && C.xw && k + 2 <= 64 * FCM_XW_MAXNW) {   // 257..1024 local vertices
                        const int res = xw_flip(C.xw, C.rows, stride32, nbk, e1.nb_off, k, e1.big, e1.small, lane, tmax);
                        { const long long wc = (lane >= 2 && lane < 16 && lane - 1 <= tmax) ? xw_count(C.xw, lane - 1) : 0ll; O.myd = (int)wc; if (ballot(wc != (long long)(int)wc)) R.sus |= 256u; }
                        wave_sync();
                        R.used_wide = 1u;
                        if (res < 0) R.sus |= 1u;
                        fres = res > 0 ? res : 0;
                    } else if (k + 2 <= 64 * maxnw) {
                        const Wide Wd = wide_carve(wide_lds, maxnw);
                        wide_zero_counts(Wd, lane);
                        const int res = wide_flip(Wd, C.rows, stride32, nbk, e1.nb_off, k, e1.big, e1.small, lane, tmax);
                        { const long long wc = (lane >= 2 && lane < 16 && lane - 1 <= tmax) ? Wd.cnt[lane - 1] : 0ll; O.myd = (int)wc; if (ballot(wc != (long long)(int)wc)) R.sus |= 256u; }
                        wave_sync();
                        R.used_wide = 1u;
                        if (res < 0) R.sus |= 1u;
                        fres = res > 0 ? res : 0;
                    } else {
                        R.sus |= 1u;
                    }
                }
            }
            if (fres > 0 && !O.need_exact) {
                R.nonempty = 1u;
                R.wid_clr = fres == 1 ? wid_bs : wid_sb; R.bit_clr = fres == 1 ? bit_bs : bit_sb;
                R.wid_set = fres == 1 ? wid_sb : wid_bs; R.bit_set = fres == 1 ? bit_sb : bit_bs;
                R.add_k = (u32)k; R.big_set = k + 2 > 48 ? 1u : 0u;
            }
        }
    } else if (move == 1 && D > 0) {
        // ---- double_edge_move (src/lib.rs:304-325)
        R.dslot = (u32)idx;
        FcmEdgeEntry e1 = {rdlane(tv, 8), rdlane(tv, 9), rdlane(tv, 10), rdlane(tv, 11)};            // the table's guess of the slot's pair
        FcmEdgeEntry e2 = {rdlane(tv, 4), rdlane(tv, 5), rdlane(tv, 6), rdlane(tv, 7)};              // candidate 0's entry
        // static loads first: the list of candidate 0
        const u32 c0v = rdlane(tv, 12);
        const bool c0ok = c0v != MW_NONE && (int)e2.k + 2 <= WAVE;
        u32 Lv2pre = MW_NONE;
        if (c0ok) Lv2pre = load_list(nbk, e2.nb_off, (int)e2.k, e2.big, e2.small, lane);
        // ... and the list of the table's guess of the slot's pair (static as well; thrown away if the slot was rewritten)
        u32 Lv1pre = MW_NONE;
        if ((int)e1.k + 2 <= WAVE) Lv1pre = load_list(nbk, e1.nb_off, (int)e1.k, e1.big, e1.small, lane);
        snap_point();
        // mutable state from here on: the slot's live entry first -- requested here, looked at after candidate 0's build, so
        // that its round trip runs beside that build's instead of in front of it (round 4: + 0.6 % with the list above)
        const u32 ed_v = dblk[R.dslot];
        // single-edge candidates (:308-313): candidates 0 and 1 come from the table; a longer search, or a
        // candidate that needs the wide path, is left to the exact run
        u64 cand = 0ull, cand_next = 0ull;
        bool found = false;
        u32 rfwd = 0u;
#pragma nounroll
        for (int ci = 0; ci < (EXACT ? WAVE : 3) && !found && !O.need_exact; ++ci) {
            if (ci < 2) {
                const u32 cv = rdlane(tv, 12 + ci);
                cand = cv == MW_NONE ? ~0ull : (u64)cv;
                if (ci == 0) R.cx0 = cv; else R.cx1 = cv;
            } else if constexpr (!EXACT) {
                O.need_exact = 1u;
                break;
            } else if ((ci & 1) == 0) {  // Philox block sub = ci/2 + 1: two candidates
                u32 v[4];
                philox4x32_10((u32)tt, (u32)(tt >> 32), gchain, (u32)(ci >> 1) + 1u, (u32)seed, (u32)(seed >> 32), v);
                cand = __umul64hi((u64)v[0] | ((u64)v[1] << 32), Mtot);
                cand_next = __umul64hi((u64)v[2] | ((u64)v[3] << 32), Mtot);
            } else {
                cand = cand_next;
            }
            if (cand < U) {
                if (ci > 0) {
                    const FcmEdgeEntry t = etabk[cand];
                    e2 = FcmEdgeEntry{mw_uni(t.big), mw_uni(t.small), mw_uni(t.nb_off), mw_uni(t.k)};
                }
                const int ck = (int)e2.k;
                u32 f = 0u, bwd = 0u;
                if (ck + 2 <= WAVE) {
                    if (ci == 0) {
                        O.Lv2 = Lv2pre;
                    } else {
                        O.Lv2 = load_list(nbk, e2.nb_off, ck, e2.big, e2.small, lane);
                    }
                    HB = SPARSE ? mw_build_sparse(rr, nbk, e2.nb_off, ck, (u32)cand, lane) : mw_build<ROWS128>(rr, stride32, O.Lv2, ck + 2, lane);
                    f = (u32)(rdlane64(HB, ck + 1) >> ck) & 1u;
                    bwd = (u32)(rdlane64(HB, ck) >> (ck + 1)) & 1u;
                } else {
                    if constexpr (!EXACT) {
                        O.need_exact = 1u;
                        break;
                    } else if constexpr (SPARSE) {
                        R.sus |= 1u;
                    } else {  // wide candidate: look at its two words directly
                        const u32 wf = mw_uni(C.rows[e2.big * stride32 + (e2.small >> 5)]), wb = mw_uni(C.rows[e2.small * stride32 + (e2.big >> 5)]);
                        f = (wf >> (e2.small & 31u)) & 1u;
                        bwd = (wb >> (e2.big & 31u)) & 1u;
                        O.Lv2 = MW_NONE;
                    }
                }
                if (!(f | bwd)) R.sus |= 1u;
                found = (f ^ bwd) != 0u;
                rfwd = f;
            }
        }
        const u32 ed = mw_uni(ed_v);
        if (ed != rdlane(tv, 1)) {   // the slot was rewritten since the table was filled (rare)
            const FcmEdgeEntry t = etabk[ed];
            e1 = FcmEdgeEntry{mw_uni(t.big), mw_uni(t.small), mw_uni(t.nb_off), mw_uni(t.k)};
        }
        R.id1 = ed; R.big1 = e1.big; R.small1 = e1.small;
        if (found) {
            const int dk = (int)e1.k, rk = (int)e2.k;
            R.id2 = (u32)cand; R.big2 = e2.big; R.small2 = e2.small;
            const u32 ea = rfwd ? e2.big : e2.small, eb = rfwd ? e2.small : e2.big;  // ea->eb is the single edge
            const u32 dfrom = coin ? e1.big : e1.small, dto = coin ? e1.small : e1.big;  // delme (:316-320)
            go_wide = dk + 2 > WAVE || rk + 2 > WAVE;
            bool okd = true;
            if (!go_wide) {
                if (ed == rdlane(tv, 1)) O.Lv1 = Lv1pre;
                else O.Lv1 = load_list(nbk, e1.nb_off, dk, e1.big, e1.small, lane);
                HA = SPARSE ? mw_build_sparse(rr, nbk, e1.nb_off, dk, ed, lane) : mw_build<ROWS128>(rr, stride32, O.Lv1, dk + 2, lane);
                // (1) remove the direction the coin picks from the reciprocal pair
                const u32 ab = (u32)((rdlane64(HA, dk + 1) >> dk) & 1ull), ba = (u32)((rdlane64(HA, dk) >> (dk + 1)) & 1ull);
                okd = (ab & ba) != 0u;
                const int iu = coin ? dk : dk + 1, iv = coin ? dk + 1 : dk;
                cA = classify(HA, iv, iu);
                // (2) add the reverse of the single edge on the graph without the removed one
                const u64 mf = ballot(lane < rk + 2 && O.Lv2 == dfrom), mt = ballot(lane < rk + 2 && O.Lv2 == dto);
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
                if constexpr (!EXACT) {
                    O.need_exact = 1u;
                } else if constexpr (SPARSE) {
                    R.sus |= 1u;
                } else if (dk + 2 > 64 * maxnw || rk + 2 > 64 * maxnw) {
                    if (C.xw && dk + 2 <= 64 * FCM_XW_MAXNW && rk + 2 <= 64 * FCM_XW_MAXNW) {
                        okd = xw_del(C.xw, C.rows, stride32, nbk, e1.nb_off, dk, e1.big, e1.small, coin, lane, tmax);
                        xw_add(C.xw, C.rows, stride32, nbk, e2.nb_off, rk, e2.big, e2.small, rfwd, dfrom, dto, lane, tmax, true);
                        { const long long wc = (lane >= 2 && lane < 16 && lane - 1 <= tmax) ? xw_count(C.xw, lane - 1) : 0ll; O.myd = (int)wc; if (ballot(wc != (long long)(int)wc)) R.sus |= 256u; }
                        wave_sync();
                        R.used_wide = 1u;
                    } else {
                        R.sus |= 1u;
                    }
                } else {
                    const Wide Wd = wide_carve(wide_lds, maxnw);
                    wide_zero_counts(Wd, lane);
                    okd = wide_del(Wd, C.rows, stride32, nbk, e1.nb_off, dk, e1.big, e1.small, coin, lane, tmax);
                    wide_add(Wd, C.rows, stride32, nbk, e2.nb_off, rk, e2.big, e2.small, rfwd, dfrom, dto, lane, tmax);
                    { const long long wc = (lane >= 2 && lane < 16 && lane - 1 <= tmax) ? Wd.cnt[lane - 1] : 0ll; O.myd = (int)wc; if (ballot(wc != (long long)(int)wc)) R.sus |= 256u; }
                    wave_sync();
                    R.used_wide = 1u;
                }
            }
            if (!okd) R.sus |= 2u;  // slot list says reciprocal, bitmap says not
            R.nonempty = 1u; R.is_dmove = 1u;
            if constexpr (SPARSE) {   // bit 2e: big -> small, 2e + 1: small -> big.  Deleted: the coin's direction of the reciprocal pair; added: the reverse of the single edge
                const u32 bc = 2u * ed + (coin ? 0u : 1u), bs = 2u * (u32)cand + (rfwd ? 1u : 0u);
                R.wid_clr = bc >> 5; R.bit_clr = bc & 31u;
                R.wid_set = bs >> 5; R.bit_set = bs & 31u;
            } else {
                R.wid_clr = dfrom * stride32 + (dto >> 5); R.bit_clr = dto & 31u;
                R.wid_set = eb * stride32 + (ea >> 5); R.bit_set = ea & 31u;
            }
            R.dnew = (u32)cand;
            R.add_k = (u32)(dk + rk); R.big_set = (dk + 2 > 48 || rk + 2 > 48) ? 1u : 0u;
        }
    }
    snap_point();   // (paths without a load of mutable state)
    // the two bitmap words a commit rewrites, read here -- lines the builds have just touched -- so that the commit is
    // two plain stores.  Issued last: the round trip runs beside the wait for the token instead of in front of it.
    if (R.nonempty && !(MW_ABL & 8)) {   // (through the descriptor: the word id goes into the scalar offset, no 64-bit address arithmetic)
        O.w_clr = __builtin_amdgcn_raw_buffer_load_b32(rr, 0, (int)(R.wid_clr * 4u), 0);
        O.w_set = __builtin_amdgcn_raw_buffer_load_b32(rr, 0, (int)(R.wid_set * 4u), 0);
    }
    R.dnew = R.id2;
    // the record, before the evaluations: nothing of it stays in registers across them, and the chain's other waves can
    // hold it against their reads from now on (not if it is incomplete: they then wait for the exact run's)
    mw_stage(stage, R, lane, (rec_q << 4) | ((EXACT || O.need_exact) ? 0u : MS_STAGED));
#if MW_ABL & 1
    nev = 0;
#endif
    if (nev) {
        int delta[MAXT + 1];
#pragma unroll
#if MW_K_ZERO
        for (int t = 0; t <= MAXT; ++t) asm volatile("v_mov_b32 %0, 0" : "=v"(delta[t]));   // (zeroed by an instruction of its own: hipcc otherwise may
                                                                                             //  keep a zero tuple alive across the whole loop -- in scratch)
#else
        for (int t = 0; t <= MAXT; ++t) delta[t] = 0;
#endif
        EvScal es = {0, 0};
        bool merged_done = false;
        if (MW_MERGED && nev == 3) merged_done = eval_flip_merged<MAXT>(HA, Hp, cA.P, cA.M, cB.M, cA.S, kA, lane, delta, es);
        if (!merged_done) {
#if MW_K_EVLOOP
#pragma nounroll
#else
#pragma unroll
#endif
            for (int ev = 0; ev < 2; ++ev) {
                const Cls c = ev ? cB : cA;
                eval_nodes<MAXT>(ev ? HB : HA, Hp, c, ev ? kB : kA, tmax, ev ? +1 : -1, lane, delta, es, nullptr, nullptr, &guard);
            }
        }
#if MW_PROBE == 2   // (instruction-cost probe: a third evaluation whose sign is an opaque 0)
        { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); eval_nodes<MAXT>(HA, Hp, cA, kA, tmax, z, lane, delta, es, nullptr, nullptr, &guard); }
#endif
        fcm_lane_guard<MAXT>(delta, guard);
        O.myd = lane_in(4ull) ? es.d1 : O.myd;   // levels 1 and 2: node and arc counts of the two split graphs (scalars)
        if (MAXT >= 2) O.myd = lane_in(8ull) ? es.d2 : O.myd;
        // the deeper levels: wave sums by DPP adds, the levels' chains interleaved step by step (no wait states between
        // the dependent steps of one chain)
#define MW_SUM_STEP(ctrl, rmask) _Pragma("unroll") for (int tq = 3; tq <= MAXT; ++tq) delta[tq] += __builtin_amdgcn_update_dpp(0, delta[tq], ctrl, rmask, 0xf, false)
        MW_SUM_STEP(0x111, 0xf); MW_SUM_STEP(0x112, 0xf); MW_SUM_STEP(0x114, 0xf); MW_SUM_STEP(0x118, 0xf);
        MW_SUM_STEP(0x142, 0xa); MW_SUM_STEP(0x143, 0xc);
#undef MW_SUM_STEP
#pragma unroll
        for (int tq = 3; tq <= MAXT; ++tq) {
            const int sum = __builtin_amdgcn_readlane(delta[tq], 63);
            O.myd = lane_in(1ull << (tq + 1)) ? sum : O.myd;
        }
        if (guard.tripped) { if (lane == 0) stage[SR_SUS] |= 256u; wave_sync(); }   // a local count may have passed 2^31: refuse rather than wrap
    }
}

// ---- out of line: the exact run.  Context from LDS; the record goes to the staging words like the hot path's, the
// per-lane results (count changes, the two bitmap words) to the arc list's place.
template <int MAXT, bool ROWS128, bool SPARSE = false>
__device__ __attribute__((noinline)) void mw_exact_call(u64 *smem, u32 wv, u32 tv, u32 q, u32 stage_off)
{
    const int lane = threadIdx.x & (WAVE - 1);
    wv = mw_uni(wv); q = mw_uni(q); stage_off = mw_uni(stage_off);
    const u32 *ctx = (const u32 *)(smem + MW_CTX_OFF);
    const u32 cv = lane < MC_WORDS ? ctx[lane] : 0u;
    const u32 W = rdlane(cv, MC_W);
    const int maxnw = (int)rdlane(cv, MC_MAXNW);
    const MwChain C = mw_chain_from_lds(ctx, lane);
    u64 *mine_lds = smem + MW_SHARED_WORDS + MW_RING_WORDS(W) + (size_t)wv * MW_WAVE_WORDS;
    u64 *wide_lds = smem + MW_SHARED_WORDS + MW_RING_WORDS(W) + (size_t)W * MW_WAVE_WORDS;
    const u64 seed = (u64)rdlane(cv, MC_SEED) | ((u64)rdlane(cv, MC_SEED + 1) << 32);
    const u64 sampled0 = (u64)rdlane(cv, MC_SAMPLED0) | ((u64)rdlane(cv, MC_SAMPLED0 + 1) << 32);
    MwOut O;
    mw_run<MAXT, ROWS128, true, SPARSE>(C, mine_lds, wide_lds, maxnw, tv, sampled0 + q, rdlane(cv, MC_GCHAIN), seed, lane, O, (u32 *)smem + stage_off, q, nullptr, 0u, 0u, 0u,
                                (u64)rdlane(cv, MC_GUARD) | ((u64)rdlane(cv, MC_GUARD + 1) << 32));
    u32 *out = (u32 *)(mine_lds + 64);
    if (lane < 16) out[lane] = (u32)O.myd;
    if (lane == 16) out[16] = O.w_clr;
    if (lane == 17) out[17] = O.w_set;
    wave_sync();
}

// does the record `ev` (lane i = word i; not an empty transition), if accepted, change anything a proposal read?  The
// proposal: its local vertex lists Lv1, Lv2 (per lane), the candidate pairs it looked at, the slot and the two bitmap
// words it rewrites.
__device__ __forceinline__ bool mw_touches(u32 ev, u32 Lv1, u32 Lv2, u32 cx0, u32 cx1, u32 dslot, u32 wid_clr, u32 wid_set)
{
    const u32 fl = rdlane(ev, SR_FLAGS);
    const u32 b1 = rdlane(ev, SR_BIG1), s1 = rdlane(ev, SR_SMALL1), i1 = rdlane(ev, SR_ID1);
    const u32 wc = rdlane(ev, SR_WCLR), ws = rdlane(ev, SR_WSET);
    bool t = mw_inside(Lv1, b1, s1) || mw_inside(Lv2, b1, s1) || i1 == cx0 || i1 == cx1
             || wc == wid_clr || wc == wid_set || ws == wid_clr || ws == wid_set;
    if (fl & SRF_DMOVE) {
        const u32 b2 = rdlane(ev, SR_BIG2), s2 = rdlane(ev, SR_SMALL2), i2 = rdlane(ev, SR_ID2);
        t = t || mw_inside(Lv1, b2, s2) || mw_inside(Lv2, b2, s2) || i2 == cx0 || i2 == cx1 || rdlane(ev, SR_DSLOT) == dslot;
    }
    return t;
}

// ---- out of line: the wait for the token with several records to check (W >= 4).  All records snap..q-1 at once, one
// per lane, and as soon as they are staged: by the time head == q nothing of that is left.  Whether a vertex is in one
// of the proposal's local sets is looked up in two maps of 2048 bits (vertex id mod 2048: exact up to 2048 vertices,
// a superset beyond), kept where the split graph was; a lane whose record looks like a conflict there is then looked
// at exactly (mw_touches).  A conflict with a record that is only staged is looked at again when that proposal is
// decided (it may be dropped).  Under the token: records checked as staged and then decided on an exact run may have
// changed -- those are checked again.  Returns with the token held: 1 = the proposal has to be run again.
__device__ __attribute__((noinline)) u32 mw_wait_staged(u64 *smem, u32 wv, u32 q, u32 snap, u32 Lv1, u32 Lv2, u32 hit0, u32 W, u32 sv)
{
    const int lane = threadIdx.x & (WAVE - 1);
    wv = mw_uni(wv); q = mw_uni(q); snap = mw_uni(snap);
    bool hit = mw_uni(hit0) != 0u;
    W = mw_uni(W);                                                        // (sv: the proposal's own record, lane i = word i)
    const u32 ring = 4u * W - 1u;
    const u32 *ringL = (const u32 *)(smem + MW_SHARED_WORDS);
    const u32 *ctl = (const u32 *)(smem + MW_HEAD_OFF);
    u64 *Hp = smem + MW_SHARED_WORDS + MW_RING_WORDS(W) + (size_t)wv * MW_WAVE_WORDS;
    u32 *tally = (u32 *)(smem + MW_TALLY_OFF);
    u64 heldm = 0ull;                                                     // (non-zero: waited for the decision of a staged record in conflict)
    const u32 cx0 = rdlane(sv, SR_CX0), cx1 = rdlane(sv, SR_CX1), dslot = rdlane(sv, SR_DSLOT);
    const u32 wid_clr = rdlane(sv, SR_WCLR), wid_set = rdlane(sv, SR_WSET);
    auto touches = [&](u32 ev) -> bool { return (rdlane(ev, SR_FLAGS) & SRF_NONEMPTY) ? mw_touches(ev, Lv1, Lv2, cx0, cx1, dslot, wid_clr, wid_set) : false; };
    const u32 nent = q - snap;                                            // 1 .. 2W - 1 <= 31
    const u64 all = ~0ull >> (64u - nent);
    u64 done = 0ull;                                                      // records that are no conflict (if only staged: as staged)
    u32 *bmA = (u32 *)Hp, *bmB = bmA + 64;
    if (!hit) {
        Hp[lane] = 0ull;
        if (Lv1 != MW_NONE) __hip_atomic_fetch_or(&bmA[(Lv1 >> 5) & 63u], 1u << (Lv1 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (Lv2 != MW_NONE) __hip_atomic_fetch_or(&bmB[(Lv2 >> 5) & 63u], 1u << (Lv2 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        wave_sync();
    }
    const u32 pidx = snap + (u32)lane;
    const u32 *e = ringL + (pidx & ring) * MW_REC_WORDS;                  // this lane's record (lanes >= nent: some record of the ring, masked below)
    const u32 a_head = mw_lds_addr(ctl), a_state = mw_lds_addr(e + SR_STATE);
    u32 st, hv, hh;
    auto doze = [&](u32 dist) {   // by how far off the token is (a decision takes several hundred cycles)
        if (dist >= 4u) __builtin_amdgcn_s_sleep(MW_SLEEP_FAR);
        else if (dist >= 2u) { if (MW_SLEEP_MID) __builtin_amdgcn_s_sleep(MW_SLEEP_MID); }
        else if (MW_SPIN_NEAR == 0) __builtin_amdgcn_s_sleep(1);
    };
    for (;;) {
        // head, then the state words -- one round trip, in this order: with head == q every state word read is final
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(hv), "=&v"(st) : "v"(a_head), "v"(a_state) : "memory");
        const u32 h = mw_uni(hv);                                         // h <= q
        if (!hit && done != all) {
            const u32 fl = e[SR_FLAGS];
            const u32 b1 = e[SR_BIG1], s1 = e[SR_SMALL1], i1 = e[SR_ID1], wc = e[SR_WCLR], ws = e[SR_WSET];
            const u32 b2 = e[SR_BIG2], s2 = e[SR_SMALL2], i2 = e[SR_ID2], ds = e[SR_DSLOT];
            auto in_map = [](const u32 *bm, u32 v) -> u32 { return (bm[(v >> 5) & 63u] >> (v & 31u)) & 1u; };
            u32 cf = (in_map(bmA, b1) & in_map(bmA, s1)) | (in_map(bmB, b1) & in_map(bmB, s1));
            cf |= (i1 == cx0 || i1 == cx1 || wc == wid_clr || wc == wid_set || ws == wid_clr || ws == wid_set) ? 1u : 0u;
            u32 cf2 = (in_map(bmA, b2) & in_map(bmA, s2)) | (in_map(bmB, b2) & in_map(bmB, s2));
            cf2 |= (i2 == cx0 || i2 == cx1 || ds == dslot) ? 1u : 0u;
            cf |= (fl & SRF_DMOVE) ? cf2 : 0u;
            cf &= fl;                                                     // SRF_NONEMPTY is bit 0: an empty transition changes nothing
            const u64 there = ballot((st >> 4) == pidx && (st & 3u) != 0u) & all & ~done;
            const u64 dropped = ballot((st & (MS_DECIDED | MS_ACCEPTED)) == MS_DECIDED);
            const u64 maybe = ballot(cf != 0u);
            done |= there & (dropped | ~maybe);
            u64 look = there & maybe & ~dropped;
            while (look && !hit) {
                const u32 i = (u32)__ffsll((long long)look) - 1u;
                look &= look - 1ull;
                const u32 ev = lane < SR_WORDS ? ringL[((snap + i) & ring) * MW_REC_WORDS + lane] : 0u;
                const u32 sti = rdlane(ev, SR_STATE);
                if ((sti >> 4) != snap + i || (sti & 3u) == 0u) continue;           // being written again: later
                if ((sti & (MS_DECIDED | MS_ACCEPTED)) == MS_DECIDED) { done |= 1ull << i; continue; }
                if (!touches(ev)) done |= 1ull << i;
                else if (sti & MS_DECIDED) hit = true;
                else heldm = 1ull;                                                  // only staged: it may yet be dropped -- wait
            }
        }
        if (hit || done == all) { hh = h; break; }   // nothing left to check (the rule, long before the token arrives)
        if (h == q) continue;         // the token is here: the records that held the checks up are decided now
        doze(q - h);
    }
    // ... then a plain wait for the token.  The path from the poll that finds head == q to the decision is what the chain's
    // other waves wait for, instruction by instruction: the wave next in line does nothing but read, compare and branch.
    if (hit) {   // (to be run again under the token: its own, slower way out)
#pragma nounroll
        while (hh != q) {
            doze(q - hh);
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(hv), "=&v"(st) : "v"(a_head), "v"(a_state) : "memory");
            hh = mw_uni(hv);
        }
        __builtin_amdgcn_s_setprio(MW_PRIO_TOKEN);
        if (heldm && lane == 0) atomicAdd(&tally[1], 1u);
        return 1u;
    }
#pragma nounroll
    while (hh != q) {
        if (MW_SPIN_NEAR && hh + 1u == q) {
#pragma nounroll
            do {
                asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(hv), "=&v"(st) : "v"(a_head), "v"(a_state) : "memory");
                hh = mw_uni(hv);
            } while (hh != q);
            break;
        }
        doze(q - hh);
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(hv), "=&v"(st) : "v"(a_head), "v"(a_state) : "memory");
        hh = mw_uni(hv);
    }
    __builtin_amdgcn_s_setprio(MW_PRIO_TOKEN);   // the chain's other waves are waiting for what follows: in front of the SIMD's other waves
    // (st: read after head == q was -- final)
    u64 again = ballot((st & (MS_REDONE | MS_ACCEPTED)) == (MS_REDONE | MS_ACCEPTED)) & all;
    if ((again | heldm) == 0ull) return 0u;      // the rule: one scalar branch between the token and the decision
    if (again && lane == 0) atomicAdd(&tally[0], 1u);
    while (again && !hit) {
        const u32 i = (u32)__ffsll((long long)again) - 1u;
        again &= again - 1ull;
        const u32 ev = lane < SR_WORDS ? ringL[((snap + i) & ring) * MW_REC_WORDS + lane] : 0u;
        hit = touches(ev);
    }
    if (heldm && lane == 0) atomicAdd(&tally[1], 1u);
    return hit ? 1u : 0u;
}

// ---- out of line: the wave's next MW_TBL_N draws.  Lane j draws the wave's j-th proposal from q on ("Philox per lane"),
// with the static data it names.
__device__ __attribute__((noinline)) void mw_fill_table(u64 *smem, u32 wv, u32 q)
{
    const int lane = threadIdx.x & (WAVE - 1);
    wv = mw_uni(wv); q = mw_uni(q);
    const u32 *ctx = (const u32 *)(smem + MW_CTX_OFF);
    const u32 cv = lane < MC_WORDS ? ctx[lane] : 0u;
    const u32 W = rdlane(cv, MC_W);
    const MwChain C = mw_chain_from_lds(ctx, lane);
    u32 *T = (u32 *)(smem + MW_SHARED_WORDS + MW_RING_WORDS(W) + (size_t)wv * MW_WAVE_WORDS + 128);
    const u64 seed = (u64)rdlane(cv, MC_SEED) | ((u64)rdlane(cv, MC_SEED + 1) << 32);
    const u64 sampled0 = (u64)rdlane(cv, MC_SAMPLED0) | ((u64)rdlane(cv, MC_SAMPLED0 + 1) << 32);
    const u64 cum0 = (u64)rdlane(cv, MC_CUM0) | ((u64)rdlane(cv, MC_CUM0 + 1) << 32);
    const u64 cum1 = (u64)rdlane(cv, MC_CUM1) | ((u64)rdlane(cv, MC_CUM1 + 1) << 32);
    const u64 cum2 = (u64)rdlane(cv, MC_CUM2) | ((u64)rdlane(cv, MC_CUM2 + 1) << 32);
    const u32 gchain = rdlane(cv, MC_GCHAIN), U = C.U, D = C.D;
    const u64 Mtot = (u64)U + D;
    const int j = lane < (int)MW_TBL_N ? lane : 0;
    const u64 tj = sampled0 + (u64)q + (u64)j * W;
    const u32 k0 = (u32)seed, k1 = (u32)(seed >> 32);
    u32 w[4];
    philox4x32_10((u32)tj, (u32)(tj >> 32), gchain, 0u, k0, k1, w);
    const int mv = ((u64)w[0] < cum0) ? 0 : (((u64)w[0] < cum1) ? 1 : (((u64)w[0] < cum2) ? 2 : 3));
    const u64 x64 = (u64)w[2] | ((u64)w[3] << 32);
    const u64 ix = mv >= 2 ? x64 : __umul64hi(x64, mv == 0 ? Mtot : (u64)D);
    FcmEdgeEntry e = {0u, 0u, 0u, 0u}, de = {0u, 0u, 0u, 0u};
    u32 ed = 0u, c0 = MW_NONE, c1 = MW_NONE;
    if (mv == 0 && ix < U) e = C.etab[ix];
    if (mv == 1 && D > 0) {
        ed = C.dbl[(u32)ix];           // a guess (the list is mutable): verified when the proposal runs
        de = C.etab[ed];
        u32 v[4];
        philox4x32_10((u32)tj, (u32)(tj >> 32), gchain, 1u, k0, k1, v);
        const u64 x0 = __umul64hi((u64)v[0] | ((u64)v[1] << 32), Mtot), x1 = __umul64hi((u64)v[2] | ((u64)v[3] << 32), Mtot);
        if (x0 < U) { c0 = (u32)x0; e = C.etab[x0]; }
        if (x1 < U) c1 = (u32)x1;
    }
    if (lane < (int)MW_TBL_N) {
        u32 *t = T + j * MW_TBL_WORDS;
        t[0] = (u32)mv | ((w[1] & 1u) << 8); t[1] = mv >= 2 ? w[1] : ed; t[2] = (u32)ix; t[3] = (u32)(ix >> 32);   // (clique moves, fcm_step_cq.hpp: the whole of w1 picks the order)
        t[4] = e.big; t[5] = e.small; t[6] = e.nb_off; t[7] = e.k;
        t[8] = de.big; t[9] = de.small; t[10] = de.nb_off; t[11] = de.k;
        t[12] = c0; t[13] = c1;
    }
    wave_sync();
}

// WFIX: the waves per chain as a compile-time constant (2: the instantiation the launcher picks for W = 2 -- ring size, LDS map,
// the stride of the proposal loop and the two-entry `vis` minimum fold into immediates and the out-of-line wait for several
// records, which W = 2 never needs, is not in the kernel), or 0: W = p.mw_waves at run time.
template <int MAXT, bool ROWS128, bool SPARSE = false, int WFIX = 0>
__device__ __forceinline__ void mw_wave(const FcmStepParams &p, u64 *smem)
{
    const u32 W = WFIX ? (u32)WFIX : p.mw_waves;           // waves per chain: 2, 4, 8 or 16 (blockDim.x / 64)
    const int lane = threadIdx.x & (WAVE - 1);
    const u32 wv = mw_uni(threadIdx.x >> 6);
    const u32 chain = blockIdx.x;
    const u32 N = (u32)p.nprop;                            // <= FCM_LAUNCH_CHUNK per launch
    if (N == 0) return;

    u64 *ent = smem;                                       // E[i] = {count i, flag, bmin i, bmax i}; E[0]'s flag: state inside the bounds?
    u32 *ctl = (u32 *)(smem + MW_HEAD_OFF);                // [0] head: the next proposal to decide
    u32 *ctx = (u32 *)(smem + MW_CTX_OFF);
    u32 *vis = (u32 *)(smem + MW_VIS_OFF);                 // [w]: every proposal of wave w below this index is in memory
    u32 *ringL = (u32 *)(smem + MW_SHARED_WORDS);          // records of the last 4W proposals
    const u32 ring = 4u * W - 1u;                          // ring mask
    const int maxnw = p.maxnw < 2 ? 2 : p.maxnw;
    u64 *mine_lds = smem + MW_SHARED_WORDS + MW_RING_WORDS(W) + (size_t)wv * MW_WAVE_WORDS;
    u64 *Hp = mine_lds;                                    // split graph + arc list (eval_nodes / walk_nodes)
    const u32 *T = (const u32 *)(mine_lds + 128);          // this wave's next MW_TBL_N proposals
    u64 *st_g = (u64 *)p.stats + (size_t)chain * FCM_DEV_NSTATS;

    const u64 guard_limit = MAXT >= 6 ? p.guard_limit : 0x7FFFFFFFull;
    MwChain C;
    C.rows = p.rows + (size_t)chain * p.rows_per_chain;
    C.dbl = p.dbl + (size_t)chain * p.dbl_stride;
    C.nb = p.nb; C.etab = p.etab;
    C.rows_bytes = p.rows_per_chain * 4ull;
    C.xw = p.xw_ws ? (u64 *)p.xw_ws + (size_t)chain * FCM_XW_WORDS : nullptr;
    C.U = p.U; C.D = p.D; C.stride32 = p.stride32;

    if (wv == 0) {
        const int NC = p.ncounts;
        const bool cl = lane < NC;
        u64 *cnt_g = (u64 *)p.counts + (size_t)chain * FCM_DEV_MAX_COUNTS;
        const u64 c0 = cl ? cnt_g[lane] : 0ull;
        const u64 mn = cl ? p.bmin[lane] : 0ull, mx = cl ? p.bmax[lane] : ~0ull;   // zero-padded (src/util.rs:53-57)
        const bool inb = ballot(cl && (c0 < mn || c0 > mx)) == 0ull;
        if (lane < 9) { ent[lane * 4 + 0] = c0; ent[lane * 4 + 1] = (lane == 0 && inb) ? 1ull : 0ull; ent[lane * 4 + 2] = mn; ent[lane * 4 + 3] = mx; }   // (lane 8: 0, 0, 0, ~0)
        if (lane < 16) vis[lane] = lane < (int)W ? (u32)lane : MW_NONE;
        for (u32 i = (u32)lane; i <= ring; i += WAVE) ringL[i * MW_REC_WORDS + SR_STATE] = MW_NONE & ~15u;   // nobody's record
        if (lane == 0) {
            ctl[0] = 0u;
            *(smem + MW_TALLY_OFF) = 0ull;
            *(u64 *)(ctx + MC_ROWS) = (u64)C.rows; *(u64 *)(ctx + MC_DBL) = (u64)C.dbl; *(u64 *)(ctx + MC_NB) = (u64)C.nb;
            *(u64 *)(ctx + MC_ETAB) = (u64)C.etab; *(u64 *)(ctx + MC_ROWS_BYTES) = C.rows_bytes; *(u64 *)(ctx + MC_SEED) = p.seed;
            *(u64 *)(ctx + MC_SAMPLED0) = st_g[0];         // Philox step index of proposal 0 of this launch
            *(u64 *)(ctx + MC_CUM0) = p.cum0; *(u64 *)(ctx + MC_CUM1) = p.cum1; *(u64 *)(ctx + MC_CUM2) = p.cum2;
            ctx[MC_U] = C.U; ctx[MC_D] = C.D; ctx[MC_STRIDE32] = C.stride32; ctx[MC_GCHAIN] = p.first_chain + chain;
            ctx[MC_MAXNW] = (u32)maxnw; ctx[MC_W] = W; *(u64 *)(ctx + MC_GUARD) = p.guard_limit; *(u64 *)(ctx + MC_XW) = (u64)C.xw;
            ctx[MC_CLIM] = p.commit_words; ctx[MC_CLIM + 1] = 0u;
        }
    }
    mw_barrier();

    // this wave's share of the counters (added to the stats row at the end)
    u32 *tly = (u32 *)(mine_lds + MW_TALLY_WAVE_OFF);       // MA_* words
    if (lane < MA_WORDS + 1) tly[lane] = 0u;
    wave_sync();
    u32 ti = MW_TBL_N;                                     // next table entry; MW_TBL_N = refill

#ifdef MW_STAMP
    u64 st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MW_T(x) const u64 x = __builtin_amdgcn_s_memtime()
#else
#define MW_T(x) do { } while (0)
#endif
#ifdef MW_TRACE   // (diagnostic build: a timeline of chain 0 from proposal MW_TRACE on -- tools/mw_trace.py.  Every wave writes its
                  //  own slice of the stamp buffer with plain stores: no atomics, nothing to wait for)
    u32 ev_n = 0u;
    const u32 ev_cap = (u32)(((u64)p.nchains * 8ull) / W);
#define MW_EV(type, extra) do { if (chain == 0u && q >= (u32)(MW_TRACE) && ev_n < ev_cap) { \
        const u64 t_ = __builtin_amdgcn_s_memtime(); \
        if (lane_id == 0) p.dbgbuf[(size_t)wv * ev_cap + ev_n] = (t_ << 24) | ((u64)(q & 0xFFFFu) << 8) | ((u64)((extra) & 15u) << 4) | (u64)(type); \
        ++ev_n; } } while (0)
#else
#define MW_EV(type, extra) do { } while (0)
#endif
    const int lane_id = lane;
    for (u32 q = wv; q < N; q += W) {
        MW_T(t_start);
        MW_EV(1, 0);
        // (a fresh copy of the lane id per proposal: comparisons with it are recomputed where they are used -- one VALU each --
        // instead of being hoisted out of the loop into SGPR pairs that are then spilled and reloaded)
        int lane = lane_id;
#if MW_K_LANE
        asm volatile("" : "+v"(lane));
#endif
        if (ti >= MW_TBL_N) {
            mw_fill_table(smem, wv, q);
            ti = 0u;
        }
        const u32 tv = lane < (int)MW_TBL_WORDS ? T[ti * MW_TBL_WORDS + lane] : 0u;
        ++ti;

        // ---- the proposal on the state as committed now.  Commits below snap are visible to every load from here on;
        // those from snap on are held against this proposal's reads before it is decided.
        // the oldest undecided proposal is what the chain's other waves end up waiting for: let it go first on its SIMD
        if (mw_uni(__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == q) __builtin_amdgcn_s_setprio(MW_PRIO_OLDEST);
        MwOut O;
        u32 *stage = ringL + (q & ring) * MW_REC_WORDS;      // this proposal's record
        MW_T(t_snap);
        mw_run<MAXT, ROWS128, false, SPARSE>(C, Hp, nullptr, maxnw, tv, 0ull, 0u, 0ull, lane, O, stage, q, vis, wv, q, W, guard_limit);
        const u32 snap = O.snap;
        MW_T(t_run);
        MW_EV(2, q - snap);

        // ---- before the token: everything the decision can have ready.  sv = the staged record (lane i = word i)
#if MW_K_LANE2
        // (and another copy for what follows the run: its lane masks -- lane < 14 is the table entry's and the record's alike -- would
        //  otherwise be kept across the evaluations in spilled SGPR pairs: two v_writelane and two v_readlane instead of one v_cmp)
        lane = lane_id;
        asm volatile("" : "+v"(lane));
#endif
        u32 sv = lane < SR_WORDS ? stage[lane] : 0u;
        u32 w_clr = O.w_clr, w_set = O.w_set;

        // ---- in-order decision.  While waiting for the token, hold the records snap..q-1 against this proposal's reads.
        bool hit = O.need_exact != 0u;
        u32 hitw;                                 // the same as one scalar word (as a lane mask it costs two spilled SGPRs and a select per use)
        const u32 nent = q - snap;                                            // <= 2W - 1
        if (WFIX != 2 && nent >= MW_VEC_MIN && !(MW_ABL & 16)) {   // (W = 2: nent <= 3)
            // several records: all at once and as soon as they are staged (out of line, W >= 4)
            hitw = mw_uni(mw_wait_staged(smem, wv, q, snap, O.Lv1, O.Lv2, hit ? 1u : 0u, W, sv));   // (0 / 1, wave-uniform, and said so: what follows branches on it with scalar branches)
        } else {
            // few records (W = 2 always): each is looked at once, when it is decided
            const u32 cx0 = rdlane(sv, SR_CX0), cx1 = rdlane(sv, SR_CX1), dslot = rdlane(sv, SR_DSLOT);
            const u32 wid_clr = rdlane(sv, SR_WCLR), wid_set = rdlane(sv, SR_WSET);
            u32 c = snap;
            for (;;) {
#if MW_ABL & 16
                break;
#endif
                const u32 h = mw_uni(__hip_atomic_load(&ctl[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));   // h <= q
                while (c < h && !hit) {
                    const u32 ev = lane < SR_WORDS ? ringL[(c & ring) * MW_REC_WORDS + lane] : 0u;
                    ++c;
                    if (rdlane(ev, SR_STATE) & MS_ACCEPTED) hit = mw_touches(ev, O.Lv1, O.Lv2, cx0, cx1, dslot, wid_clr, wid_set);   // (accepted: not empty)
                }
                if (h == q) break;
#ifdef MW_STAMP
                st_acc[7] += 0;   // (W = 2 path: polls were counted here once; the slot now holds the token-tail split)
#endif
                // not yet: doze by how far off the token is (a decision takes several hundred cycles)
                const u32 dist = q - h;
                if (dist >= 4u) __builtin_amdgcn_s_sleep(10);
                else if (dist >= 2u) __builtin_amdgcn_s_sleep(4);
                else if (WFIX == 2) __builtin_amdgcn_s_sleep(MW_SLEEP_NEAR2);
                else __builtin_amdgcn_s_sleep(MW_SLEEP_NEAR);
            }
            __builtin_amdgcn_s_setprio(MW_PRIO_TOKEN);   // the chain's other waves are waiting for what follows: in front of the SIMD's other waves
            hitw = mw_uni(hit ? 1u : 0u);
        }
        asm volatile("" : "+s"(hitw));            // (opaque, or hipcc folds it back into the mask)
        MW_T(t_token);
        MW_EV(3, hitw << 1);
#ifdef MW_STAMP
        const u32 handed = mw_uni(ctl[1]);
#endif
        if (hitw) {
            // under the token nobody else can commit; once every earlier commit is in memory too, run it again, all of it
            { const u32 gone = q << 4; MW_LDS_ST32(mw_lds_addr(stage + SR_STATE), gone, "1"); }   // the staged record is void from here on
            while (mw_vis_min(vis, W, lane) < q) __builtin_amdgcn_s_sleep(1);
            mw_exact_call<MAXT, ROWS128, SPARSE>(smem, wv, tv, q, (u32)(stage - (u32 *)smem));
            const u32 *out = (const u32 *)(mine_lds + 64);
            const u32 xv = lane < 18 ? out[lane] : 0u;
            O.myd = lane < 16 ? (int)xv : 0;
            w_clr = rdlane(xv, 16); w_set = rdlane(xv, 17);
            sv = lane < SR_WORDS ? stage[lane] : 0u;
            wave_sync();
        }

        // ---- Bounds::check; accept or drop (src/lib.rs:185-191) -- under the token, as little as possible: LDS only.
        // The commit's global stores are issued after the token is handed on: until this wave publishes them as
        // visible (vis, at its next proposal) every later proposal holds this decision against its reads anyway.
        MW_T(t_redo);
        const u32 eoff = (u32)min(lane, 8) * 32u;                              // this lane's count entry (lanes without one: entry 8)
        const u32 ebase = mw_lds_addr(ent) + eoff;
        // The entry's round trip is issued first; whatever does not depend on it -- flags, addresses, the state word less
        // its commit bit, the new head -- is made ready while it is on its way (hipcc would wait first): a lone wave
        // issues an instruction every ten cycles or so, and every one between the token and the head store is the chain's.
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u64 cnt; u32 flagw; u32x4 stat;                                        // count | inside-the-bounds flag | bmin, bmax
        asm volatile("ds_read_b64 %0, %3\n\tds_read_b32 %1, %3 offset:8\n\tds_read_b128 %2, %3 offset:16"
                     : "=&v"(cnt), "=&v"(flagw), "=&v"(stat) : "v"(ebase) : "memory");
        const u32 flg = rdlane(sv, SR_FLAGS);
        u32 nonempty = flg & SRF_NONEMPTY;
        const u32 is_dmove = (flg >> 1) & 1u;
        u64 md = (u64)(long long)O.myd;
        static_assert(MS_REDONE == 8u && MS_ACCEPTED == 4u, "state word: flags by shifts");
        u32 fin0 = (q << 4) | (hitw << 3) | MS_DECIDED;                        // the record's state word, less MS_ACCEPTED
        u32 va_state = mw_lds_addr(stage + SR_STATE), va_head = mw_lds_addr(ctl), v_nh = q + 1u;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(md), "+s"(fin0), "+s"(nonempty), "+v"(va_state), "+v"(va_head), "+v"(v_nh), "+v"(cnt), "+v"(flagw), "+v"(stat) :: "memory");
        const u64 bmin = (u64)stat.x | ((u64)stat.y << 32), bmax = (u64)stat.z | ((u64)stat.w << 32);
        const u32 in_bounds = rdlane(flagw, 0);
        const u64 ncnt = cnt + md;
        const u64 outside = ballot(ncnt < bmin) | ballot(ncnt > bmax);         // (two compares into SGPR pairs and a scalar OR)
        u32 commit;                                                            // = outside == 0 ? nonempty : 0, kept on the scalar side (hipcc goes through a VGPR and back)
        asm("s_cmp_eq_u64 %1, 0\n\ts_cselect_b32 %0, %2, 0" : "=s"(commit) : "s"(outside), "s"(nonempty) : "scc");
        MW_T(t_dec1);
        // the counts, and with them -- second word of the entry, read from entry 0 only -- "the state is inside the bounds":
        // after a commit it is (one store of two words, 8-byte aligned, instead of a test of the flag and a store of its own)
        if (commit) { const u64 onew = 1ull; asm volatile("s_mov_b64 exec, 0xff\n\tds_write2_b64 %0, %1, %2 offset1:1\n\ts_mov_b64 exec, -1" :: "v"(ebase), "v"(ncnt), "v"(onew) : "memory"); }
        const u32 fin = fin0 | (commit << 2);
        MW_LDS_ST32(va_state, fin, "1");
        MW_T(t_dec2);
        MW_LDS_ST32(va_head, v_nh, "1");                                             // the token: after the entry and the counts, in order
        MW_T(t_head);
        MW_EV(4, 0);
#ifdef MW_STAMP
        if (lane == 0) ctl[1] = (u32)t_head;   // (diagnostic: when the token was passed on; the next holder measures the hand-over against it)
#endif
        __builtin_amdgcn_s_setprio(0);
        u32 badw = 0u;                                                         // 0x200: a commit whose indices were out of range (not stored)
#if MW_PROBE == 9 || (MW_ABL & 4)   // (timing probe: no commit stores -- wrong results by design)
        if (commit && q == 0xFFFFFFFFu) {
#else
        if (commit) {
#endif
            // the commit's stores (one value twice if both changes fall into one word: double-edge move only)
            const u32 wid_clr = rdlane(sv, SR_WCLR), wid_set = rdlane(sv, SR_WSET);
            const u32 bit_clr = 1u << ((flg >> 20) & 31u), bit_set = 1u << ((flg >> 25) & 31u);
            u32 nclr = w_clr & ~bit_clr, nset = w_set | bit_set;
            if (wid_clr == wid_set) { nclr |= bit_set; nset = nclr; }
            // These are the hot path's only flat stores whose addresses come out of mutable data (the record; every load of
            // mutable state goes through a clamping buffer descriptor or takes its index from a static table): hold the word
            // indices against the chain's record and the slot and pair against D and U -- the token is gone, this is off the
            // serial path -- and refuse instead of storing (status 0x200 -> FCM_ERR_INTERNAL at the next read-out).
            const u32 clim = mw_uni(ctx[MC_CLIM]);
            const u32 dslot = rdlane(sv, SR_DSLOT), id2 = rdlane(sv, SR_ID2);
            const bool bad = max(wid_clr, wid_set) >= clim || (is_dmove && (dslot >= C.D || id2 >= C.U));
            if (bad) badw = 0x200u;
            else if (lane == 0) {
                C.rows[wid_clr] = nclr;
                C.rows[wid_set] = nset;
                if (is_dmove) C.dbl[dslot] = id2;
            }
        }
#ifdef MW_STAMP
        {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const u64 t_rel = __builtin_amdgcn_s_memtime();
            st_acc[0] += t_snap - t_start;     // table entry, vis publish (waits for this wave's earlier stores), snap
            st_acc[1] += t_run - t_snap;       // the proposal: lists, builds, evaluations
            st_acc[2] += t_token - t_run;      // staging, checks, waiting for the token
            st_acc[3] += q ? (u64)((u32)t_token - handed) : 0ull;   // hand-over: from the previous holder's head store to this wave holding the token
            st_acc[4] += t_rel - t_redo;       // decision under the token
            st_acc[5] += 1;
            st_acc[6] += (t_dec1 - t_redo) | ((t_dec2 - t_dec1) << 32);   // decision split: LDS reads + check | stores
            st_acc[7] += ((t_head - t_dec2) << 32) | ((t_rel - t_head) & 0xFFFFFFFFull);   // ... head store | commit's stores issued (after the token is gone)
        }
#endif

        // ---- the token is gone: this wave's tallies (sampled += 1, src/lib.rs:185).  Lanes 0..7 add to the counters --
        // the low four flag bits are the lanes of their counters as they stand -- lanes 0..8 OR into the flags.
        {
            const u32 acc_inc = commit | ((nonempty ^ 1u) & in_bounds);       // an empty transition is accepted iff the state is inside the bounds
            const u64 im = (u64)((flg & 0xFu) | (1u << MA_MINE) | (acc_inc << MA_ACCEPTED) | (hitw << MA_REDO));
            u32 inc = lane_in(im) ? 1u : 0u;
            inc = lane_in(1ull << MA_SUMK) ? ((flg >> 8) & 0xFFFu) : inc;
            // a count never goes below zero (reference assert, src/lib.rs:65; counts stay far below 2^63, so a negative sum is that)
            const u32 below = (nonempty && (ballot((int)(u32)(ncnt >> 32) < 0) & 0xFFull)) ? 8u : 0u;
            const u32 stw = rdlane(sv, SR_SUS) | below | badw | ((rdlane(tv, 0) & 0xFEu) ? 4u : 0u);   // (move >= 2: this kernel has no clique moves)
            u32 orv = (nonempty && ncnt != 0ull) ? 1u : 0u;                   // lanes 0..7: count entry non-zero after this transition
#if MW_PROBE >= 9 || MW_ABL   // (ablation / timing probes compute wrong results by design: their status word stays clean so that the bench can read the counters)
            orv = lane_in(1ull << 8) ? 0u : orv;
#else
            orv = lane_in(1ull << 8) ? stw : orv;
#endif
            const u32 taddr = mw_lds_addr(tly) + (u32)lane * 4u;
            asm volatile("s_mov_b64 exec, 0xff\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, 0x1ff\n\tds_or_b32 %0, %2 offset:32\n\ts_mov_b64 exec, -1"
                         :: "v"(taddr), "v"(inc), "v"(orv) : "memory");
        }
    }

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(&vis[wv], MW_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // all of this wave's commits are in memory
#ifdef MW_STAMP
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long *)&p.dbgbuf[(size_t)chain * 8 + i], (unsigned long long)st_acc[i]);
#endif
    mw_barrier();   // every proposal decided
    if (wv == 0 && lane < p.ncounts) ((u64 *)p.counts + (size_t)chain * FCM_DEV_MAX_COUNTS)[lane] = ent[lane * 4];
    const u32 tl = lane < MA_WORDS ? tly[lane] : 0u;
    const u64 nzm = ballot(tl != 0u) >> MA_NZ0 & 0xFFull;
    const u32 count_len = nzm ? (u32)(64 - __clzll((long long)nzm)) : 0u;
    const u32 mine = rdlane(tl, MA_MINE), n_nonempty = rdlane(tl, MA_NONEMPTY), n_dmove = rdlane(tl, MA_DMOVE);
    const u32 n_redo = rdlane(tl, MA_REDO), n_wide = rdlane(tl, MA_WIDE), n_big = rdlane(tl, MA_BIG), status = rdlane(tl, MA_STATUS);
    if (lane == 0) {
        atomicAdd((unsigned long long *)&st_g[0], (unsigned long long)mine);
        atomicAdd((unsigned long long *)&st_g[1], (unsigned long long)rdlane(tl, MA_ACCEPTED));
        atomicAdd((unsigned long long *)&st_g[2], (unsigned long long)(mine - n_nonempty));
        atomicAdd((unsigned long long *)&st_g[3], (unsigned long long)(n_nonempty - n_dmove));
        atomicAdd((unsigned long long *)&st_g[4], (unsigned long long)n_dmove);
        atomicAdd((unsigned long long *)&st_g[5], (unsigned long long)rdlane(tl, MA_SUMK));
        atomicMax((unsigned long long *)&st_g[6], (unsigned long long)count_len);
        if (status) atomicOr((unsigned long long *)&st_g[7], (unsigned long long)status);
        if (n_redo) atomicAdd((unsigned long long *)&st_g[11], (unsigned long long)n_redo);
        if (n_wide) atomicAdd((unsigned long long *)&st_g[12], (unsigned long long)n_wide);
        if (n_big) atomicAdd((unsigned long long *)&st_g[13], (unsigned long long)n_big);
        if (wv == 0) {
            const u32 *tally = (const u32 *)(smem + MW_TALLY_OFF);
            if (tally[0]) atomicAdd((unsigned long long *)&st_g[14], (unsigned long long)tally[0]);
            if (tally[1]) atomicAdd((unsigned long long *)&st_g[15], (unsigned long long)tally[1]);
        }
    }
}

// (W is a launch parameter: the block is W x 64 threads.  8 waves per SIMD whatever W is: at most 64 VGPRs.)
template <int MAXT, bool ROWS128, bool SPARSE = false, int WFIX = 0>
__global__ __launch_bounds__(16 * WAVE, MW_MINW) void fcm_step_mw_kernel(const FcmStepParams p)
{
    extern __shared__ u64 smem[];
    if (blockIdx.x >= p.nchains) return;
    mw_wave<MAXT, ROWS128, SPARSE, WFIX>(p, smem);
}
