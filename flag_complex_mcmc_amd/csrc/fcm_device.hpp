// fcm_device.hpp — structs shared by the host code (fcm_host.cpp) and the
// gfx950 kernels (fcm_kernels.hip).
#pragma once
#include <stdint.h>

#define FCM_DEV_MAX_COUNTS 16
#define FCM_DEV_NSTATS 18
#define FCM_STAT_COUNT_LEN_DEV 6   // (= FCM_STAT_COUNT_LEN of include/fcm.h)
#define FCM_MAX_SUB 32           // philox blocks (2 candidates each) tried for the single edge of a double-edge move
#define FCM_LAUNCH_CHUNK (1u << 16)  // proposals per chain per kernel launch

// One entry per undirected edge e of pr(G), ascending (big, small):
// the endpoints and the CSR slice of its static common neighbourhood
// (reference State::edge_neighborhood, src/lib.rs:32, built at :331-356).
struct FcmEdgeEntry {
    uint32_t big, small, nb_off, k;
};

// HBM layout (DESIGN.md "Data layout"):
//   rows  [n_chains][n][stride32] u32   out-row bitmaps, one chain after another; row = 128-B multiple
//   dbl   [n_chains][dbl_stride]  u32   reciprocal-pair slot list (undirected edge ids)
//   counts[n_chains][16]          u64   flag_count per chain
//   stats [n_chains][18]          u64   FCM_STAT_* counters
struct FcmStepParams {
    const FcmEdgeEntry *etab;  // [U]
    const uint32_t *nb;        // concatenated neighbour lists
    uint32_t *rows;
    uint32_t *dbl;
    uint64_t *counts;
    uint64_t *stats;
    uint64_t bmin[FCM_DEV_MAX_COUNTS];
    uint64_t bmax[FCM_DEV_MAX_COUNTS];
    uint64_t cum0, cum1;       // move thresholds scaled to 2^32: flip if w0<cum0, double-move if w0<cum1
    uint64_t seed;
    uint64_t nprop;            // proposals to run in this launch
    uint64_t rows_per_chain;   // n * stride32 (u32 words)
    uint32_t n, stride32, U, D, dbl_stride, first_chain, nchains;
    // clique moves (reference src/lib.rs:214-290); all zero / null when their weights are 0
    const uint32_t *clq;       // maximal cliques of pr(G): bucket o-1 = cl_count[o-1] cliques of o vertices from clq[cl_base[o-1]]
    const uint32_t *clq_pairs; // etab index of every vertex pair of every maximal clique: bucket o-1 from clq_pairs[clp_base[o-1]],
                               // o(o-1)/2 per clique, positions (0,1),(0,2),..,(1,2),..
    const uint32_t *efirst;    // [n+1] first etab index whose `big` is v (etab is sorted by (big, small))
    uint32_t *slot_of;         // [n_chains][U] inverse of dbl: slot of a reciprocal pair, 0xFFFFFFFF otherwise
    uint64_t cl_base[FCM_DEV_MAX_COUNTS];
    uint64_t cl_count[FCM_DEV_MAX_COUNTS];
    uint64_t clp_base[FCM_DEV_MAX_COUNTS];
    uint64_t cumo[FCM_DEV_MAX_COUNTS];   // clique_order_distribution as 2^32-scaled thresholds (sample.rs:87-88)
    uint64_t cum2;             // clique_permute if w0 < cum2, else clique_swap
    int32_t cl_orders;         // cliques_by_order.len()
    uint32_t chg_cap;          // u64 words of LDS for the changed-pair list of a clique move
    uint64_t *dbgbuf;          // [n_chains][8] cycle sums of a -DFCM_STAMP diagnostic build; unused otherwise
    int32_t ncounts;           // tracked count entries NC (<= 16)
    int32_t maxnw;             // mask words the largest local set needs: ceil((k_max+2)/64), 1..4
    uint64_t *xw_ws;           // [n_chains][FCM_XW_WORDS] workspace of the evaluator for local sets of 257..1024 vertices (fcm_xwide.hpp); null if the graph has none
    uint64_t guard_limit;      // largest local count bound a walk may reach before it refuses (2^31 - 1; fcm_count_guard)
    uint32_t mw_waves;         // multi-wave kernel (fcm_step_mw.hpp): waves per chain, a power of two 2..16; 0 = one-wave kernel
                               // (the cooperative clique-move kernel, fcm_step_cq.hpp: its waves per chain, 0 = 1)
    uint32_t commit_words;     // u32 words of one chain's mutable record (= rows_per_chain): the multi-wave and cooperative kernels hold a commit's
                               // word indices against it before they store (test hook FCM_TEST_COMMIT_LIMIT lowers it)
    uint32_t sparse;           // 1: `rows` holds, per chain, two bits per adjacent pair (rows_per_chain u32 words) instead of row bitmaps, and `nb`
                               // the local pair ids behind every neighbour list (fcm_step_mw.hpp, mw_build_sparse); multi-wave kernel only
};

struct FcmCountParams {
    const uint32_t *rows;      // [n][stride32] out-row bitmap of the graph to count
    const uint32_t *edges;     // [m][2] directed edges (from,to)
    uint64_t m;
    uint64_t *counts;          // [16], entries 2.. accumulated with atomics
    uint32_t *flags;           // [0] != 0: local set too large; [1] != 0: dimension > 15 present
    uint32_t n, stride32;
    uint32_t *xlist;           // [0] = number of edges with more than 256 common out-neighbours, [1 ..] their indices (second pass)
    uint32_t xcap;             // entries xlist can hold
    uint64_t *xw_ws;           // second pass: [grid][FCM_XW_WORDS] workspaces (fcm_xwide.hpp)
};

// Batched State API (fcm_sampler_apply_transitions / _revert_transitions / _single_edge_flips): one transition per chain, all
// chains in one launch.  A transition the kernel takes changes the directions of ONE adjacent pair (every transition the
// reference's generators and search tools make) whose local set has at most 64 vertices; the host folds its change edges into
// `ops` (per direction: keep / set / clear) and runs the rest through the one-chain path.
#define FCM_TR_SKIP 0xFFFFFFFFu      // pair[c]: nothing for the kernel to do on this chain
enum { FCM_TRS_OK = 0, FCM_TRS_HOST = 1, FCM_TRS_ASSERT = 2, FCM_TRS_UNSUPPORTED = 3, FCM_TRS_DEEP = 4, FCM_TRS_TABLE = 5 };
struct FcmApplyParams {
    const FcmEdgeEntry *etab;
    const uint32_t *nb;
    uint32_t *rows;
    uint64_t *counts, *stats;
    const uint32_t *pair;      // [n_chains] etab index of the transition's pair, or FCM_TR_SKIP
    const uint32_t *ops;       // [n_chains] bits 0-1: big -> small (0 keep, 1 set, 2 clear), bits 2-3: small -> big
    uint64_t *pre, *post;      // [n_chains][FCM_DEV_MAX_COUNTS]: apply writes them; revert reads them
    uint32_t *lens;            // [n_chains][2] pre_len, post_len (apply writes, revert reads)
    uint32_t *status;          // [n_chains] FCM_TRS_*
    uint64_t rows_per_chain;
    uint32_t stride32, nchains, ncounts, revert;
};
struct FcmFlipDrawParams {
    const FcmEdgeEntry *etab;
    const uint32_t *rows;
    const uint64_t *x;         // [n_chains] one uniform 64-bit number per chain
    uint32_t *out;             // [n_chains][3]: pair id (FCM_TR_SKIP: empty transition), from, to
    uint64_t rows_per_chain;
    uint32_t stride32, nchains, U, D, sparse;
};

#ifdef __cplusplus
extern "C" {
#endif
int fcm_launch_apply_batch(const FcmApplyParams *p, void *stream);
int fcm_launch_flip_draw(const FcmFlipDrawParams *p, void *stream);
// launchers implemented in fcm_kernels.hip; `stream` is a hipStream_t
int fcm_launch_step(const FcmStepParams *p, int tmax, int clique, void *stream);
int fcm_launch_count(const FcmCountParams *p, void *stream);
int fcm_launch_count_xw(const FcmCountParams *p, uint32_t nflagged, void *stream);   // the edges of xlist, one wave each
int fcm_launch_gather_sub(const uint32_t *rows, uint32_t stride32, const uint32_t *list, uint32_t nl, uint32_t nlw, uint32_t *out, void *stream);
int fcm_launch_set_edges(uint32_t *rows, uint32_t stride32, const uint32_t *changes, uint32_t nchanges, void *stream);
int fcm_launch_broadcast_rows(uint32_t *rows, const uint32_t *base, uint64_t words_per_chain, uint32_t nchains, void *stream);
#ifdef __cplusplus
}
#endif
