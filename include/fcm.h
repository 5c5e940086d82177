/*
 * fcm.h — C ABI of libfcm.so, the MI355X-native drop-in for the hot path of
 * TheJonny/flag-complex-mcmc: the `sample` binary's edge-flip MCMC loop
 * (reference src/lib.rs:181-194 in `--simple` mode) and the directed-flag-
 * complex simplex counter it calls (`flagser_count`, reference
 * src/lib.rs:51,63,71,130; legacy FFI src/flagser.rs:7-10).
 *
 * Plain C types only.  All functions return an int status (0 = FCM_OK) unless
 * stated otherwise; fcm_last_error() returns a thread-local message for the
 * last failing call.  Nothing throws across this boundary.
 *
 * Every compute entry point runs on an AMD GPU through HIP.  There is no CPU
 * fallback: without a usable device the call fails with FCM_ERR_NO_DEVICE.
 *
 * Threading: handles are not thread-safe; distinct handles may be used from
 * distinct threads (the reference runs independent `State`s on OS threads,
 * src/bin/all_cxs.rs:33-38).
 *
 * Side effect on the caller: every entry point that touches the GPU makes its
 * `device` (the handle's, or the argument) the calling thread's current HIP
 * device (hipSetDevice) and leaves it so.  A host application that keeps its
 * own current device (PyTorch, ...) must set it again after calling in.
 */
#ifndef FCM_H
#define FCM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCM_OK 0
#define FCM_ERR_INVALID 1      /* bad argument */
#define FCM_ERR_NO_DEVICE 2    /* no HIP device / device index out of range */
#define FCM_ERR_HIP 3          /* a HIP runtime call failed */
#define FCM_ERR_UNSUPPORTED 4  /* valid input outside what this build supports */
#define FCM_ERR_IO 5           /* file could not be read / written / parsed */
#define FCM_ERR_PANIC 6        /* the reference would panic on this input */
#define FCM_ERR_NOMEM 7
#define FCM_ERR_INTERNAL 8     /* device-side consistency check failed */

/* Count vectors handled by the device kernels hold at most this many entries
 * (dimensions 0..15).  The reference has no cap; inputs whose
 * undirected clique number exceeds 16 are refused unless a dim_cap is given. */
#define FCM_MAX_COUNTS 16

/* Largest local vertex set a move may have (|N(a) cap N(b)| + 2).  Up to 64 vertices: the fast evaluator (one mask word
 * per lane); up to 256: the wide one (masks in LDS); up to 1024: masks in a per-chain workspace in HBM, allocated only for
 * graphs that have such a pair (simple moves and clique moves alike).  The counter (flagser_count) takes the
 * same 1024 common out-neighbours of a directed edge (beyond 256 in a second pass, for at most 4096 such edges).  The
 * reference has no such limits; beyond them: FCM_ERR_UNSUPPORTED. */
#define FCM_MAX_LOCAL 1024
#define FCM_MAX_COUNT_LOCAL 256   /* first pass of the counter */

typedef uint32_t fcm_node;     /* reference `Node` = u32 (src/flagser.rs:5,9) */

const char *fcm_last_error(void);
const char *fcm_version(void);

/* Number of visible HIP devices (0 when none). */
int fcm_device_count(int *count);

/* ------------------------------------------------------------------------ */
/* Legacy entry point.  Replaces, symbol for symbol, the C function the      */
/* reference binds at src/flagser.rs:7-10:                                   */
/*   fn flagser_count_unweighted(nvertices: size_t, nedges: size_t,          */
/*        edges: *const [Node; 2], res_size: *mut size_t) -> *mut size_t;    */
/* Ownership as at src/flagser.rs:14-19: the result is malloc'd here, its    */
/* length is written to *res_size, the caller copies and free()s it.  The    */
/* edge buffer is borrowed for the call.  On failure returns NULL with       */
/* *res_size = 0.  Runs on HIP device 0 (env FCM_DEVICE overrides).          */
/* ------------------------------------------------------------------------ */
size_t *flagser_count_unweighted(size_t nvertices, size_t nedges,
                                 const fcm_node (*edges)[2], size_t *res_size);

/* ------------------------------------------------------------------------ */
/* Graph: the `flag_complex::Graph` surface the reference uses               */
/* (call sites src/lib.rs:69,83,125-128,294,310,333,341; */
/* src/io.rs:26,31,39,41).  Host-side handle; adjacency kept as out-row      */
/* bitmaps, the layout the device kernels consume.                           */
/* ------------------------------------------------------------------------ */
typedef struct fcm_graph fcm_graph;

int fcm_graph_new_disconnected(uint32_t nnodes, fcm_graph **out);
int fcm_graph_from_edges(uint32_t nnodes, uint64_t nedges, const fcm_node *edges /* [nedges][2] */,
                         fcm_graph **out);
int fcm_graph_clone(const fcm_graph *g, fcm_graph **out);
void fcm_graph_destroy(fcm_graph *g);

uint32_t fcm_graph_nnodes(const fcm_graph *g);
uint64_t fcm_graph_nedges(const fcm_graph *g);
int fcm_graph_has_edge(const fcm_graph *g, fcm_node a, fcm_node b);        /* 1/0; 0 if out of range */
int fcm_graph_set_edge(fcm_graph *g, fcm_node a, fcm_node b, int present);
int fcm_graph_add_edge(fcm_graph *g, fcm_node a, fcm_node b);
int fcm_graph_remove_edge(fcm_graph *g, fcm_node a, fcm_node b);
/* edges(): ascending (from,to).  Writes min(cap,m) pairs, *m = total. */
int fcm_graph_edges(const fcm_graph *g, fcm_node *out, uint64_t cap, uint64_t *m);
/* undirected_edges(): one [a,b] per adjacent pair, a > b, ascending (a,b). */
int fcm_graph_undirected_edges(const fcm_graph *g, fcm_node *out, uint64_t cap, uint64_t *m);

/* flagser_count() / flag_complex::count_cells (src/lib.rs:51,130): counts[d]
 * = number of directed d-simplices; *len = 1 + highest dimension present.
 * Runs the HIP counting kernel on `device`. */
int fcm_graph_flagser_count(const fcm_graph *g, int device, uint64_t *counts, int cap, int *len);

/* io::read_flag_file (src/io.rs:18-35) and io::save_flag_file (:37-48). */
int fcm_read_flag_file(const char *path, fcm_graph **out);
int fcm_save_flag_file(const char *path, const fcm_graph *g);

/* ------------------------------------------------------------------------ */
/* Bounds (src/lib.rs:113-161) and the init maths of                         */
/* initialize_new_sampler (src/bin/sample.rs:87-102).                        */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint64_t flag_count_min[FCM_MAX_COUNTS + 1];
    int32_t min_len;
    uint64_t flag_count_max[FCM_MAX_COUNTS + 1];
    int32_t max_len;
} fcm_bounds;

/* Target bounds: exact for d<2, floor(s*(1-r)) / floor(s*(1+r)) in f64 for
 * d>=2 (src/bin/sample.rs:89-95). */
int fcm_target_bounds(const uint64_t *flag_count, int len, double target_relaxation, fcm_bounds *out);

/* Bounds::calculate (src/lib.rs:119-156), including the SEO shortcut and the
 * reference's (x-1)! `factorial` (src/util.rs:65-71).  `flag_count` is the
 * initial state's count vector.  The undirected clique counts are computed
 * with the HIP counting kernel on `device` and optionally returned.  Inputs on
 * which the reference panics give FCM_ERR_PANIC. */
int fcm_bounds_calculate(const fcm_graph *initial_graph, const uint64_t *flag_count, int len,
                         const fcm_bounds *target_bounds, int device, fcm_bounds *out,
                         uint64_t *ncliques /* optional, FCM_MAX_COUNTS+1 */, int *ncliques_len);

/* Bounds::check (src/lib.rs:157-160) via all_le (src/util.rs:53-63). */
int fcm_bounds_check(const fcm_bounds *b, const uint64_t *flag_count, int len);

/* ceil(2 * E * log2(E)) (src/bin/sample.rs:102). */
uint64_t fcm_default_sample_distance(uint64_t nedges);

/* ------------------------------------------------------------------------ */
/* Sampler: a batch of independent MCMCSampler chains (src/lib.rs:163-198)   */
/* bound to one device.  One persistent 64-lane workgroup per chain.         */
/* ------------------------------------------------------------------------ */
typedef struct fcm_sampler fcm_sampler;

typedef struct {
    uint32_t n_chains;        /* chains held by this handle */
    uint32_t first_chain_id;  /* global id of chain 0: RNG stream = (seed, first_chain_id + i) */
    uint64_t seed;            /* `--seed` (src/bin/sample.rs:43-45) */
    double move_weights[4];   /* [flip, double-move, clique_permute, clique_swap] (sample.rs:16-17) */
    uint64_t sample_distance; /* proposals per next(); 0 = default (sample.rs:102) */
    int32_t dim_cap;          /* 0 = lossless (track every reachable dimension); d>0 = track
                                 dimensions 0..d only ("truncated") */
    int32_t device;           /* HIP device index */
} fcm_sampler_config;

/* State::new + MCMCSampler construction (src/lib.rs:38-58; sample.rs:104):
 * builds the static neighbourhood table (src/lib.rs:331-356), counts the
 * initial graph on the device, replicates the orientation bitmap per chain. */
int fcm_sampler_create(const fcm_graph *graph, const fcm_bounds *bounds,
                       const fcm_sampler_config *cfg, fcm_sampler **out);
void fcm_sampler_destroy(fcm_sampler *s);

/* Launch stream (a hipStream_t passed as void*); NULL = the handle's own. */
int fcm_sampler_set_stream(fcm_sampler *s, void *hip_stream);

/* `n_proposals` iterations of the loop at src/lib.rs:182-192 on every chain.
 * Asynchronous: returns after the launch; fcm_sampler_sync waits and reports
 * device-side failures. */
int fcm_sampler_step(fcm_sampler *s, uint64_t n_proposals);
/* MCMCSampler::next (src/lib.rs:181-194): sample_distance proposals, then sync. */
int fcm_sampler_next(fcm_sampler *s);
int fcm_sampler_sync(fcm_sampler *s);
/* Duration of the most recent step kernel, from HIP events recorded on the
 * launch stream around it.  Syncs. */
int fcm_sampler_last_step_ms(fcm_sampler *s, float *ms);

/* Number of count entries tracked per chain (<= FCM_MAX_COUNTS). */
int fcm_sampler_ncounts(const fcm_sampler *s);
uint64_t fcm_sampler_sample_distance(const fcm_sampler *s);
/* Sync, then copy per-chain flag_count vectors: out[n_chains][ncounts];
 * count_len[i] (optional) = the reference's flag_count.len() for chain i
 * (never shrinks, src/lib.rs:72-74,89-91). */
int fcm_sampler_get_counts(fcm_sampler *s, uint64_t *out, int32_t *count_len);

/* Per-chain counters, out[n_chains][FCM_NSTATS].  Every slot means the same thing on every sampler; a slot that the
 * sampler's kernel does not count stays 0 (RECHECK / HELD: simple-move samplers on the multi-wave kernel only; CPERM,
 * CSWAP, CHANGES, PAIRS, SHARED_ROWS: samplers with clique moves only). */
#define FCM_NSTATS 18
#define FCM_STAT_SAMPLED 0    /* MCMCSampler::sampled (src/lib.rs:176) */
#define FCM_STAT_ACCEPTED 1   /* MCMCSampler::accepted (src/lib.rs:177) */
#define FCM_STAT_EMPTY 2      /* proposals whose transition was empty */
#define FCM_STAT_FLIP 3       /* non-empty single_edge_flip proposals */
#define FCM_STAT_DMOVE 4      /* non-empty double_edge_move proposals */
#define FCM_STAT_SUM_K 5      /* sum over evaluated edges of |N(a) cap N(b)| */
#define FCM_STAT_COUNT_LEN 6
#define FCM_STAT_STATUS 7     /* 0 ok; non-zero = device-side check failed (bit 0x200: a commit's word or slot index was out of range and
                                 was not stored) */
#define FCM_STAT_CPERM 8      /* non-empty clique_permute proposals */
#define FCM_STAT_CSWAP 9      /* non-empty clique_swap proposals */
#define FCM_STAT_CHANGES 10   /* directed edges changed by clique-move proposals */
#define FCM_STAT_REDO 11      /* multi-wave kernel: proposals run again under the token (a commit in flight touched what they had read, or they
                                 need the full candidate search / the wide evaluator); a timing diagnostic, not chain state */
#define FCM_STAT_WIDE 12      /* proposals evaluated by the wide (multi-word, LDS) evaluator */
#define FCM_STAT_BIG 13       /* evaluated local sets of more than 48 vertices (second trip of the whole-row build) */
#define FCM_STAT_RECHECK 14   /* multi-wave kernel, W >= 4: proposals that had to check a record again under the token (it was decided on an exact run after they had checked it as staged) */
#define FCM_STAT_HELD 15      /* multi-wave kernel, W >= 4: proposals that waited for the decision of a staged record in conflict with their reads */
#define FCM_STAT_PAIRS 16        /* clique moves: vertex pairs with a changed direction (one local build each; FCM_STAT_CHANGES counts directions) */
#define FCM_STAT_SHARED_ROWS 17  /* clique_permute: (changed pairs - 1) x clique order per move -- rows of the clique's own vertices that the builds
                                    of one move read more than once (bench.py's traffic model of a clique move) */
int fcm_sampler_get_stats(fcm_sampler *s, uint64_t *out);

/* Directed edge list of one chain's current graph, ascending (from,to). */
int fcm_sampler_get_edges(fcm_sampler *s, uint32_t chain, fcm_node *out, uint64_t cap, uint64_t *m);
/* One `BitOutput::save` record (src/io.rs:169-205): slots = both directions
 * of every adjacent pair sorted by (max,min,a<b), packed LSB-first, padded to
 * a byte.  *nbytes = ceil(2*U/8). */
int fcm_sampler_get_edgebits(fcm_sampler *s, uint32_t chain, uint8_t *out, uint64_t cap, uint64_t *nbytes);
/* The chain's reciprocal-pair slot list (DESIGN.md draw spec), for parity tests. */
int fcm_sampler_get_double_slots(fcm_sampler *s, uint32_t chain, uint32_t *out, uint64_t cap, uint64_t *n);

/* ------------------------------------------------------------------------ */
/* The State API on one chain of the batch: what the reference's search tools  */
/* call between proposals (src/bin/seo_search_counterexample.rs:51-89,         */
/* seo_bt_flip_only_once.rs:37-39,65-69, all_cxs.rs:54-86).  A Transition      */
/* (src/lib.rs:200-204) is `n` change edges: edges[i] = [from, to],            */
/* add[i] != 0 = add the edge, 0 = remove it.  Every change edge must lie on   */
/* an adjacent pair of pr(G) -- the reference indexes edge_neighborhood with   */
/* it and panics otherwise (FCM_ERR_PANIC).  Host-orchestrated (a few small    */
/* launches and copies per call); the calls sync the handle's stream first.    */
/* ------------------------------------------------------------------------ */
/* State::edgeset_neighborhood (src/lib.rs:99-111): the union of the edges'    */
/* common neighbourhoods and endpoints, ascending, deduplicated.  Writes        */
/* min(cap, *k) vertices.                                                       */
int fcm_sampler_edgeset_neighborhood(fcm_sampler *s, const fcm_node *edges /* [n][2] */, uint32_t n,
                                     fcm_node *out, uint64_t cap, uint64_t *k);
/* State::apply_transition (src/lib.rs:61-79) on chain `chain`: returns the    */
/* reference's (pre, post) = flagser_count of the induced subgraph on the      */
/* edge set's neighbourhood before and after the changes (pre, post: room for  */
/* FCM_MAX_COUNTS entries each; *_len = the vectors' lengths), and updates the */
/* chain's graph and flag_count.  If the reference's `assert!(*s >= *p)`       */
/* (:65) would fire: FCM_ERR_PANIC and nothing is changed.  Not supported      */
/* (FCM_ERR_UNSUPPORTED, nothing changed): a transition that changes the       */
/* NUMBER of reciprocal pairs (the kernels draw double-edge moves over a fixed */
/* number of slots; none of the reference's own move generators does that), or */
/* that leaves an adjacent pair with no edge (pr(G) would change).             */
int fcm_sampler_apply_transition(fcm_sampler *s, uint32_t chain, const fcm_node *edges /* [n][2] */, const int32_t *add /* [n] */,
                                 uint32_t n, uint64_t *pre, int32_t *pre_len, uint64_t *post, int32_t *post_len);
/* State::revert_transition (src/lib.rs:81-95): set_edge(a, b, !add) for every */
/* change, flag_count -= post, += pre.  No counting.                           */
int fcm_sampler_revert_transition(fcm_sampler *s, uint32_t chain, const fcm_node *edges, const int32_t *add, uint32_t n,
                                  const uint64_t *pre, int32_t pre_len, const uint64_t *post, int32_t post_len);
/* Transition::single_edge_flip (src/lib.rs:292-299) drawn on the chain's      */
/* current graph, not applied.  The reference draws with the caller's rng;     */
/* here the caller passes one uniform 64-bit number `x` (the draw itself is    */
/* r = mulhi64(x, U + D) names a directed edge).  *n = 0 (empty   */
/* transition: the edge's reverse is present, or the graph has no edge) or 2:  */
/* edges = {[from,to], [to,from]}, add = {0, 1}.                               */
int fcm_sampler_single_edge_flip(fcm_sampler *s, uint32_t chain, uint64_t x, fcm_node *edges /* [2][2] */, int32_t *add /* [2] */,
                                 uint32_t *n);

/* ------------------------------------------------------------------------ */
/* The same, one transition per chain, every chain of the handle in one launch -- what a search that drives many States at   */
/* once needs (the reference's all_cxs runs 100 of them on OS threads, src/bin/all_cxs.rs:33-86).  Chain c's transition is   */
/* m[c] <= m_cap change edges at edges[c][0 .. m[c])[2] / add[c][..]; m[c] = 0 is the empty transition ((pre, post) = ([],  */
/* [])).  pre / post: [n_chains][FCM_MAX_COUNTS], *_len: [n_chains].  A transition whose change edges lie on ONE adjacent    */
/* pair with a local set of at most 64 vertices (every transition of the reference's generators and search tools) is counted  */
/* and applied on the GPU by one wave per chain; others go through the one-chain call above, chain by chain.  status (may be  */
/* NULL): [n_chains] FCM_OK or the code the one-chain call would have returned for that chain -- a failing chain is left       */
/* unchanged, the others are applied, and the call returns FCM_OK; with status == NULL the call returns the first failing      */
/* chain's code (the other chains are applied all the same).                                                                   */
/* ------------------------------------------------------------------------ */
int fcm_sampler_apply_transitions(fcm_sampler *s, const fcm_node *edges /* [n_chains][m_cap][2] */, const int32_t *add /* [n_chains][m_cap] */,
                                  const uint32_t *m /* [n_chains] */, uint32_t m_cap, uint64_t *pre, int32_t *pre_len, uint64_t *post,
                                  int32_t *post_len, int32_t *status);
int fcm_sampler_revert_transitions(fcm_sampler *s, const fcm_node *edges, const int32_t *add, const uint32_t *m, uint32_t m_cap,
                                   const uint64_t *pre, const int32_t *pre_len, const uint64_t *post, const int32_t *post_len, int32_t *status);
/* Transition::single_edge_flip drawn on every chain at once: x[c] = one uniform 64-bit number for chain c; edges:           */
/* [n_chains][2][2], add: [n_chains][2], n: [n_chains] (0 or 2), laid out for fcm_sampler_apply_transitions with m_cap = 2.  */
int fcm_sampler_single_edge_flips(fcm_sampler *s, const uint64_t *x, fcm_node *edges, int32_t *add, uint32_t *n);

/* ------------------------------------------------------------------------ */
/* Checkpoint / resume: the role of io::save_state / io::load_state           */
/* (src/io.rs:51-62; called at src/bin/sample.rs:114-115,129-132,146).  Own    */
/* format (the reference's is bincode of serde-derived types that live in the */
/* absent crate): header, bounds, config, the adjacent pairs,   */
/* and per chain the 2-bit orientation record (= one edgebits record), the    */
/* reciprocal-pair slot list, counts and counters (incl. the Philox position  */
/* = sampled).  Written to <path>.tmp then renamed, like the reference.       */
/* `sample_number` is the caller's loop index, stored and returned verbatim.  */
/* ------------------------------------------------------------------------ */
int fcm_sampler_save_state(fcm_sampler *s, const char *path, uint64_t sample_number);
int fcm_sampler_load_state(const char *path, int device, fcm_sampler **out, uint64_t *sample_number);
/* A run whose chains are spread over several handles (one per device) is saved as one file per handle.  Every file of  */
/* the set says which shard of how many it is, how many chains the whole run has and carries the set's id (any number   */
/* the caller picks per save, e.g. sample_number: files of two different saves must not be mixed); with the handle's     */
/* first_chain_id and n_chains that is enough to check, at resume, that the files present tile chains 0..total-1 of ONE  */
/* save -- whatever the number of devices then is.  fcm_sampler_save_state = shard 0 of 1 of a run whose chains end with */
/* the handle's (total_chains = first_chain_id + n_chains).                                                              */
int fcm_sampler_save_state_shard(fcm_sampler *s, const char *path, uint64_t sample_number, uint32_t shard_index, uint32_t shard_count,
                                 uint64_t total_chains, uint64_t set_id);
typedef struct {
    uint64_t sample_number, total_chains, set_id, seed;
    uint32_t n, n_chains, first_chain_id, shard_index, shard_count;
} fcm_state_info;
/* The header of a state file, without loading it. */
int fcm_state_file_info(const char *path, fcm_state_info *out);

/* Diagnostic: per-chain phase cycle sums of a -DFCM_STAMP build of the step kernel
 * (tools/run_stamps.sh); all zero in the product build.  out[n_chains][8]. */
int fcm_sampler_debug_stamps(fcm_sampler *s, uint64_t *out);

/* Static facts about the sampler (for bench accounting). */
typedef struct {
    uint32_t n;                /* vertices */
    uint32_t row_words;        /* 64-bit words per bitmap row actually stored (padded) */
    uint64_t n_undirected;     /* U */
    uint64_t n_double;         /* D (constant under the simple moves) */
    uint32_t k_max;            /* max |N(a) cap N(b)| */
    double k_mean;
    uint64_t bytes_per_chain;  /* HBM bytes of mutable state per chain */
    uint64_t bytes_static;     /* HBM bytes of shared read-only tables */
    int32_t ncounts;
    int32_t lossless;          /* 1 if every reachable dimension is tracked */
    uint32_t n_chains;
    uint32_t waves_per_chain;  /* Move mixes with clique moves: the waves that share a move's changed pairs in the cooperative kernel
                                  (fcm_step_cq_kernel: 8 up to 512 chains, 4 up to 1024, 2 up to 2048; FCM_CQW overrides); 1 = the one-wave
                                  kernel (more than 2048 chains, more than 8 count entries, or FCM_CQ=0).  Simple moves:
                                  1: one wave per chain (fcm_step_kernel); W = 2, 4, 8, 16: the multi-wave kernel (fcm_step_mw_kernel),
                                  W consecutive proposals of a chain in flight, decided in order.  The library chooses (simple
                                  moves, <= 8 count entries; the largest W of 8, 4, 2 with chains x W <= 8192 wave slots; 16 on graphs of more than 1024 vertices whose builds touch many cache lines, or with up to 256 chains); environment FCM_MW=<W> overrides
                                  (1 = one-wave kernel).  Trajectories are identical whatever W is. */
    uint32_t sparse_state;     /* 1: the chains' graphs are held as two bits per adjacent pair of pr(G) (the edgebits layout, src/io.rs:152-159)
                                  instead of row bitmaps: chosen for graphs of more than 1024 vertices with local sets of at most 11
                                  vertices and at most two common neighbours per pair on average, under the simple moves (BASELINE
                                  configs[4]: 250 KB per chain instead of 115 MB).  Environment FCM_SPARSE=0 / 1 overrides.  Results do
                                  not depend on it. */
    uint32_t cooperative_clique_kernel;   /* 1: move mixes with clique moves run on fcm_step_cq_kernel (waves_per_chain of them share a move's
                                             changed pairs; it may be 1), 0: on the one-wave kernel's clique path (or no clique moves) */
} fcm_sampler_info;
int fcm_sampler_get_info(const fcm_sampler *s, fcm_sampler_info *out);
/* The Bounds the sampler checks against (MCMCSampler::bounds, src/lib.rs:170). */
int fcm_sampler_get_bounds(const fcm_sampler *s, fcm_bounds *out);

/* ------------------------------------------------------------------------ */
/* Environment variables the library reads (all optional):                     */
/*   FCM_DEVICE=<d>            device of the legacy flagser_count_unweighted   */
/*   FCM_MW=<1|2|4|8|16>       waves per chain of the step kernel (a tuning    */
/*                             override: trajectories do not depend on it)     */
/*   FCM_CQ=<0|1>              clique moves: 0 = the one-wave kernel, 1 = the       */
/*                             cooperative kernel (default: cooperative up to 2048    */
/*                             chains);  FCM_CQW=<1|2|4|8> its waves per chain        */
/*   FCM_SPARSE=<0|1>          per-chain state as two bits per adjacent pair          */
/*                             (fcm_sampler_info.sparse_state) off / on where possible */
/*   FCM_TEST_GUARD_LIMIT=<v>  TEST HOOK: lowers the bound at which a local    */
/*                             count is taken to risk passing 2^31 (DESIGN.md  */
/*                             4.5), so that the tests can drive the guarded   */
/*                             paths on small graphs.  Read once, at            */
/*                             fcm_sampler_create.  Results are unchanged       */
/*                             (wide 64-bit path) or the run fails loudly.      */
/* ------------------------------------------------------------------------ */

#ifdef __cplusplus
}
#endif
#endif /* FCM_H */
