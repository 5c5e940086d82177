/*
 * fcm_oracle.c — CPU restatement of the reference's edge-flip MCMC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product library links, imports or
 * calls this file.  It is the checker used by tests/, __graft_entry__.smoke()
 * and the `cpu_baseline` leg of bench.py, never the thing shipped or measured
 * as the product.
 *
 * PARITY STATUS: "parity unpinned" for everything except intersect_sorted.
 *   - The reference's arithmetic (Graph, flagser_count, subgraph, sample_edge,
 *     sample_double_edge) lives in the external crate `flag-complex`
 *     (Cargo.toml:26, git HEAD, no revision pin, Cargo.lock ignored).  Its
 *     source is not in /root/reference and no Rust toolchain exists here, so
 *     the reference cannot be built or run.
 *   - The only golden vectors the reference holds for this path are the eight
 *     `test_intersect` cases (src/util.rs:107-156); they are checked in
 *     tests/test_oracle_golden.py.
 *   - The simplex counter below restates the published definition of the
 *     directed flag complex (flagser): count[d] = number of
 *     ordered (d+1)-tuples of distinct vertices with v_i -> v_j for all i<j.
 *     It is cross-checked in tests/ by a brute-force permutation enumerator
 *     written independently in numpy, and by the structural facts the
 *     reference's own fixtures imply (see tests/test_oracle_golden.py).
 *
 * Every function cites the reference file:line it follows.
 *
 * The proposal *draw* (how RNG output maps to an edge) cannot follow the
 * reference bit-for-bit: `sample_edge`/`sample_double_edge` are in the absent
 * crate and the reference RNG is Xoshiro256** through the `rand` crate.  The
 * draw here is this build's own specification (DESIGN.md "Draw spec"):
 * counter-based Philox4x32-10, distributionally identical to the reference
 * (uniform directed edge; uniform reciprocal pair; rejection-sampled single
 * edge; fair coin).  The oracle implements that spec in the reference's
 * algorithmic shape (neighbourhood lookup, induced-subgraph recount before and
 * after, integer bounds check, revert from saved vectors).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <errno.h>

#define FO_MAXDIM 64 /* count vectors hold at most FO_MAXDIM entries */

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11; Random123).  Independent restatement */
/* of the published algorithm; the product has its own copy in csrc/.        */
/* ------------------------------------------------------------------------ */
static void fo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void fo_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    fo_philox4x32_10(ctr, key, out);
}

static inline uint64_t fo_mulhi64(uint64_t a, uint64_t b)
{
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
}

/* ------------------------------------------------------------------------ */
/* Graph: the `flag_complex::Graph` surface the reference uses               */
/* (SURVEY.md App. A.1).  Out-row bitmaps.                                   */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint32_t n;
    uint32_t wq;     /* 64-bit words per row */
    uint64_t *out;   /* n * wq */
} fo_graph;

/* Graph::new_disconnected (called at src/lib.rs:126, src/io.rs:26) */
fo_graph *fo_graph_new(uint32_t n)
{
    fo_graph *g = (fo_graph *)calloc(1, sizeof *g);
    if (!g) return NULL;
    g->n = n;
    g->wq = (n + 63) / 64;
    if (g->wq == 0) g->wq = 1;
    g->out = (uint64_t *)calloc((size_t)n * g->wq + 1, sizeof(uint64_t));
    if (!g->out) { free(g); return NULL; }
    return g;
}

void fo_graph_free(fo_graph *g)
{
    if (!g) return;
    free(g->out);
    free(g);
}

fo_graph *fo_graph_clone(const fo_graph *g)
{
    fo_graph *h = fo_graph_new(g->n);
    if (!h) return NULL;
    memcpy(h->out, g->out, (size_t)g->n * g->wq * sizeof(uint64_t));
    return h;
}

uint32_t fo_graph_nnodes(const fo_graph *g) { return g->n; }

/* has_edge (src/lib.rs:294,310) */
int fo_graph_has_edge(const fo_graph *g, uint32_t a, uint32_t b)
{
    return (int)((g->out[(size_t)a * g->wq + (b >> 6)] >> (b & 63)) & 1u);
}

/* set_edge (src/lib.rs:69,83); self-loops are not representable in a flag
 * complex and are ignored. */
void fo_graph_set_edge(fo_graph *g, uint32_t a, uint32_t b, int present)
{
    if (a == b) return;
    uint64_t *w = &g->out[(size_t)a * g->wq + (b >> 6)];
    uint64_t bit = 1ull << (b & 63);
    if (present) *w |= bit; else *w &= ~bit;
}

/* add_edge (src/lib.rs:128, src/io.rs:31) */
void fo_graph_add_edge(fo_graph *g, uint32_t a, uint32_t b) { fo_graph_set_edge(g, a, b, 1); }

/* edges(): all directed edges, here in ascending (from,to) order.  Returns
 * the number of edges; writes up to cap pairs. */
uint64_t fo_graph_edges(const fo_graph *g, uint32_t *out_pairs, uint64_t cap)
{
    uint64_t m = 0;
    for (uint32_t a = 0; a < g->n; ++a) {
        const uint64_t *row = &g->out[(size_t)a * g->wq];
        for (uint32_t w = 0; w < g->wq; ++w) {
            uint64_t x = row[w];
            while (x) {
                uint32_t b = w * 64 + (uint32_t)__builtin_ctzll(x);
                x &= x - 1;
                if (out_pairs && m < cap) { out_pairs[2 * m] = a; out_pairs[2 * m + 1] = b; }
                ++m;
            }
        }
    }
    return m;
}

/* undirected_edges(): one [a,b] per adjacent pair with a > b
 * (src/lib.rs:125,341,344).  Order: ascending (a,b). */
uint64_t fo_graph_undirected_edges(const fo_graph *g, uint32_t *out_pairs, uint64_t cap)
{
    uint64_t u = 0;
    for (uint32_t a = 0; a < g->n; ++a)
        for (uint32_t b = 0; b < a; ++b)
            if (fo_graph_has_edge(g, a, b) || fo_graph_has_edge(g, b, a)) {
                if (out_pairs && u < cap) { out_pairs[2 * u] = a; out_pairs[2 * u + 1] = b; }
                ++u;
            }
    return u;
}

/* Graph::subgraph(&g, &nodes): induced subgraph, vertices relabelled by
 * position in `nodes` (src/lib.rs:63,71; src/bin/edgeset_nbhd.rs:20-32). */
fo_graph *fo_graph_subgraph(const fo_graph *g, const uint32_t *nodes, uint32_t k)
{
    fo_graph *s = fo_graph_new(k);
    if (!s) return NULL;
    for (uint32_t i = 0; i < k; ++i)
        for (uint32_t j = 0; j < k; ++j)
            if (i != j && fo_graph_has_edge(g, nodes[i], nodes[j]))
                fo_graph_set_edge(s, i, j, 1);
    return s;
}

/* ------------------------------------------------------------------------ */
/* flagser_count / count_cells (external; called at src/lib.rs:51,63,71,130; */
/* legacy C form src/flagser.rs:9).  Definition: SURVEY.md App. A.2.         */
/* counts[d] = #ordered (d+1)-tuples of distinct vertices, all forward edges.*/
/* Returns the vector length (1 + highest dimension present; 0 if n == 0).   */
/* ------------------------------------------------------------------------ */
typedef struct {
    const fo_graph *g;
    uint64_t *counts;
    uint64_t *stack; /* FO_MAXDIM * wq words */
} fo_count_ctx;

static void fo_count_rec(fo_count_ctx *c, const uint64_t *cand, int depth)
{
    const fo_graph *g = c->g;
    if (depth >= FO_MAXDIM) return;
    uint64_t *next = c->stack + (size_t)depth * g->wq;
    for (uint32_t w = 0; w < g->wq; ++w) {
        uint64_t x = cand[w];
        while (x) {
            uint32_t v = w * 64 + (uint32_t)__builtin_ctzll(x);
            x &= x - 1;
            c->counts[depth] += 1;
            const uint64_t *row = &g->out[(size_t)v * g->wq];
            uint64_t any = 0;
            for (uint32_t q = 0; q < g->wq; ++q) { next[q] = cand[q] & row[q]; any |= next[q]; }
            if (any) fo_count_rec(c, next, depth + 1);
        }
    }
}

int fo_graph_flagser_count(const fo_graph *g, uint64_t *counts /* FO_MAXDIM */)
{
    memset(counts, 0, FO_MAXDIM * sizeof(uint64_t));
    if (g->n == 0) return 0;
    fo_count_ctx c;
    c.g = g;
    c.counts = counts;
    c.stack = (uint64_t *)malloc((size_t)FO_MAXDIM * g->wq * sizeof(uint64_t));
    if (!c.stack) return -1;
    counts[0] = g->n;
    for (uint32_t v = 0; v < g->n; ++v)
        fo_count_rec(&c, &g->out[(size_t)v * g->wq], 1);
    free(c.stack);
    int len = 0;
    for (int d = 0; d < FO_MAXDIM; ++d) if (counts[d]) len = d + 1;
    return len;
}

/* Legacy C entry point shape (src/flagser.rs:7-10): edges as [from,to] u32
 * pairs; returns malloc'd size_t array, length in *res_size. */
uint64_t *fo_flagser_count_unweighted(uint64_t nvertices, uint64_t nedges,
                                      const uint32_t *edges, uint64_t *res_size)
{
    *res_size = 0;
    fo_graph *g = fo_graph_new((uint32_t)nvertices);
    if (!g) return NULL;
    for (uint64_t i = 0; i < nedges; ++i) {
        if (edges[2 * i] >= nvertices || edges[2 * i + 1] >= nvertices) { fo_graph_free(g); return NULL; }
        fo_graph_add_edge(g, edges[2 * i], edges[2 * i + 1]);
    }
    uint64_t counts[FO_MAXDIM];
    int len = fo_graph_flagser_count(g, counts);
    fo_graph_free(g);
    if (len < 0) return NULL;
    uint64_t *res = (uint64_t *)malloc((len ? len : 1) * sizeof(uint64_t));
    if (!res) return NULL;
    memcpy(res, counts, (size_t)len * sizeof(uint64_t));
    *res_size = (uint64_t)len;
    return res;
}

/* ------------------------------------------------------------------------ */
/* util.rs                                                                    */
/* ------------------------------------------------------------------------ */

/* intersect_sorted (src/util.rs:5-26): two-pointer walk, advances BOTH sides
 * on equality, so duplicates are emitted once per matched pair. */
uint64_t fo_intersect_sorted(const uint32_t *a, uint64_t na, const uint32_t *b, uint64_t nb, uint32_t *out)
{
    uint64_t ai = 0, bi = 0, n = 0;
    while (ai < na && bi < nb) {
        if (a[ai] == b[bi]) { out[n++] = a[ai]; ++ai; ++bi; }
        else if (a[ai] < b[bi]) ++ai;
        else ++bi;
    }
    return n;
}

/* all_le (src/util.rs:53-63): element-wise <= over max(len), shorter side
 * padded with z. */
int fo_all_le(const uint64_t *a, int na, const uint64_t *b, int nb, uint64_t z)
{
    int maxlen = na > nb ? na : nb;
    for (int i = 0; i < maxlen; ++i) {
        uint64_t l = i < na ? a[i] : z;
        uint64_t r = i < nb ? b[i] : z;
        if (l > r) return 0;
    }
    return 1;
}

/* factorial (src/util.rs:65-71): loop is `1..x` (exclusive) so this returns
 * (x-1)! for x >= 1 and 1 for x == 0.  Wrapping multiply like release Rust. */
uint64_t fo_factorial(uint64_t x)
{
    uint64_t res = 1;
    for (uint64_t i = 1; i < x; ++i) res *= i;
    return res;
}

/* binomial (src/util.rs:73-77), built on the off-by-one factorial.  The
 * reference panics when k > n (usize underflow in n-k); we return 0 and the
 * caller treats that as an error. */
uint64_t fo_binomial(uint64_t n, uint64_t k)
{
    if (k > n) return 0;
    return fo_factorial(n) / (fo_factorial(k) * fo_factorial(n - k));
}

/* OEIS A058298 (src/util.rs:98-105): triangle n!/(n-k), 1 <= k < n, read by
 * rows; the reference keeps the first 64 terms.  Generated from the formula. */
static uint64_t fo_a058298[64];
static int fo_a058298_ready = 0;
static void fo_a058298_init(void)
{
    if (fo_a058298_ready) return;
    int idx = 0;
    uint64_t fact = 1;
    for (uint64_t n = 2; idx < 64; ++n) {
        fact *= n; /* n! (fact was (n-1)!) */
        for (uint64_t k = 1; k < n && idx < 64; ++k)
            fo_a058298[idx++] = fact / (n - k);
    }
    fo_a058298_ready = 1;
}
uint64_t fo_oeis_a058298(int i) { fo_a058298_init(); return (i >= 0 && i < 64) ? fo_a058298[i] : 0; }

/* calc_relax_de (src/util.rs:79-93).  The reference indexes past the table
 * (panic) when sc[d] > 159 667 200; we report that as error (-1). */
int fo_calc_relax_de(const uint64_t *sc, int len, uint64_t *relax_de)
{
    fo_a058298_init();
    for (int d = 0; d < len; ++d) {
        int ind = 1;
        uint64_t best = 0; int have = 0;
        for (;;) {
            if (ind >= 64) return -1; /* reference: index out of bounds panic */
            if (!(fo_a058298[ind] < sc[d])) break;
            uint64_t lost = fo_a058298[ind] - fo_a058298[ind - 1];
            if (!have || lost > best) { best = lost; have = 1; }
            ++ind;
        }
        uint64_t a = have ? best : 1;
        uint64_t b = fo_factorial((uint64_t)d + 1);
        relax_de[d] = a < b ? a : b;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* State (src/lib.rs:29-112)                                                 */
/* ------------------------------------------------------------------------ */
typedef struct {
    fo_graph *graph;
    uint64_t flag_count[FO_MAXDIM];
    int flag_count_len;
    /* edge_neighborhood: HashMap<[big,small], Vec<Node>> (src/lib.rs:32), kept
     * as CSR over the undirected edges in ascending (big,small) order. */
    uint64_t n_uedges;
    uint32_t *uedges;     /* 2 * n_uedges: big, small */
    uint64_t *nb_off;     /* n_uedges + 1 */
    uint32_t *nb;         /* concatenated neighbour lists (may hold duplicates, src/lib.rs:333-336) */
    /* cliques_by_order (src/lib.rs:31,41-49): maximal cliques of pr(G) bucketed by
     * order; bucket o-1 holds cl_count[o-1] cliques of o vertices each, stored
     * back to back from cl_flat[cl_base[o-1]].  Canonical layout: vertices of a
     * clique ascending, cliques of a bucket in lexicographic order.  Built on
     * demand (only the clique moves read it). */
    int cliques_ready;
    int cl_orders;                      /* cliques_by_order.len() = largest order */
    uint64_t cl_count[FO_MAXDIM];
    uint64_t cl_base[FO_MAXDIM];
    uint32_t *cl_flat;
} fo_state;

static int fo_cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* compute_edge_neighborhoods (src/lib.rs:331-356): undirected adjacency lists
 * built from the DIRECTED edge list (so a reciprocal pair contributes the
 * neighbour twice), sorted; per undirected edge [a,b], a>b:
 * intersect_sorted(adj[a], adj[b]). */
static int fo_compute_edge_neighborhoods(fo_state *st)
{
    const fo_graph *g = st->graph;
    uint32_t n = g->n;
    uint64_t m = fo_graph_edges(g, NULL, 0);
    uint32_t *edges = (uint32_t *)malloc((m ? m : 1) * 2 * sizeof(uint32_t));
    uint64_t *deg = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    if (!edges || !deg) return -1;
    fo_graph_edges(g, edges, m);
    for (uint64_t i = 0; i < m; ++i) { deg[edges[2 * i]]++; deg[edges[2 * i + 1]]++; }
    uint64_t *off = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    off[0] = 0;
    for (uint32_t v = 0; v < n; ++v) off[v + 1] = off[v] + deg[v];
    uint32_t *adj = (uint32_t *)malloc((m ? 2 * m : 1) * sizeof(uint32_t));
    uint64_t *fill = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    for (uint64_t i = 0; i < m; ++i) {
        uint32_t a = edges[2 * i], b = edges[2 * i + 1];
        adj[off[a] + fill[a]++] = b;
        adj[off[b] + fill[b]++] = a;
    }
    for (uint32_t v = 0; v < n; ++v) qsort(adj + off[v], deg[v], sizeof(uint32_t), fo_cmp_u32);

    st->n_uedges = fo_graph_undirected_edges(g, NULL, 0);
    st->uedges = (uint32_t *)malloc((st->n_uedges ? st->n_uedges : 1) * 2 * sizeof(uint32_t));
    fo_graph_undirected_edges(g, st->uedges, st->n_uedges);
    st->nb_off = (uint64_t *)malloc((st->n_uedges + 1) * sizeof(uint64_t));
    /* two passes: size, then fill */
    uint64_t total = 0, maxdeg = 0;
    for (uint32_t v = 0; v < n; ++v) if (deg[v] > maxdeg) maxdeg = deg[v];
    uint32_t *tmp = (uint32_t *)malloc((maxdeg ? maxdeg : 1) * sizeof(uint32_t));
    for (uint64_t e = 0; e < st->n_uedges; ++e) {
        uint32_t a = st->uedges[2 * e], b = st->uedges[2 * e + 1];
        st->nb_off[e] = total;
        total += fo_intersect_sorted(adj + off[a], deg[a], adj + off[b], deg[b], tmp);
    }
    st->nb_off[st->n_uedges] = total;
    st->nb = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    for (uint64_t e = 0; e < st->n_uedges; ++e) {
        uint32_t a = st->uedges[2 * e], b = st->uedges[2 * e + 1];
        fo_intersect_sorted(adj + off[a], deg[a], adj + off[b], deg[b], st->nb + st->nb_off[e]);
    }
    free(tmp); free(fill); free(adj); free(off); free(deg); free(edges);
    return 0;
}

/* State::new (src/lib.rs:38-58), minus maximal-clique bucketing, which only
 * the clique moves (out of scope) consume.  Takes ownership of
 * a clone of g. */
fo_state *fo_state_new(const fo_graph *g)
{
    fo_state *st = (fo_state *)calloc(1, sizeof *st);
    if (!st) return NULL;
    st->graph = fo_graph_clone(g);
    st->flag_count_len = fo_graph_flagser_count(st->graph, st->flag_count);
    if (fo_compute_edge_neighborhoods(st) != 0) { return NULL; }
    return st;
}

void fo_state_free(fo_state *st)
{
    if (!st) return;
    fo_graph_free(st->graph);
    free(st->uedges); free(st->nb_off); free(st->nb); free(st->cl_flat);
    free(st);
}

/* ------------------------------------------------------------------------ */
/* compute_maximal_cliques (external crate, called at src/lib.rs:41): all      */
/* maximal cliques of pr(G).  Bron-Kerbosch with pivoting over bitsets, one    */
/* outer pass per smallest vertex.  The reference's enumeration order is       */
/* unknown; it is irrelevant to the sampling distribution (cliques are picked  */
/* uniformly, src/lib.rs:216) and the canonical order above replaces it.       */
/* ------------------------------------------------------------------------ */
typedef struct {
    const uint64_t *und; uint32_t n, wq;
    uint32_t R[FO_MAXDIM]; int rlen;
    uint64_t *work;           /* (FO_MAXDIM+1) * 2 * wq words */
    uint32_t *out; uint64_t out_len, out_cap;   /* records: len, v0..v(len-1) */
    int err;
} fo_bk;

static void fo_bk_emit(fo_bk *b)
{
    if (b->out_len + (uint64_t)b->rlen + 1 > b->out_cap) {
        uint64_t nc = b->out_cap ? b->out_cap * 2 : (1u << 20);
        uint32_t *no = (uint32_t *)realloc(b->out, nc * sizeof(uint32_t));
        if (!no) { b->err = 1; return; }
        b->out = no; b->out_cap = nc;
    }
    b->out[b->out_len++] = (uint32_t)b->rlen;
    for (int i = 0; i < b->rlen; ++i) b->out[b->out_len++] = b->R[i];
}

static void fo_bk_rec(fo_bk *b, uint64_t *P, uint64_t *X, int depth)
{
    const uint32_t wq = b->wq;
    uint64_t anyP = 0, anyX = 0;
    for (uint32_t q = 0; q < wq; ++q) { anyP |= P[q]; anyX |= X[q]; }
    if (!anyP) { if (!anyX) fo_bk_emit(b); return; }
    if (depth >= FO_MAXDIM - 1 || b->err) { b->err = 1; return; }
    /* pivot: vertex of P|X with most neighbours in P */
    int best = -1; uint32_t pivot = 0;
    for (uint32_t q = 0; q < wq; ++q) {
        uint64_t x = P[q] | X[q];
        while (x) {
            uint32_t u = q * 64 + (uint32_t)__builtin_ctzll(x); x &= x - 1;
            const uint64_t *nu = b->und + (size_t)u * wq;
            int c = 0;
            for (uint32_t r = 0; r < wq; ++r) c += __builtin_popcountll(P[r] & nu[r]);
            if (c > best) { best = c; pivot = u; }
        }
    }
    uint64_t *nP = b->work + (size_t)(depth + 1) * 2 * wq, *nX = nP + wq;
    const uint64_t *np = b->und + (size_t)pivot * wq;
    for (uint32_t q = 0; q < wq; ++q) {
        uint64_t cand = P[q] & ~np[q];
        while (cand) {
            uint32_t v = q * 64 + (uint32_t)__builtin_ctzll(cand); cand &= cand - 1;
            const uint64_t *nv = b->und + (size_t)v * wq;
            for (uint32_t r = 0; r < wq; ++r) { nP[r] = P[r] & nv[r]; nX[r] = X[r] & nv[r]; }
            b->R[b->rlen++] = v;
            fo_bk_rec(b, nP, nX, depth + 1);
            b->rlen--;
            P[q] &= ~(1ull << (v & 63));
            X[q] |= 1ull << (v & 63);
        }
    }
}

static int fo_cmp_clique_o;  /* order of the bucket being sorted */
static int fo_cmp_clique(const void *a, const void *b)
{
    const uint32_t *x = (const uint32_t *)a, *y = (const uint32_t *)b;
    for (int i = 0; i < fo_cmp_clique_o; ++i) if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
    return 0;
}

int fo_state_ensure_cliques(fo_state *st)
{
    if (st->cliques_ready) return 0;
    const fo_graph *g = st->graph;
    const uint32_t n = g->n, wq = g->wq;
    uint64_t *und = (uint64_t *)calloc((size_t)n * wq + 1, sizeof(uint64_t));
    if (!und) return -1;
    for (uint64_t e = 0; e < st->n_uedges; ++e) {
        uint32_t a = st->uedges[2 * e], b = st->uedges[2 * e + 1];
        und[(size_t)a * wq + (b >> 6)] |= 1ull << (b & 63);
        und[(size_t)b * wq + (a >> 6)] |= 1ull << (a & 63);
    }
    fo_bk b; memset(&b, 0, sizeof b);
    b.und = und; b.n = n; b.wq = wq;
    b.work = (uint64_t *)malloc((size_t)(FO_MAXDIM + 1) * 2 * wq * sizeof(uint64_t));
    if (!b.work) { free(und); return -1; }
    for (uint32_t v = 0; v < n && !b.err; ++v) {
        uint64_t *P = b.work, *X = b.work + wq;
        const uint64_t *nv = und + (size_t)v * wq;
        for (uint32_t q = 0; q < wq; ++q) {   /* later neighbours may extend, earlier ones exclude */
            uint64_t lo = (q < (v >> 6)) ? ~0ull : (q == (v >> 6) ? ((1ull << (v & 63)) - 1ull) : 0ull);
            X[q] = nv[q] & lo;
            P[q] = nv[q] & ~lo & ~((q == (v >> 6)) ? (1ull << (v & 63)) : 0ull);
        }
        b.R[0] = v; b.rlen = 1;
        fo_bk_rec(&b, P, X, 0);
    }
    free(b.work); free(und);
    if (b.err) { free(b.out); return -1; }
    /* bucket by order (src/lib.rs:42-49), canonical order inside a bucket */
    memset(st->cl_count, 0, sizeof st->cl_count);
    st->cl_orders = 0;
    for (uint64_t i = 0; i < b.out_len; i += b.out[i] + 1) {
        int o = (int)b.out[i];
        st->cl_count[o - 1]++;
        if (o > st->cl_orders) st->cl_orders = o;
    }
    uint64_t total = 0;
    for (int o = 1; o <= st->cl_orders; ++o) { st->cl_base[o - 1] = total; total += st->cl_count[o - 1] * (uint64_t)o; }
    st->cl_flat = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    uint64_t fill[FO_MAXDIM]; memset(fill, 0, sizeof fill);
    for (uint64_t i = 0; i < b.out_len; i += b.out[i] + 1) {
        int o = (int)b.out[i];
        uint32_t *dst = st->cl_flat + st->cl_base[o - 1] + fill[o - 1] * (uint64_t)o;
        memcpy(dst, &b.out[i + 1], (size_t)o * sizeof(uint32_t));
        qsort(dst, (size_t)o, sizeof(uint32_t), fo_cmp_u32);
        fill[o - 1]++;
    }
    free(b.out);
    for (int o = 1; o <= st->cl_orders; ++o) {
        fo_cmp_clique_o = o;
        qsort(st->cl_flat + st->cl_base[o - 1], st->cl_count[o - 1], (size_t)o * sizeof(uint32_t), fo_cmp_clique);
    }
    st->cliques_ready = 1;
    return 0;
}

/* number of maximal cliques per order; returns cliques_by_order.len() */
int fo_state_clique_counts(fo_state *st, uint64_t *out /* FO_MAXDIM */)
{
    if (fo_state_ensure_cliques(st)) return -1;
    memcpy(out, st->cl_count, sizeof st->cl_count);
    return st->cl_orders;
}
/* copies bucket `order` (order*count vertices) */
int64_t fo_state_cliques_of_order(fo_state *st, int order, uint32_t *out, uint64_t cap)
{
    if (fo_state_ensure_cliques(st) || order < 1 || order > st->cl_orders) return -1;
    uint64_t nn = st->cl_count[order - 1] * (uint64_t)order;
    if (out) memcpy(out, st->cl_flat + st->cl_base[order - 1], (size_t)(nn < cap ? nn : cap) * sizeof(uint32_t));
    return (int64_t)st->cl_count[order - 1];
}

fo_graph *fo_state_graph(fo_state *st) { return st->graph; }
int fo_state_flag_count(const fo_state *st, uint64_t *out) { memcpy(out, st->flag_count, sizeof st->flag_count); return st->flag_count_len; }
uint64_t fo_state_n_uedges(const fo_state *st) { return st->n_uedges; }
const uint32_t *fo_state_uedges(const fo_state *st) { return st->uedges; }

/* index of undirected edge [big,small] in the sorted list, or -1 */
static int64_t fo_state_uedge_index(const fo_state *st, uint32_t big, uint32_t small)
{
    int64_t lo = 0, hi = (int64_t)st->n_uedges - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) / 2;
        uint32_t a = st->uedges[2 * mid], b = st->uedges[2 * mid + 1];
        if (a == big && b == small) return mid;
        if (a < big || (a == big && b < small)) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

/* neighbourhood list of one undirected edge (for tests) */
int64_t fo_state_edge_neighborhood(const fo_state *st, uint32_t a, uint32_t b, uint32_t *out, uint64_t cap)
{
    uint32_t big = a > b ? a : b, small = a > b ? b : a;
    int64_t e = fo_state_uedge_index(st, big, small);
    if (e < 0) return -1;
    uint64_t len = st->nb_off[e + 1] - st->nb_off[e];
    for (uint64_t i = 0; i < len && i < cap; ++i) out[i] = st->nb[st->nb_off[e] + i];
    return (int64_t)len;
}

/* edgeset_neighborhood (src/lib.rs:99-111): extend by each edge's list, push
 * both endpoints, sort_unstable, dedup.  `edges` = ne pairs.  Returns count,
 * or -1 when an edge is not in the table (reference: HashMap index panic). */
int64_t fo_state_edgeset_neighborhood(const fo_state *st, const uint32_t *edges, uint32_t ne, uint32_t *out, uint64_t cap)
{
    uint64_t k = 0;
    for (uint32_t i = 0; i < ne; ++i) {
        uint32_t a = edges[2 * i], b = edges[2 * i + 1];
        uint32_t big = a > b ? a : b, small = a > b ? b : a;
        int64_t e = fo_state_uedge_index(st, big, small);
        if (e < 0) return -1;
        for (uint64_t q = st->nb_off[e]; q < st->nb_off[e + 1]; ++q) { if (k >= cap) return -2; out[k++] = st->nb[q]; }
        if (k + 2 > cap) return -2;
        out[k++] = a; out[k++] = b;
    }
    qsort(out, k, sizeof(uint32_t), fo_cmp_u32);
    uint64_t u = 0;
    for (uint64_t i = 0; i < k; ++i) if (u == 0 || out[u - 1] != out[i]) out[u++] = out[i];
    return (int64_t)u;
}

/* Transition (src/lib.rs:200-204): change_edges = ([from,to], add?) */
#define FO_MAX_CHANGES 1024 /* clique_swap on two 16-cliques changes at most 2*16*15 edges */
typedef struct {
    uint32_t n;            /* 0 or 2 for the simple moves, up to order^2 for the clique moves */
    uint32_t edge[FO_MAX_CHANGES][2];
    int add[FO_MAX_CHANGES];
} fo_transition;

typedef struct {
    uint64_t pre[FO_MAXDIM]; int pre_len;
    uint64_t post[FO_MAXDIM]; int post_len;
} fo_counters;

/* apply_transition (src/lib.rs:61-79).  Returns 0, or -1 if the reference's
 * `assert!(*s >= *p)` would fire. */
int fo_state_apply_transition(fo_state *st, const fo_transition *t, fo_counters *c)
{
    static __thread uint32_t norm[2 * FO_MAX_CHANGES];
    for (uint32_t i = 0; i < t->n; ++i) {
        uint32_t a = t->edge[i][0], b = t->edge[i][1];
        norm[2 * i] = a > b ? a : b; norm[2 * i + 1] = a > b ? b : a;
    }
    uint64_t cap = 4; /* endpoints */
    for (uint32_t i = 0; i < t->n; ++i) {
        int64_t e = fo_state_uedge_index(st, norm[2 * i], norm[2 * i + 1]);
        if (e < 0) return -2;
        cap += st->nb_off[e + 1] - st->nb_off[e] + 2;
    }
    uint32_t *nbhd = (uint32_t *)malloc(cap * sizeof(uint32_t));
    int64_t k = fo_state_edgeset_neighborhood(st, norm, t->n, nbhd, cap);
    if (k < 0) { free(nbhd); return -2; }
    fo_graph *sub = fo_graph_subgraph(st->graph, nbhd, (uint32_t)k);
    c->pre_len = fo_graph_flagser_count(sub, c->pre);
    fo_graph_free(sub);
    for (int d = 0; d < c->pre_len && d < st->flag_count_len; ++d) {
        if (st->flag_count[d] < c->pre[d]) { free(nbhd); return -1; }
        st->flag_count[d] -= c->pre[d];
    }
    for (uint32_t i = 0; i < t->n; ++i)
        fo_graph_set_edge(st->graph, t->edge[i][0], t->edge[i][1], t->add[i]);
    sub = fo_graph_subgraph(st->graph, nbhd, (uint32_t)k);
    c->post_len = fo_graph_flagser_count(sub, c->post);
    fo_graph_free(sub);
    if (c->post_len > st->flag_count_len) st->flag_count_len = c->post_len; /* resize(.., 0) */
    for (int d = 0; d < c->post_len; ++d) st->flag_count[d] += c->post[d];
    free(nbhd);
    return 0;
}

/* revert_transition (src/lib.rs:81-95): no counting, reuses (pre, post). */
int fo_state_revert_transition(fo_state *st, const fo_transition *t, const fo_counters *c)
{
    for (uint32_t i = 0; i < t->n; ++i)
        fo_graph_set_edge(st->graph, t->edge[i][0], t->edge[i][1], !t->add[i]);
    for (int d = 0; d < c->post_len && d < st->flag_count_len; ++d) {
        if (st->flag_count[d] < c->post[d]) return -1;
        st->flag_count[d] -= c->post[d];
    }
    if (c->pre_len > st->flag_count_len) st->flag_count_len = c->pre_len;
    for (int d = 0; d < c->pre_len; ++d) st->flag_count[d] += c->pre[d];
    return 0;
}

/* flat-argument wrappers for ctypes */
int fo_state_apply_flat(fo_state *st, uint32_t n, const uint32_t *edges, const int *add,
                        uint64_t *pre, int *pre_len, uint64_t *post, int *post_len)
{
    static __thread fo_transition t; fo_counters c;
    if (n > FO_MAX_CHANGES) return -3;
    t.n = n;
    for (uint32_t i = 0; i < n; ++i) { t.edge[i][0] = edges[2 * i]; t.edge[i][1] = edges[2 * i + 1]; t.add[i] = add[i]; }
    int rc = fo_state_apply_transition(st, &t, &c);
    if (rc) return rc;
    memcpy(pre, c.pre, sizeof c.pre); *pre_len = c.pre_len;
    memcpy(post, c.post, sizeof c.post); *post_len = c.post_len;
    return 0;
}
int fo_state_revert_flat(fo_state *st, uint32_t n, const uint32_t *edges, const int *add,
                         const uint64_t *pre, int pre_len, const uint64_t *post, int post_len)
{
    static __thread fo_transition t; fo_counters c;
    if (n > FO_MAX_CHANGES) return -3;
    t.n = n;
    for (uint32_t i = 0; i < n; ++i) { t.edge[i][0] = edges[2 * i]; t.edge[i][1] = edges[2 * i + 1]; t.add[i] = add[i]; }
    memcpy(c.pre, pre, sizeof c.pre); c.pre_len = pre_len;
    memcpy(c.post, post, sizeof c.post); c.post_len = post_len;
    return fo_state_revert_transition(st, &t, &c);
}

/* ------------------------------------------------------------------------ */
/* Bounds (src/lib.rs:113-161)                                               */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint64_t min[FO_MAXDIM]; int min_len;
    uint64_t max[FO_MAXDIM]; int max_len;
} fo_bounds;

/* target bounds (src/bin/sample.rs:89-95): exact for d<2, f64 floor(s*(1-+r))
 * for d>=2. */
void fo_target_bounds(const uint64_t *flag_count, int len, double relaxation, fo_bounds *b)
{
    memset(b, 0, sizeof *b);
    b->min_len = b->max_len = len;
    for (int d = 0; d < len; ++d) {
        if (d < 2) { b->min[d] = b->max[d] = flag_count[d]; }
        else {
            b->min[d] = (uint64_t)floor((double)flag_count[d] * (1. - relaxation));
            b->max[d] = (uint64_t)floor((double)flag_count[d] * (1. + relaxation));
        }
    }
}

/* Bounds::calculate (src/lib.rs:119-156).  Returns 0 ok; -1 where the
 * reference would panic (index out of range / table overrun). */
int fo_bounds_calculate(const fo_state *initial, const fo_bounds *target, fo_bounds *out,
                        uint64_t *ncliques_out /* FO_MAXDIM, optional */, int *ncliques_len)
{
    const fo_graph *g = initial->graph;
    /* normalized graph: total order on vertices, src/lib.rs:125-129 */
    fo_graph *norm = fo_graph_new(g->n);
    for (uint64_t e = 0; e < initial->n_uedges; ++e)
        fo_graph_add_edge(norm, initial->uedges[2 * e], initial->uedges[2 * e + 1]);
    uint64_t ncl[FO_MAXDIM];
    int ncl_len = fo_graph_flagser_count(norm, ncl); /* src/lib.rs:130 */
    fo_graph_free(norm);
    if (ncliques_out) memcpy(ncliques_out, ncl, sizeof ncl);
    if (ncliques_len) *ncliques_len = ncl_len;

    /* SEO shortcut, src/lib.rs:135-137 */
    if (initial->flag_count_len < 2) return -1; /* reference: index panic on flag_count[1] */
    if (initial->n_uedges == initial->flag_count[1]) {
        memset(out, 0, sizeof *out);
        memcpy(out->min, target->min, sizeof out->min); out->min_len = target->min_len;
        memcpy(out->max, ncl, sizeof ncl); out->max_len = ncl_len;
        return 0;
    }
    *out = *target;
    int len = initial->flag_count_len;
    uint64_t relax_de[FO_MAXDIM];
    if (fo_calc_relax_de(initial->flag_count, len, relax_de) != 0) return -1;
    for (int d = 2; d < len; ++d) {
        if (d >= out->max_len || d >= out->min_len) return -1;
        uint64_t f = fo_binomial((uint64_t)len - 2, (uint64_t)d - 1); /* src/lib.rs:144 */
        uint64_t relax = relax_de[d] * f;
        uint64_t a = out->min[d] + relax;
        out->max[d] = a > out->max[d] ? a : out->max[d];          /* :148 */
        uint64_t b = out->max[d] - relax;                          /* wrapping, :149 */
        out->min[d] = b < out->min[d] ? b : out->min[d];
    }
    if (out->max_len < 3) return -1;  /* reference: flag_count_max[2] index panic, :151 */
    out->max[2] = UINT64_MAX;          /* :151 */
    if (out->max_len >= FO_MAXDIM) return -1;
    out->max[out->max_len++] = 10;     /* :152 */
    return 0;
}

/* Bounds::check (src/lib.rs:157-160) */
int fo_bounds_check(const fo_bounds *b, const uint64_t *flag_count, int len)
{
    return fo_all_le(b->min, b->min_len, flag_count, len, 0) &&
           fo_all_le(flag_count, len, b->max, b->max_len, 0);
}

/* sample_distance (src/bin/sample.rs:102) */
uint64_t fo_default_sample_distance(uint64_t nedges)
{
    double e = (double)nedges;
    return (uint64_t)ceil(2. * e * log2(e));
}

/* ------------------------------------------------------------------------ */
/* MCMCSampler (src/lib.rs:163-198), one chain                               */
/* ------------------------------------------------------------------------ */
typedef struct {
    fo_state *state;
    fo_bounds bounds;
    uint64_t cum[4];          /* cumulative move thresholds scaled to 2^32 */
    uint64_t sample_distance;
    uint64_t sampled, accepted;
    /* RNG: Philox key = seed, counter = (step_lo, step_hi, chain, sub) */
    uint64_t seed; uint32_t chain_id;
    /* reciprocal-pair slot list (DESIGN.md draw spec) */
    uint64_t n_double; uint64_t *dbl;
    /* stats */
    uint64_t n_empty;     /* proposals with an empty transition */
    uint64_t n_flip, n_dmove; /* non-empty proposals by kind */
    uint64_t sum_k;       /* sum over evaluated edges of |N(a) cap N(b)| (dedup'd) */
    /* clique moves (src/lib.rs:214-290) */
    uint64_t cumo[FO_MAXDIM];  /* clique_order_distribution (src/bin/sample.rs:87-88) as 2^32-scaled thresholds */
    uint64_t n_cperm, n_cswap; /* non-empty clique_permute / clique_swap proposals */
    uint64_t n_changes;        /* directed edges changed by clique moves (proposed) */
} fo_chain;

void fo_chain_free(fo_chain *c);

#define FO_MAX_SUB 32 /* philox blocks tried for the single edge of a double-edge move */

/* weights -> cumulative u32-scaled thresholds.  Same formula as the product
 * (DESIGN.md draw spec): cum[i] = floor(2^32 * (w0+..+wi)/total), last = 2^32. */
void fo_move_thresholds(const double w[4], uint64_t cum[4])
{
    double total = w[0] + w[1] + w[2] + w[3];
    double acc = 0;
    for (int i = 0; i < 4; ++i) {
        acc += w[i];
        cum[i] = (uint64_t)floor(4294967296.0 * (acc / total));
    }
    /* the last move with non-zero weight absorbs the rounding remainder, so a
     * zero-weight move can never be picked */
    int last = 0;
    for (int i = 0; i < 4; ++i) if (w[i] > 0.0) last = i;
    for (int i = last; i < 4; ++i) cum[i] = 4294967296ull;
}

fo_chain *fo_chain_new(const fo_graph *g, const fo_bounds *bounds, const double weights[4],
                       uint64_t sample_distance, uint64_t seed, uint32_t chain_id)
{
    fo_chain *c = (fo_chain *)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->state = fo_state_new(g);
    if (!c->state) { free(c); return NULL; }
    c->bounds = *bounds;
    fo_move_thresholds(weights, c->cum);
    c->sample_distance = sample_distance;
    c->seed = seed; c->chain_id = chain_id;
    if (weights[2] > 0.0 || weights[3] > 0.0) {
        /* clique_order_distribution: weights (#cliques of the order)^0.2 (src/bin/sample.rs:87-88) */
        if (fo_state_ensure_cliques(c->state)) { fo_chain_free(c); return NULL; }
        double tot = 0, acc = 0;
        int last = 0;
        for (int o = 0; o < c->state->cl_orders; ++o) {
            double wgt = pow((double)c->state->cl_count[o], 0.2);
            tot += wgt;
            if (wgt > 0.0) last = o;
        }
        for (int o = 0; o < c->state->cl_orders; ++o) {
            acc += pow((double)c->state->cl_count[o], 0.2);
            c->cumo[o] = (uint64_t)floor(4294967296.0 * (acc / tot));
        }
        for (int o = last; o < FO_MAXDIM; ++o) c->cumo[o] = 4294967296ull;
    }
    /* reciprocal pairs in ascending undirected-edge order */
    uint64_t nd = 0;
    for (uint64_t e = 0; e < c->state->n_uedges; ++e) {
        uint32_t a = c->state->uedges[2 * e], b = c->state->uedges[2 * e + 1];
        if (fo_graph_has_edge(c->state->graph, a, b) && fo_graph_has_edge(c->state->graph, b, a)) ++nd;
    }
    c->n_double = nd;
    c->dbl = (uint64_t *)malloc((nd ? nd : 1) * sizeof(uint64_t));
    nd = 0;
    for (uint64_t e = 0; e < c->state->n_uedges; ++e) {
        uint32_t a = c->state->uedges[2 * e], b = c->state->uedges[2 * e + 1];
        if (fo_graph_has_edge(c->state->graph, a, b) && fo_graph_has_edge(c->state->graph, b, a)) c->dbl[nd++] = e;
    }
    return c;
}

void fo_chain_free(fo_chain *c)
{
    if (!c) return;
    fo_state_free(c->state);
    free(c->dbl);
    free(c);
}

fo_state *fo_chain_state(fo_chain *c) { return c->state; }
void fo_chain_stats(const fo_chain *c, uint64_t out[9])
{
    out[0] = c->sampled; out[1] = c->accepted; out[2] = c->n_empty;
    out[3] = c->n_flip; out[4] = c->n_dmove; out[5] = c->sum_k;
    out[6] = c->n_cperm; out[7] = c->n_cswap; out[8] = c->n_changes;
}
uint64_t fo_chain_n_double(const fo_chain *c) { return c->n_double; }
const uint64_t *fo_chain_dbl(const fo_chain *c) { return c->dbl; }

static uint64_t fo_dedup_k(const fo_state *st, uint64_t e)
{
    uint64_t k = 0;
    for (uint64_t q = st->nb_off[e]; q < st->nb_off[e + 1]; ++q)
        if (q == st->nb_off[e] || st->nb[q] != st->nb[q - 1]) ++k;
    return k;
}

/* 32-bit words for the shuffles of the clique moves: Philox blocks sub = 2, 3, ... */
typedef struct { uint32_t key[2]; uint32_t ctr[4]; uint32_t w[4]; int pos; } fo_words;
static uint32_t fo_next_word(fo_words *s)
{
    if (s->pos == 4) { fo_philox4x32_10(s->ctr, s->key, s->w); s->ctr[3]++; s->pos = 0; }
    return s->w[s->pos++];
}
/* random_perm (src/util.rs:28-32): l..h shuffled.  Fisher-Yates from the top,
 * j = floor(word * (i+1) / 2^32)  (rand's `shuffle`; draw spec in DESIGN.md). */
static void fo_random_perm(uint32_t l, uint32_t h, fo_words *ws, uint32_t *out)
{
    uint32_t len = h - l;
    for (uint32_t i = 0; i < len; ++i) out[i] = l + i;
    for (uint32_t i = len; i-- > 1;) {
        uint32_t j = (uint32_t)(((uint64_t)fo_next_word(ws) * (uint64_t)(i + 1)) >> 32);
        uint32_t tmp = out[i]; out[i] = out[j]; out[j] = tmp;
    }
}
static int fo_cmp_edge(const void *a, const void *b)
{
    const uint32_t *x = (const uint32_t *)a, *y = (const uint32_t *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
    return 0;
}
static uint32_t fo_sort_dedup_edges(uint32_t (*e)[2], uint32_t n)
{
    qsort(e, n, sizeof e[0], fo_cmp_edge);
    uint32_t u = 0;
    for (uint32_t i = 0; i < n; ++i) if (u == 0 || fo_cmp_edge(e[u - 1], e[i]) != 0) { e[u][0] = e[i][0]; e[u][1] = e[i][1]; ++u; }
    return u;
}
static int fo_cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* One proposal = one iteration of MCMCSampler::next's loop (src/lib.rs:182-192).
 * Returns 0, or a negative code where the reference would panic. */
static int fo_chain_propose(fo_chain *c)
{
    fo_state *st = c->state;
    const uint64_t U = st->n_uedges, D = c->n_double, M = U + D;
    uint32_t key[2] = { (uint32_t)c->seed, (uint32_t)(c->seed >> 32) };
    uint32_t ctr[4] = { (uint32_t)c->sampled, (uint32_t)(c->sampled >> 32), c->chain_id, 0 };
    uint32_t w[4];
    fo_philox4x32_10(ctr, key, w);

    /* Transition::random_move (src/lib.rs:207-212): WeightedIndex pick */
    int move = 0;
    while (move < 3 && (uint64_t)w[0] >= c->cum[move]) ++move;
    int coin = (int)(w[1] & 1u);
    uint64_t x64 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);

    static __thread fo_transition t; t.n = 0;
    uint64_t dbl_slot = 0, new_double_edge = 0; int is_dmove = 0;
    int move_kind = move;                      /* 2 = clique_permute, 3 = clique_swap */
    static __thread uint64_t touched[FO_MAX_CHANGES];
    static __thread uint8_t was_double[FO_MAX_CHANGES];
    uint32_t n_touched = 0;

    if (move == 0) {
        /* single_edge_flip (src/lib.rs:292-299): uniform directed edge; flip
         * iff the reverse is absent.  Index r in [0,U+D): r<U names undirected
         * edge r (its present direction, or its big->small direction when
         * reciprocal); r>=U names the small->big direction of a reciprocal
         * pair.  Either way a reciprocal pair yields the empty transition. */
        if (M > 0) {
            uint64_t r = fo_mulhi64(x64, M);
            if (r < U) {
                uint32_t a = st->uedges[2 * r], b = st->uedges[2 * r + 1];
                int fwd = fo_graph_has_edge(st->graph, a, b), bwd = fo_graph_has_edge(st->graph, b, a);
                if (!(fwd && bwd)) {
                    uint32_t from = fwd ? a : b, to = fwd ? b : a;
                    t.n = 2;
                    t.edge[0][0] = from; t.edge[0][1] = to; t.add[0] = 0;
                    t.edge[1][0] = to; t.edge[1][1] = from; t.add[1] = 1;
                    c->sum_k += fo_dedup_k(st, r);
                }
            }
        }
    } else if (move == 1) {
        /* double_edge_move (src/lib.rs:304-325) */
        if (D > 0) {
            dbl_slot = fo_mulhi64(x64, D);
            uint64_t ed = c->dbl[dbl_slot];
            uint32_t x = st->uedges[2 * ed], y = st->uedges[2 * ed + 1];
            /* rejection-sample a single edge: uniform directed edge, retry
             * while its reverse exists (src/lib.rs:308-313).  Bounded (the
             * reference spins forever when no single edge exists, :307). */
            int found = 0; uint64_t r = 0;
            for (uint32_t sub = 1; sub <= FO_MAX_SUB && !found; ++sub) {
                uint32_t c2[4] = { ctr[0], ctr[1], ctr[2], sub }, v[4];
                fo_philox4x32_10(c2, key, v);
                for (int h = 0; h < 2 && !found; ++h) {
                    uint64_t y64 = (uint64_t)v[2 * h] | ((uint64_t)v[2 * h + 1] << 32);
                    uint64_t rr = fo_mulhi64(y64, M);
                    if (rr >= U) continue;
                    uint32_t a = st->uedges[2 * rr], b = st->uedges[2 * rr + 1];
                    if (fo_graph_has_edge(st->graph, a, b) && fo_graph_has_edge(st->graph, b, a)) continue;
                    r = rr; found = 1;
                }
            }
            if (found) {
                uint32_t ua = st->uedges[2 * r], ub = st->uedges[2 * r + 1];
                int fwd = fo_graph_has_edge(st->graph, ua, ub);
                uint32_t a = fwd ? ua : ub, b = fwd ? ub : ua; /* a->b is the single edge */
                t.n = 2;
                t.edge[0][0] = b; t.edge[0][1] = a; t.add[0] = 1;   /* ([b,a], true) */
                if (coin) { t.edge[1][0] = x; t.edge[1][1] = y; }    /* gen_bool(0.5): double_edge as is */
                else      { t.edge[1][0] = y; t.edge[1][1] = x; }    /* or reversed */
                t.add[1] = 0;
                is_dmove = 1; new_double_edge = r;
                c->sum_k += fo_dedup_k(st, r) + fo_dedup_k(st, ed);
            }
        }
    } else {
        /* clique_permute (src/lib.rs:214-232) / clique_swap (:234-290) */
        if (fo_state_ensure_cliques(st)) return -12;
        int oi = 0;
        while (oi < st->cl_orders - 1 && (uint64_t)w[1] >= c->cumo[oi]) ++oi;   /* clique_order_distribution.sample */
        const uint32_t order = (uint32_t)oi + 1;
        const uint64_t cnt_o = st->cl_count[oi];
        if (cnt_o == 0) return -13;
        const uint32_t *m1 = st->cl_flat + st->cl_base[oi] + fo_mulhi64(x64, cnt_o) * order;   /* choose(rng) */
        fo_words ws;
        ws.key[0] = key[0]; ws.key[1] = key[1];
        ws.ctr[0] = ctr[0]; ws.ctr[1] = ctr[1]; ws.ctr[2] = ctr[2]; ws.ctr[3] = 2; ws.pos = 4;
        if (move == 2) {
            uint32_t perm[FO_MAXDIM];
            fo_random_perm(0, order, &ws, perm);
            for (uint32_t i = 0; i < order; ++i)
                for (uint32_t j = 0; j < order; ++j) {
                    int pre = fo_graph_has_edge(st->graph, m1[perm[i]], m1[perm[j]]);
                    int post = fo_graph_has_edge(st->graph, m1[i], m1[j]);
                    if (pre != post) {
                        if (t.n >= FO_MAX_CHANGES) return -14;
                        t.edge[t.n][0] = m1[perm[i]]; t.edge[t.n][1] = m1[perm[j]]; t.add[t.n] = post; ++t.n;
                    }
                }
            move_kind = 2;
        } else {
            uint32_t c1[4] = { ctr[0], ctr[1], ctr[2], 1 }, v[4];
            fo_philox4x32_10(c1, key, v);
            const uint32_t *m2 = st->cl_flat + st->cl_base[oi] + fo_mulhi64((uint64_t)v[0] | ((uint64_t)v[1] << 32), cnt_o) * order;
            uint32_t d[2 * FO_MAXDIM], n_c = 0, n_d = 0;
            for (uint32_t i = 0; i < order; ++i) {              /* c = vec_intersect(m1, m2) */
                int in2 = 0;
                for (uint32_t j = 0; j < order; ++j) if (m2[j] == m1[i]) in2 = 1;
                if (in2) d[n_d++] = m1[i];
            }
            n_c = n_d;
            for (uint32_t i = 0; i < order; ++i) {              /* m1 - c */
                int inc = 0;
                for (uint32_t j = 0; j < n_c; ++j) if (d[j] == m1[i]) inc = 1;
                if (!inc) d[n_d++] = m1[i];
            }
            for (uint32_t i = 0; i < order; ++i) {              /* m2 - c */
                int inc = 0;
                for (uint32_t j = 0; j < n_c; ++j) if (d[j] == m2[i]) inc = 1;
                if (!inc) d[n_d++] = m2[i];
            }
            const uint32_t n_a = order - n_c;
            uint32_t perm_c[FO_MAXDIM], perm_a[FO_MAXDIM], perm_b[FO_MAXDIM], perm_d[2 * FO_MAXDIM];
            fo_random_perm(0, n_c, &ws, perm_c);
            fo_random_perm(n_c, n_c + n_a, &ws, perm_a);
            fo_random_perm(n_c + n_a, n_d, &ws, perm_b);
            uint32_t q = 0;
            for (uint32_t i = 0; i < n_c; ++i) perm_d[q++] = perm_c[i];
            for (uint32_t i = 0; i < n_d - n_c - n_a; ++i) perm_d[q++] = perm_b[i];
            for (uint32_t i = 0; i < n_a; ++i) perm_d[q++] = perm_a[i];
            static __thread uint32_t new_e[FO_MAX_CHANGES][2], old_e[FO_MAX_CHANGES][2];
            uint32_t nn = 0, no = 0;
            for (int pass = 0; pass < 2; ++pass) {              /* m1's range, then c + m2's range (:256-271) */
                uint32_t idx[2 * FO_MAXDIM], ni = 0;
                if (pass == 0) { for (uint32_t i = 0; i < n_c + n_a; ++i) idx[ni++] = i; }
                else { for (uint32_t i = 0; i < n_c; ++i) idx[ni++] = i; for (uint32_t i = n_c + n_a; i < n_d; ++i) idx[ni++] = i; }
                for (uint32_t a = 0; a < ni; ++a)
                    for (uint32_t b = 0; b < ni; ++b) {
                        uint32_t i = idx[a], j = idx[b];
                        if (fo_graph_has_edge(st->graph, d[i], d[j])) {
                            if (nn >= FO_MAX_CHANGES) return -14;
                            new_e[nn][0] = d[perm_d[i]]; new_e[nn][1] = d[perm_d[j]]; ++nn;
                            old_e[no][0] = d[i]; old_e[no][1] = d[j]; ++no;
                        }
                    }
            }
            nn = fo_sort_dedup_edges(new_e, nn);
            no = fo_sort_dedup_edges(old_e, no);
            uint8_t keep_old[FO_MAX_CHANGES];
            memset(keep_old, 1, no);
            for (uint32_t i = 0; i < nn; ++i) {
                int found = 0;
                for (uint32_t j = 0; j < no; ++j) if (keep_old[j] && fo_cmp_edge(new_e[i], old_e[j]) == 0) { keep_old[j] = 0; found = 1; }
                if (!found) { t.edge[t.n][0] = new_e[i][0]; t.edge[t.n][1] = new_e[i][1]; t.add[t.n] = 1; ++t.n; }
            }
            for (uint32_t j = 0; j < no; ++j)
                if (keep_old[j]) { if (t.n >= FO_MAX_CHANGES) return -14; t.edge[t.n][0] = old_e[j][0]; t.edge[t.n][1] = old_e[j][1]; t.add[t.n] = 0; ++t.n; }
            move_kind = 3;
        }
        /* bookkeeping shared with the device: pairs touched, which of them are reciprocal now */
        for (uint32_t i = 0; i < t.n; ++i) {
            uint32_t a = t.edge[i][0], b = t.edge[i][1];
            int64_t e = fo_state_uedge_index(st, a > b ? a : b, a > b ? b : a);
            if (e < 0) return -2;
            touched[n_touched++] = (uint64_t)e;
            c->sum_k += fo_dedup_k(st, (uint64_t)e);
        }
        c->n_changes += t.n;
        qsort(touched, n_touched, sizeof(uint64_t), fo_cmp_u64);
        uint32_t u = 0;
        for (uint32_t i = 0; i < n_touched; ++i) if (u == 0 || touched[u - 1] != touched[i]) touched[u++] = touched[i];
        n_touched = u;
        for (uint32_t i = 0; i < n_touched; ++i) {
            uint32_t a = st->uedges[2 * touched[i]], b = st->uedges[2 * touched[i] + 1];
            was_double[i] = (uint8_t)(fo_graph_has_edge(st->graph, a, b) && fo_graph_has_edge(st->graph, b, a));
        }
    }

    fo_counters cnt;
    int rc = fo_state_apply_transition(st, &t, &cnt);       /* src/lib.rs:184 */
    if (rc) return rc;
    c->sampled += 1;                                         /* :185 */
    if (t.n == 0) c->n_empty++;
    else if (move_kind == 2) c->n_cperm++;
    else if (move_kind == 3) c->n_cswap++;
    else if (is_dmove) c->n_dmove++;
    else c->n_flip++;
    if (fo_bounds_check(&c->bounds, st->flag_count, st->flag_count_len)) {  /* :186 */
        c->accepted += 1;                                    /* :187 */
        if (is_dmove) c->dbl[dbl_slot] = new_double_edge;
        if (move_kind >= 2 && t.n > 0) {
            /* reciprocal-pair slot list: the i-th pair (ascending id) that stopped being
             * reciprocal hands its slot to the i-th pair that became reciprocal */
            uint64_t lost[FO_MAX_CHANGES], gained[FO_MAX_CHANGES]; uint32_t nl = 0, ng = 0;
            for (uint32_t i = 0; i < n_touched; ++i) {
                uint32_t a = st->uedges[2 * touched[i]], b = st->uedges[2 * touched[i] + 1];
                int is = fo_graph_has_edge(st->graph, a, b) && fo_graph_has_edge(st->graph, b, a);
                if (was_double[i] && !is) lost[nl++] = touched[i];
                if (!was_double[i] && is) gained[ng++] = touched[i];
            }
            if (nl != ng) return -11;
            for (uint32_t i = 0; i < nl; ++i) {
                uint64_t slot = 0;
                while (slot < c->n_double && c->dbl[slot] != lost[i]) ++slot;
                if (slot == c->n_double) return -11;
                c->dbl[slot] = gained[i];
            }
        }
    } else {
        rc = fo_state_revert_transition(st, &t, &cnt);       /* :190 */
        if (rc) return rc;
    }
    return 0;
}

/* run `n` proposals */
int fo_chain_step(fo_chain *c, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) {
        int rc = fo_chain_propose(c);
        if (rc) return rc;
    }
    return 0;
}

/* MCMCSampler::next (src/lib.rs:181-194) */
int fo_chain_next(fo_chain *c) { return fo_chain_step(c, c->sample_distance); }

/* acceptance_ratio (src/lib.rs:195-197) */
double fo_chain_acceptance_ratio(const fo_chain *c) { return (double)c->accepted / (double)c->sampled; }

/* ------------------------------------------------------------------------ */
/* io.rs                                                                      */
/* ------------------------------------------------------------------------ */

/* read_flag_file (src/io.rs:18-35): skip line 1; n = number of non-empty
 * space-separated tokens on line 2; skip line 3; every further line with at
 * least two tokens is an edge i j (further tokens ignored).  Returns NULL on
 * I/O or parse errors (reference: panic). */
fo_graph *fo_read_flag_file(const char *fname)
{
    FILE *f = fopen(fname, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 2);
    if (!buf) { fclose(f); return NULL; }
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return NULL; }
    fclose(f);
    buf[sz] = 0;
    fo_graph *g = NULL;
    char *p = buf, *end = buf + sz;
    int lineno = 0;
    int ok = 1;
    while (p < end) {
        char *nl = memchr(p, '\n', (size_t)(end - p));
        char *le = nl ? nl : end;
        char *next = nl ? nl + 1 : end;
        if (le > p && le[-1] == '\r') --le; /* str::lines() strips \r\n */
        *le = 0;
        if (lineno == 1) {
            uint32_t n = 0;
            for (char *q = p; q < le;) {
                while (q < le && *q == ' ') ++q;
                if (q < le) { ++n; while (q < le && *q != ' ') ++q; }
            }
            g = fo_graph_new(n);
        } else if (lineno >= 3) {
            if (!g) { ok = 0; break; }
            char *tok[2]; int nt = 0;
            for (char *q = p; q < le && nt < 2;) {
                while (q < le && *q == ' ') ++q;
                if (q < le) { tok[nt++] = q; while (q < le && *q != ' ') ++q; if (q < le) *q++ = 0; }
            }
            if (nt == 2) {
                char *e1, *e2;
                errno = 0;
                unsigned long long a = strtoull(tok[0], &e1, 10), b = strtoull(tok[1], &e2, 10);
                if (*e1 || *e2 || e1 == tok[0] || e2 == tok[1] || errno || a >= g->n || b >= g->n) { ok = 0; break; }
                fo_graph_add_edge(g, (uint32_t)a, (uint32_t)b);
            }
        }
        ++lineno;
        p = next;
    }
    free(buf);
    if (!ok || !g) { fo_graph_free(g); return NULL; }
    return g;
}

/* ------------------------------------------------------------------------ */
/* Multi-threaded timing helper for bench.py's cpu_baseline leg: runs          */
/* `nchains` independent chains (ids chain0..), `nprop` proposals each, on     */
/* `nthreads` host threads.  Returns wall seconds of the stepping region only. */
/* ------------------------------------------------------------------------ */
#include <pthread.h>
#include <time.h>

typedef struct { fo_chain **chains; int first, last; uint64_t nprop; int rc; } fo_worker;

static void *fo_worker_main(void *arg)
{
    fo_worker *w = (fo_worker *)arg;
    w->rc = 0;
    for (int i = w->first; i < w->last; ++i) {
        int rc = fo_chain_step(w->chains[i], w->nprop);
        if (rc) { w->rc = rc; break; }
    }
    return NULL;
}

double fo_chains_step_mt(fo_chain **chains, int nchains, uint64_t nprop, int nthreads, int *rc_out)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nchains) nthreads = nchains;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    fo_worker *ws = (fo_worker *)malloc(sizeof(fo_worker) * (size_t)nthreads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nthreads; ++t) {
        ws[t].chains = chains;
        ws[t].first = (int)((long long)nchains * t / nthreads);
        ws[t].last = (int)((long long)nchains * (t + 1) / nthreads);
        ws[t].nprop = nprop;
        pthread_create(&th[t], NULL, fo_worker_main, &ws[t]);
    }
    int rc = 0;
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); if (ws[t].rc) rc = ws[t].rc; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th); free(ws);
    if (rc_out) *rc_out = rc;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
