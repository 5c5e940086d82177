// fcm_host.cpp — host side of libfcm.so: the C ABI declared in include/fcm.h.
//
// Holds what the reference keeps on the host around its hot loop: the Graph
// surface, .flag I/O (src/io.rs:18-48), Bounds
// (src/lib.rs:113-161, src/util.rs:53-105), the static neighbourhood table
// (src/lib.rs:331-356) and sampler construction (src/lib.rs:38-58,
// src/bin/sample.rs:87-104).  All counting and all MCMC stepping is done by
// the HIP kernels in fcm_kernels.hip; there is no CPU fallback for either.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <exception>
#include <mutex>
#include <new>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fcm.h"
#include "fcm_device.hpp"

static_assert(FCM_MAX_COUNTS == FCM_DEV_MAX_COUNTS, "count vector width");
static_assert(FCM_NSTATS == FCM_DEV_NSTATS, "stats width");
// u64 words of the per-chain workspace of fcm_xwide.hpp (FCM_XW_WORDS there): H[1024][16] | stack | counts | list
static const size_t FCM_XW_WORDS_HOST = 1024u * 16u + (16u * 2u * 16u + 16u) + 16u + 512u;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) return fail(FCM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// Nothing throws across the C boundary (include/fcm.h): every entry point that can allocate is a
// function-try-block that ends in one of these.
static int guard_fail()
{
    try { throw; }
    catch (const std::bad_alloc &) { return fail(FCM_ERR_NOMEM, "out of memory"); }
    catch (const std::length_error &e) { return fail(FCM_ERR_NOMEM, "allocation size out of range (%s)", e.what()); }
    catch (const std::exception &e) { return fail(FCM_ERR_INTERNAL, "unexpected exception: %s", e.what()); }
    catch (...) { return fail(FCM_ERR_INTERNAL, "unexpected exception"); }
}
#define FCM_CATCH catch (...) { return guard_fail(); }
#define FCM_CATCH_PTR catch (...) { (void)guard_fail(); return nullptr; }
#define FCM_CATCH_FALSE catch (...) { (void)guard_fail(); return 0; }   // entry points whose int result is a boolean

extern "C" const char *fcm_last_error(void) { return g_err; }
extern "C" const char *fcm_version(void) { return "fcm-amd 0.1 (gfx950)"; }

extern "C" int fcm_device_count(int *count)
try {
    if (!count) return fail(FCM_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    *count = n;
    return FCM_OK;
} FCM_CATCH

static int use_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(FCM_ERR_NO_DEVICE, "no HIP device available (%s); libfcm has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n) return fail(FCM_ERR_NO_DEVICE, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    return FCM_OK;
}

// ---------------------------------------------------------------------------
// Graph
// ---------------------------------------------------------------------------
struct fcm_graph {
    uint32_t n = 0;
    uint32_t stride32 = 32;            // u32 words per row, multiple of 32 (128 B)
    std::vector<uint32_t> rows;        // n * stride32, out-row bitmaps
    uint64_t m = 0;                    // directed edges

    bool has(uint32_t a, uint32_t b) const { return (rows[(size_t)a * stride32 + (b >> 5)] >> (b & 31)) & 1u; }
    void set(uint32_t a, uint32_t b, bool present)
    {
        uint32_t &w = rows[(size_t)a * stride32 + (b >> 5)];
        const uint32_t bit = 1u << (b & 31);
        if (present) { if (!(w & bit)) { w |= bit; ++m; } }
        else { if (w & bit) { w &= ~bit; --m; } }
    }
};

static uint32_t stride_for(uint32_t n);
static uint32_t stride_for(uint32_t n)
{
    uint32_t w = (n + 31) / 32;
    w = (w + 31) / 32 * 32;
    return w ? w : 32;
}

extern "C" int fcm_graph_new_disconnected(uint32_t nnodes, fcm_graph **out)
try {
    if (!out) return fail(FCM_ERR_INVALID, "out is NULL");
    // the kernels address one bitmap through a buffer descriptor with 32-bit byte offsets
    if ((uint64_t)nnodes * stride_for(nnodes) * 4ull >= (1ull << 32))
        return fail(FCM_ERR_UNSUPPORTED, "%u vertices: the row bitmap would exceed 4 GiB", nnodes);
    fcm_graph *g = new (std::nothrow) fcm_graph;
    if (!g) return fail(FCM_ERR_NOMEM, "out of memory");
    g->n = nnodes;
    g->stride32 = stride_for(nnodes);
    try { g->rows.assign((size_t)nnodes * g->stride32, 0u); }
    catch (...) { delete g; return fail(FCM_ERR_NOMEM, "out of memory for %u x %u bitmap", nnodes, nnodes); }
    *out = g;
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_graph_from_edges(uint32_t nnodes, uint64_t nedges, const fcm_node *edges, fcm_graph **out)
try {
    if (nedges && !edges) return fail(FCM_ERR_INVALID, "edges is NULL");
    fcm_graph *g = nullptr;
    int rc = fcm_graph_new_disconnected(nnodes, &g);
    if (rc) return rc;
    for (uint64_t i = 0; i < nedges; ++i) {
        const uint32_t a = edges[2 * i], b = edges[2 * i + 1];
        if (a >= nnodes || b >= nnodes) { delete g; return fail(FCM_ERR_INVALID, "edge %llu = (%u,%u) out of range", (unsigned long long)i, a, b); }
        if (a == b) continue;  // a flag complex has no loops
        g->set(a, b, true);
    }
    *out = g;
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_graph_clone(const fcm_graph *g, fcm_graph **out)
try {
    if (!g || !out) return fail(FCM_ERR_INVALID, "NULL argument");
    fcm_graph *h = new (std::nothrow) fcm_graph(*g);
    if (!h) return fail(FCM_ERR_NOMEM, "out of memory");
    *out = h;
    return FCM_OK;
} FCM_CATCH

extern "C" void fcm_graph_destroy(fcm_graph *g) { delete g; }
extern "C" uint32_t fcm_graph_nnodes(const fcm_graph *g) { return g ? g->n : 0; }
extern "C" uint64_t fcm_graph_nedges(const fcm_graph *g) { return g ? g->m : 0; }

extern "C" int fcm_graph_has_edge(const fcm_graph *g, fcm_node a, fcm_node b)
try {
    if (!g || a >= g->n || b >= g->n) return 0;
    return g->has(a, b) ? 1 : 0;
} FCM_CATCH_FALSE

extern "C" int fcm_graph_set_edge(fcm_graph *g, fcm_node a, fcm_node b, int present)
try {
    if (!g) return fail(FCM_ERR_INVALID, "graph is NULL");
    if (a >= g->n || b >= g->n) return fail(FCM_ERR_INVALID, "edge (%u,%u) out of range", a, b);
    if (a == b) return fail(FCM_ERR_INVALID, "self-loop (%u,%u)", a, b);
    g->set(a, b, present != 0);
    return FCM_OK;
} FCM_CATCH
extern "C" int fcm_graph_add_edge(fcm_graph *g, fcm_node a, fcm_node b) { return fcm_graph_set_edge(g, a, b, 1); }
extern "C" int fcm_graph_remove_edge(fcm_graph *g, fcm_node a, fcm_node b) { return fcm_graph_set_edge(g, a, b, 0); }

template <class F>
static void for_each_edge(const fcm_graph &g, F f)
{
    const uint32_t nw = (g.n + 31) / 32;
    for (uint32_t a = 0; a < g.n; ++a) {
        const uint32_t *row = &g.rows[(size_t)a * g.stride32];
        for (uint32_t w = 0; w < nw; ++w) {
            uint32_t x = row[w];
            while (x) {
                const uint32_t b = w * 32 + (uint32_t)__builtin_ctz(x);
                x &= x - 1;
                f(a, b);
            }
        }
    }
}

extern "C" int fcm_graph_edges(const fcm_graph *g, fcm_node *out, uint64_t cap, uint64_t *m)
try {
    if (!g) return fail(FCM_ERR_INVALID, "graph is NULL");
    uint64_t i = 0;
    for_each_edge(*g, [&](uint32_t a, uint32_t b) {
        if (out && i < cap) { out[2 * i] = a; out[2 * i + 1] = b; }
        ++i;
    });
    if (m) *m = i;
    return FCM_OK;
} FCM_CATCH

// undirected adjacency bitmap: und[a] = out[a] | in[a]
static std::vector<uint32_t> undirected_bitmap(const fcm_graph &g)
{
    std::vector<uint32_t> und(g.rows);
    for_each_edge(g, [&](uint32_t a, uint32_t b) { und[(size_t)b * g.stride32 + (a >> 5)] |= 1u << (a & 31); });
    return und;
}

// undirected_edges(): [a,b] with a > b, ascending (a,b) (src/lib.rs:341,344)
static void undirected_edge_list(const fcm_graph &g, const std::vector<uint32_t> &und, std::vector<uint32_t> &out)
{
    out.clear();
    for (uint32_t a = 0; a < g.n; ++a) {
        const uint32_t *row = &und[(size_t)a * g.stride32];
        const uint32_t nw = a / 32 + 1;
        for (uint32_t w = 0; w < nw; ++w) {
            uint32_t x = row[w];
            while (x) {
                const uint32_t b = w * 32 + (uint32_t)__builtin_ctz(x);
                x &= x - 1;
                if (b < a) { out.push_back(a); out.push_back(b); }
            }
        }
    }
}

extern "C" int fcm_graph_undirected_edges(const fcm_graph *g, fcm_node *out, uint64_t cap, uint64_t *m)
try {
    if (!g) return fail(FCM_ERR_INVALID, "graph is NULL");
    std::vector<uint32_t> und = undirected_bitmap(*g), ue;
    undirected_edge_list(*g, und, ue);
    const uint64_t u = ue.size() / 2;
    if (out) memcpy(out, ue.data(), sizeof(uint32_t) * 2 * (size_t)std::min<uint64_t>(u, cap));
    if (m) *m = u;
    return FCM_OK;
} FCM_CATCH

// ---------------------------------------------------------------------------
// device counting (flagser_count)
// ---------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) { p = nullptr; return fail(FCM_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        return FCM_OK;
    }
    template <class T> T *as() const { return (T *)p; }
};

// counts[0..15]; *len = 1 + highest non-zero dimension.  `rows`/`edges` host.
static int device_count(const uint32_t *rows, uint32_t n, uint32_t stride32, const std::vector<uint32_t> &edges,
                        int device, uint64_t counts[FCM_MAX_COUNTS], int *len)
{
    int rc = use_device(device);
    if (rc) return rc;
    memset(counts, 0, sizeof(uint64_t) * FCM_MAX_COUNTS);
    const uint64_t m = edges.size() / 2;
    counts[0] = n;
    counts[1] = m;
    if (m > 0) {
        DevBuf d_rows, d_edges, d_counts, d_flags, d_xlist, d_xw;
        const uint32_t xcap = 4096;   // edges with more than 256 common out-neighbours a graph may have (second pass, 141 KB of workspace each)
        const size_t row_bytes = (size_t)n * stride32 * sizeof(uint32_t);
        if ((rc = d_rows.alloc(row_bytes))) return rc;
        if ((rc = d_edges.alloc(edges.size() * sizeof(uint32_t)))) return rc;
        if ((rc = d_counts.alloc(sizeof(uint64_t) * FCM_MAX_COUNTS))) return rc;
        if ((rc = d_flags.alloc(sizeof(uint32_t) * 2))) return rc;
        if ((rc = d_xlist.alloc(sizeof(uint32_t) * (1 + xcap)))) return rc;
        HIP_TRY(hipMemset(d_xlist.p, 0, sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(d_rows.p, rows, row_bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_edges.p, edges.data(), edges.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(d_counts.p, 0, sizeof(uint64_t) * FCM_MAX_COUNTS));
        HIP_TRY(hipMemset(d_flags.p, 0, sizeof(uint32_t) * 2));
        FcmCountParams p;
        p.rows = d_rows.as<uint32_t>();
        p.edges = d_edges.as<uint32_t>();
        p.m = m;
        p.counts = d_counts.as<uint64_t>();
        p.flags = d_flags.as<uint32_t>();
        p.n = n;
        p.stride32 = stride32;
        p.xlist = d_xlist.as<uint32_t>();
        p.xcap = xcap;
        p.xw_ws = nullptr;
        int lrc = fcm_launch_count(&p, nullptr);
        if (lrc) return fail(FCM_ERR_HIP, "count kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
        HIP_TRY(hipDeviceSynchronize());
        uint32_t nflagged = 0;
        HIP_TRY(hipMemcpy(&nflagged, d_xlist.p, sizeof nflagged, hipMemcpyDeviceToHost));
        if (nflagged > xcap)
            return fail(FCM_ERR_UNSUPPORTED, "%u directed edges have more than %d common out-neighbours; this build takes at most %u such edges", nflagged, FCM_MAX_COUNT_LOCAL, xcap);
        if (nflagged) {   // second pass: 257 .. 1024 common out-neighbours, masks in a workspace
            if ((rc = d_xw.alloc((size_t)nflagged * FCM_XW_WORDS_HOST * sizeof(uint64_t)))) return rc;
            p.xw_ws = d_xw.as<uint64_t>();
            lrc = fcm_launch_count_xw(&p, nflagged, nullptr);
            if (lrc) return fail(FCM_ERR_HIP, "count kernel (second pass) launch failed: %s", hipGetErrorString((hipError_t)lrc));
            HIP_TRY(hipDeviceSynchronize());
        }
        uint64_t dc[FCM_MAX_COUNTS];
        uint32_t flags[2];
        HIP_TRY(hipMemcpy(dc, d_counts.p, sizeof dc, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(flags, d_flags.p, sizeof flags, hipMemcpyDeviceToHost));
        if (flags[0])
            return fail(FCM_ERR_UNSUPPORTED, "a directed edge has more than %d common out-neighbours; not supported by this build", FCM_MAX_LOCAL);
        if (flags[1])
            return fail(FCM_ERR_UNSUPPORTED, "graph holds simplices of dimension > %d", FCM_MAX_COUNTS - 1);
        for (int d = 2; d < FCM_MAX_COUNTS; ++d) counts[d] = dc[d];
    }
    int l = 0;
    for (int d = 0; d < FCM_MAX_COUNTS; ++d) if (counts[d]) l = d + 1;
    *len = l;
    return FCM_OK;
}

static void edge_list(const fcm_graph &g, std::vector<uint32_t> &edges)
{
    edges.clear();
    edges.reserve((size_t)g.m * 2);
    for_each_edge(g, [&](uint32_t a, uint32_t b) { edges.push_back(a); edges.push_back(b); });
}

extern "C" int fcm_graph_flagser_count(const fcm_graph *g, int device, uint64_t *counts, int cap, int *len)
try {
    if (!g || !counts || !len) return fail(FCM_ERR_INVALID, "NULL argument");
    std::vector<uint32_t> edges;
    edge_list(*g, edges);
    uint64_t c[FCM_MAX_COUNTS];
    int l = 0;
    int rc = device_count(g->rows.data(), g->n, g->stride32, edges, device, c, &l);
    if (rc) return rc;
    if (l > cap) return fail(FCM_ERR_INVALID, "counts buffer too small: need %d, have %d", l, cap);
    for (int d = 0; d < l; ++d) counts[d] = c[d];
    *len = l;
    return FCM_OK;
} FCM_CATCH

static int default_device()
{
    const char *e = getenv("FCM_DEVICE");
    return e ? atoi(e) : 0;
}

// Legacy symbol, src/flagser.rs:7-10.
extern "C" size_t *flagser_count_unweighted(size_t nvertices, size_t nedges, const fcm_node (*edges)[2], size_t *res_size)
try {
    if (res_size) *res_size = 0;
    if (!res_size || nvertices > 0xFFFFFFFFull) { fail(FCM_ERR_INVALID, "bad arguments"); return nullptr; }
    fcm_graph *g = nullptr;
    if (fcm_graph_from_edges((uint32_t)nvertices, nedges, (const fcm_node *)edges, &g)) return nullptr;
    uint64_t c[FCM_MAX_COUNTS];
    int len = 0;
    int rc = fcm_graph_flagser_count(g, default_device(), c, FCM_MAX_COUNTS, &len);
    fcm_graph_destroy(g);
    if (rc) return nullptr;
    size_t *res = (size_t *)malloc(sizeof(size_t) * (size_t)(len ? len : 1));
    if (!res) { fail(FCM_ERR_NOMEM, "out of memory"); return nullptr; }
    for (int d = 0; d < len; ++d) res[d] = (size_t)c[d];
    *res_size = (size_t)len;
    return res;
} FCM_CATCH_PTR

// ---------------------------------------------------------------------------
// .flag I/O (src/io.rs:18-48)
// ---------------------------------------------------------------------------
static void split_spaces(const std::string &line, std::vector<std::string> &tok)
{
    tok.clear();
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && line[i] == ' ') ++i;
        size_t j = i;
        while (j < line.size() && line[j] != ' ') ++j;
        if (j > i) tok.emplace_back(line, i, j - i);
        i = j;
    }
}

extern "C" int fcm_read_flag_file(const char *path, fcm_graph **out)
try {
    if (!path || !out) return fail(FCM_ERR_INVALID, "NULL argument");
    std::ifstream f(path, std::ios::binary);
    if (!f) return fail(FCM_ERR_IO, "could not find .flag input file %s", path);
    std::string line;
    std::vector<std::string> tok;
    fcm_graph *g = nullptr;
    int lineno = 0;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (lineno == 1) {                       // vertex line: count tokens (io.rs:25)
            split_spaces(line, tok);
            int rc = fcm_graph_new_disconnected((uint32_t)tok.size(), &g);
            if (rc) return rc;
        } else if (lineno >= 3) {                // edge lines (io.rs:28-32)
            if (!g) return fail(FCM_ERR_IO, "%s: truncated header", path);
            split_spaces(line, tok);
            if (tok.size() >= 2) {
                char *e1 = nullptr, *e2 = nullptr;
                errno = 0;
                const unsigned long long a = strtoull(tok[0].c_str(), &e1, 10), b = strtoull(tok[1].c_str(), &e2, 10);
                if (errno || *e1 || *e2 || a >= g->n || b >= g->n) {
                    delete g;
                    return fail(FCM_ERR_IO, "%s:%d: bad edge line '%s'", path, lineno + 1, line.c_str());
                }
                if (a != b) g->set((uint32_t)a, (uint32_t)b, true);
            }
        }
        ++lineno;
    }
    if (!g) return fail(FCM_ERR_IO, "%s: no vertex line", path);
    *out = g;
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_save_flag_file(const char *path, const fcm_graph *g)
try {
    if (!path || !g) return fail(FCM_ERR_INVALID, "NULL argument");
    std::ofstream f(path, std::ios::binary);
    if (!f) return fail(FCM_ERR_IO, "Unable to write file %s", path);
    f << "dim 0:\n";
    for (uint32_t i = 0; i < g->n; ++i) f << (i ? " 1" : "1");
    f << "\ndim 1:\n";
    for_each_edge(*g, [&](uint32_t a, uint32_t b) { f << a << ' ' << b << " 1\n"; });  // ascending = sort_unstable (io.rs:42)
    if (!f) return fail(FCM_ERR_IO, "write to %s failed", path);
    return FCM_OK;
} FCM_CATCH

// ---------------------------------------------------------------------------
// Bounds (src/lib.rs:113-161; src/util.rs:53-105; src/bin/sample.rs:89-102)
// ---------------------------------------------------------------------------
static bool all_le(const uint64_t *a, int na, const uint64_t *b, int nb)  // util.rs:53-63, z = 0
{
    const int ml = std::max(na, nb);
    for (int i = 0; i < ml; ++i) {
        const uint64_t l = i < na ? a[i] : 0, r = i < nb ? b[i] : 0;
        if (l > r) return false;
    }
    return true;
}

// util.rs:65-71: the loop is `1..x`, i.e. (x-1)!
static uint64_t ref_factorial(uint64_t x)
{
    uint64_t r = 1;
    for (uint64_t i = 1; i < x; ++i) r *= i;
    return r;
}

// OEIS A058298, n!/(n-k), 1 <= k < n, by rows; first 64 terms (util.rs:98-105)
static std::vector<uint64_t> make_a058298()
{
    std::vector<uint64_t> t;
    uint64_t fact = 1;
    for (uint64_t n = 2; t.size() < 64; ++n) {
        fact *= n;
        for (uint64_t k = 1; k < n && t.size() < 64; ++k) t.push_back(fact / (n - k));
    }
    return t;
}
static const std::vector<uint64_t> &a058298()
{
    static const std::vector<uint64_t> t = make_a058298();   // thread-safe initialisation (distinct handles, distinct threads)
    return t;
}

extern "C" int fcm_target_bounds(const uint64_t *flag_count, int len, double r, fcm_bounds *out)
try {
    if (!flag_count || !out || len < 0 || len > FCM_MAX_COUNTS) return fail(FCM_ERR_INVALID, "bad arguments");
    memset(out, 0, sizeof *out);
    out->min_len = out->max_len = len;
    for (int d = 0; d < len; ++d) {
        if (d < 2) out->flag_count_min[d] = out->flag_count_max[d] = flag_count[d];
        else {
            out->flag_count_min[d] = (uint64_t)std::floor((double)flag_count[d] * (1. - r));
            out->flag_count_max[d] = (uint64_t)std::floor((double)flag_count[d] * (1. + r));
        }
    }
    return FCM_OK;
} FCM_CATCH

static int clique_counts(const fcm_graph &g, int device, uint64_t ncl[FCM_MAX_COUNTS], int *ncl_len, uint64_t *n_undirected)
{
    // normalized graph: edge big->small for every adjacent pair (lib.rs:125-129)
    std::vector<uint32_t> und = undirected_bitmap(g), ue;
    undirected_edge_list(g, und, ue);
    std::vector<uint32_t> rows((size_t)g.n * g.stride32, 0u);
    for (size_t e = 0; e < ue.size(); e += 2) rows[(size_t)ue[e] * g.stride32 + (ue[e + 1] >> 5)] |= 1u << (ue[e + 1] & 31);
    if (n_undirected) *n_undirected = ue.size() / 2;
    return device_count(rows.data(), g.n, g.stride32, ue, device, ncl, ncl_len);
}

extern "C" int fcm_bounds_calculate(const fcm_graph *g, const uint64_t *flag_count, int len, const fcm_bounds *target,
                                    int device, fcm_bounds *out, uint64_t *ncliques, int *ncliques_len)
try {
    if (!g || !flag_count || !target || !out) return fail(FCM_ERR_INVALID, "NULL argument");
    if (len > FCM_MAX_COUNTS) return fail(FCM_ERR_INVALID, "len > %d", FCM_MAX_COUNTS);
    uint64_t ncl[FCM_MAX_COUNTS];
    int ncl_len = 0;
    uint64_t U = 0;
    int rc = clique_counts(*g, device, ncl, &ncl_len, &U);   // lib.rs:130
    if (rc) return rc;
    if (ncliques) { memset(ncliques, 0, sizeof(uint64_t) * (FCM_MAX_COUNTS + 1)); memcpy(ncliques, ncl, sizeof ncl); }
    if (ncliques_len) *ncliques_len = ncl_len;
    if (len < 2) return fail(FCM_ERR_PANIC, "reference indexes flag_count[1] (src/lib.rs:135)");
    if (U == flag_count[1]) {  // SEO case, lib.rs:135-137
        memset(out, 0, sizeof *out);
        memcpy(out->flag_count_min, target->flag_count_min, sizeof out->flag_count_min);
        out->min_len = target->min_len;
        memcpy(out->flag_count_max, ncl, sizeof ncl);
        out->max_len = ncl_len;
        return FCM_OK;
    }
    *out = *target;
    // calc_relax_de, util.rs:79-93
    const std::vector<uint64_t> &A = a058298();
    uint64_t relax_de[FCM_MAX_COUNTS];
    for (int d = 0; d < len; ++d) {
        size_t ind = 1;
        uint64_t best = 0;
        bool have = false;
        for (;;) {
            if (ind >= A.size()) return fail(FCM_ERR_PANIC, "calc_relax_de table overrun: count[%d]=%llu (src/util.rs:84)", d, (unsigned long long)flag_count[d]);
            if (!(A[ind] < flag_count[d])) break;
            const uint64_t lost = A[ind] - A[ind - 1];
            if (!have || lost > best) { best = lost; have = true; }
            ++ind;
        }
        relax_de[d] = std::min(have ? best : (uint64_t)1, ref_factorial((uint64_t)d + 1));
    }
    for (int d = 2; d < len; ++d) {
        if (d >= out->max_len || d >= out->min_len) return fail(FCM_ERR_PANIC, "target bounds shorter than flag_count (src/lib.rs:148)");
        const uint64_t nn = (uint64_t)len - 2, kk = (uint64_t)d - 1;    // binomial(len-2, d-1), lib.rs:144
        if (kk > nn) return fail(FCM_ERR_PANIC, "binomial underflow (src/util.rs:76)");
        const uint64_t f = ref_factorial(nn) / (ref_factorial(kk) * ref_factorial(nn - kk));
        const uint64_t relax = relax_de[d] * f;
        out->flag_count_max[d] = std::max(out->flag_count_min[d] + relax, out->flag_count_max[d]);   // :148
        out->flag_count_min[d] = std::min(out->flag_count_max[d] - relax, out->flag_count_min[d]);   // :149
    }
    if (out->max_len < 3) return fail(FCM_ERR_PANIC, "flag_count_max[2] out of range (src/lib.rs:151)");
    out->flag_count_max[2] = UINT64_MAX;                // :151
    out->flag_count_max[out->max_len++] = 10;           // :152
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_bounds_check(const fcm_bounds *b, const uint64_t *flag_count, int len)
try {
    if (!b || !flag_count) return 0;
    return all_le(b->flag_count_min, b->min_len, flag_count, len) && all_le(flag_count, len, b->flag_count_max, b->max_len);
} FCM_CATCH_FALSE

extern "C" uint64_t fcm_default_sample_distance(uint64_t nedges)
{
    // E = 0: 2*0*log2(0) is NaN, and the reference's `NaN as usize` is 0 (src/bin/sample.rs:102); E = 1 gives 0 too
    const double e = (double)nedges;
    const double d = std::ceil(2. * e * std::log2(e));
    if (!(d > 0.0) || !std::isfinite(d)) return 0;
    return d >= 18446744073709551615.0 ? UINT64_MAX : (uint64_t)d;
}

// ---------------------------------------------------------------------------
// Maximal cliques of pr(G): `compute_maximal_cliques` (external crate, called
// at src/lib.rs:41) and the bucketing by order (src/lib.rs:42-49).  Tomita-
// style pivoting on bitset rows.  Layout handed to the device: bucket o-1 =
// all maximal cliques of o vertices, each clique ascending, the bucket in
// lexicographic order (the reference's enumeration order is unknown and, as
// cliques are drawn uniformly, immaterial).
// ---------------------------------------------------------------------------
struct CliqueTable {
    int orders = 0;                                   // cliques_by_order.len()
    uint64_t count[FCM_MAX_COUNTS] = {0}, base[FCM_MAX_COUNTS] = {0};
    std::vector<uint32_t> flat;
};

namespace {
struct CliqueFinder {
    const std::vector<uint32_t> &und;                 // undirected bitmap, stride32 words per row
    uint32_t n, stride, nw;
    std::vector<std::vector<uint32_t>> byorder;       // flat vertex lists per order
    std::vector<uint32_t> R;
    bool too_deep = false;

    CliqueFinder(const std::vector<uint32_t> &u, uint32_t n_, uint32_t stride_) : und(u), n(n_), stride(stride_), nw((n_ + 31) / 32) {}
    const uint32_t *row(uint32_t v) const { return &und[(size_t)v * stride]; }

    void expand(std::vector<uint32_t> &P, std::vector<uint32_t> &X)
    {
        bool anyP = false, anyX = false;
        for (uint32_t w = 0; w < nw; ++w) { anyP |= P[w] != 0; anyX |= X[w] != 0; }
        if (!anyP) {
            if (!anyX) {
                if (R.size() > FCM_MAX_COUNTS) { too_deep = true; return; }
                if (byorder.size() < R.size()) byorder.resize(R.size());
                std::vector<uint32_t> c(R);
                std::sort(c.begin(), c.end());
                byorder[R.size() - 1].insert(byorder[R.size() - 1].end(), c.begin(), c.end());
            }
            return;
        }
        if (too_deep || R.size() > FCM_MAX_COUNTS) { too_deep = true; return; }
        // pivot with the most neighbours inside P
        int best = -1;
        uint32_t pivot = 0;
        for (uint32_t w = 0; w < nw; ++w) {
            uint32_t x = P[w] | X[w];
            while (x) {
                const uint32_t u = w * 32 + (uint32_t)__builtin_ctz(x);
                x &= x - 1;
                const uint32_t *ru = row(u);
                int c = 0;
                for (uint32_t q = 0; q < nw; ++q) c += __builtin_popcount(P[q] & ru[q]);
                if (c > best) { best = c; pivot = u; }
            }
        }
        const uint32_t *rp = row(pivot);
        std::vector<uint32_t> nP(nw), nX(nw);
        for (uint32_t w = 0; w < nw; ++w) {
            uint32_t cand = P[w] & ~rp[w];
            while (cand) {
                const uint32_t v = w * 32 + (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1;
                const uint32_t *rv = row(v);
                for (uint32_t q = 0; q < nw; ++q) { nP[q] = P[q] & rv[q]; nX[q] = X[q] & rv[q]; }
                R.push_back(v);
                expand(nP, nX);
                R.pop_back();
                P[w] &= ~(1u << (v & 31));
                X[w] |= 1u << (v & 31);
            }
        }
    }
};
}  // namespace

static int maximal_cliques(const fcm_graph &g, const std::vector<uint32_t> &und, CliqueTable &out)
{
    CliqueFinder f(und, g.n, g.stride32);
    std::vector<uint32_t> P(f.nw), X(f.nw);
    for (uint32_t v = 0; v < g.n; ++v) {   // v = smallest vertex of the clique
        const uint32_t *rv = f.row(v);
        for (uint32_t w = 0; w < f.nw; ++w) {
            const uint32_t below = w < (v >> 5) ? 0xFFFFFFFFu : (w == (v >> 5) ? ((1u << (v & 31)) - 1u) : 0u);
            X[w] = rv[w] & below;
            P[w] = rv[w] & ~below;
        }
        f.R.assign(1, v);
        f.expand(P, X);
        if (f.too_deep) return fail(FCM_ERR_UNSUPPORTED, "pr(G) has a clique of more than %d vertices", FCM_MAX_COUNTS);
    }
    out.orders = (int)f.byorder.size();
    uint64_t total = 0;
    for (int o = 1; o <= out.orders; ++o) {
        std::vector<uint32_t> &b = f.byorder[o - 1];
        const size_t cnt = b.size() / o;
        // lexicographic order of the bucket
        std::vector<size_t> idx(cnt);
        for (size_t i = 0; i < cnt; ++i) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) {
            return std::lexicographical_compare(b.begin() + x * o, b.begin() + (x + 1) * o, b.begin() + y * o, b.begin() + (y + 1) * o);
        });
        out.count[o - 1] = cnt;
        out.base[o - 1] = total;
        for (size_t i = 0; i < cnt; ++i) out.flat.insert(out.flat.end(), b.begin() + idx[i] * o, b.begin() + (idx[i] + 1) * o);
        total += (uint64_t)cnt * o;
        std::vector<uint32_t>().swap(b);
    }
    return FCM_OK;
}

// ---------------------------------------------------------------------------
// Sampler
// ---------------------------------------------------------------------------
struct fcm_sampler {
    int device = 0;
    fcm_sampler_config cfg{};
    fcm_bounds bounds{};
    fcm_sampler_info info{};
    FcmStepParams params{};
    int maxt_variant = 6, maxnw_variant = 1;
    uint32_t n = 0, stride32 = 0;
    std::vector<uint32_t> ue;          // [U][2] big, small
    // device buffers
    DevBuf d_etab, d_nb, d_rows, d_dbl, d_counts, d_stats, d_clq, d_clq_pairs, d_efirst, d_slot_of, d_dbg, d_xw;
    bool clique_moves = false;
    bool sparse = false;               // per chain two bits per adjacent pair instead of row bitmaps (d_rows holds them; include/fcm.h, fcm_sampler_info)
    uint32_t bits_stride = 0;          // ... u32 words per chain
    bool use_cq = false;               // move mixes with clique moves on the fcm_step_cq kernel (<= 8 count entries)
    // host copies of the static tables, fetched on the first use of the State API (apply/revert/edgeset_neighborhood)
    std::vector<FcmEdgeEntry> h_etab;
    std::vector<uint32_t> h_nb;
    bool h_tables = false;
    DevBuf d_tr_list, d_tr_out, d_tr_chg;   // scratch of that API
    size_t tr_list_cap = 0, tr_out_cap = 0, tr_chg_cap = 0;
    DevBuf d_bt_pair, d_bt_ops, d_bt_pre, d_bt_post, d_bt_lens, d_bt_status, d_bt_x, d_bt_out;   // the batched State API (one transition per chain), allocated on its first use
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;

    ~fcm_sampler()
    {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (own_stream) (void)hipStreamDestroy(own_stream);
    }
};

static void move_thresholds(const double w[4], uint64_t cum[4])
{
    const double total = w[0] + w[1] + w[2] + w[3];
    double acc = 0;
    for (int i = 0; i < 4; ++i) {
        acc += w[i];
        cum[i] = (uint64_t)std::floor(4294967296.0 * (acc / total));
    }
    int last = 0;
    for (int i = 0; i < 4; ++i) if (w[i] > 0.0) last = i;
    for (int i = last; i < 4; ++i) cum[i] = 4294967296ull;
}

extern "C" int fcm_sampler_create(const fcm_graph *g, const fcm_bounds *bounds, const fcm_sampler_config *cfg, fcm_sampler **out)
try {
    if (!g || !bounds || !cfg || !out) return fail(FCM_ERR_INVALID, "NULL argument");
    if (cfg->n_chains == 0) return fail(FCM_ERR_INVALID, "n_chains must be > 0");
    for (int i = 0; i < 4; ++i)
        if (!(cfg->move_weights[i] >= 0.0) || !std::isfinite(cfg->move_weights[i]))
            return fail(FCM_ERR_INVALID, "move weight %d is negative or not finite", i);
    if (cfg->move_weights[0] + cfg->move_weights[1] + cfg->move_weights[2] + cfg->move_weights[3] <= 0.0)
        return fail(FCM_ERR_INVALID, "all move weights are zero");
    if (bounds->min_len < 0 || bounds->min_len > FCM_MAX_COUNTS || bounds->max_len < 0 || bounds->max_len > FCM_MAX_COUNTS)
        return fail(FCM_ERR_UNSUPPORTED, "bounds longer than %d entries", FCM_MAX_COUNTS);
    int rc = use_device(cfg->device);
    if (rc) return rc;

    fcm_sampler *s = new (std::nothrow) fcm_sampler;
    if (!s) return fail(FCM_ERR_NOMEM, "out of memory");
    struct Guard { fcm_sampler *s; ~Guard() { delete s; } } guard{s};
    s->device = cfg->device;
    s->cfg = *cfg;
    s->bounds = *bounds;
    s->n = g->n;
    s->stride32 = g->stride32;

    // --- static tables: undirected edges and their common neighbourhoods ---
    std::vector<uint32_t> und = undirected_bitmap(*g);
    undirected_edge_list(*g, und, s->ue);
    const uint64_t U = s->ue.size() / 2;
    if (U >= 0xFFFFFFFFull) return fail(FCM_ERR_UNSUPPORTED, "too many undirected edges");
    std::vector<FcmEdgeEntry> etab((size_t)U);
    std::vector<uint32_t> nb;
    std::vector<uint32_t> dbl0;
    const uint32_t nw = (g->n + 31) / 32;
    uint32_t kmax = 0;
    uint64_t ksum = 0;
    {
        // compute_edge_neighborhoods (src/lib.rs:331-356; rayon there, host threads here): N(a) cap N(b) for every
        // adjacent pair.  Per pair the cheaper of two ways: AND of the two undirected bitmap rows (n/32 words), or the
        // reference's own two-pointer intersection of the sorted adjacency lists (src/util.rs:5-26; deg(a) + deg(b)
        // steps) -- on a sparse graph of 30000 vertices the rows are 938 words and the lists 67 entries.
        std::vector<uint64_t> adj_off((size_t)g->n + 1, 0);
        for (uint64_t e = 0; e < U; ++e) { adj_off[s->ue[2 * e] + 1]++; adj_off[s->ue[2 * e + 1] + 1]++; }
        for (uint32_t v = 0; v < g->n; ++v) adj_off[v + 1] += adj_off[v];
        std::vector<uint32_t> adj((size_t)adj_off[g->n]);
        {
            std::vector<uint64_t> fill(adj_off.begin(), adj_off.end() - 1);
            // ue is ascending in (big, small): a vertex first meets its smaller neighbours (as `big`) in ascending order, its
            // larger ones (as `small`) in ascending order of big too -- but the two runs interleave, so sort each list
            for (uint64_t e = 0; e < U; ++e) { const uint32_t a = s->ue[2 * e], b = s->ue[2 * e + 1]; adj[fill[a]++] = b; adj[fill[b]++] = a; }
        }
        const unsigned hw = std::thread::hardware_concurrency();
        const unsigned nthreads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(hw ? hw : 1, 16), U / 4096 + 1));
        // Shares 0 .. nthreads-1 of `fn`, one host thread each.  A thread that cannot be started (std::system_error at the
        // thread limit, bad_alloc) must not leave joinable threads behind -- the vector's destructor would call
        // std::terminate, across the C boundary: the shares without a thread run inline, and an exception out of a share
        // itself is carried to this thread and rethrown once every thread has been joined.
        auto run_parallel = [&](auto &&fn) {
            std::vector<std::thread> th;
            std::exception_ptr err;
            std::mutex err_mu;
            auto guarded = [&](unsigned t) {
                try { fn(t); }
                catch (...) { std::lock_guard<std::mutex> lk(err_mu); if (!err) err = std::current_exception(); }
            };
            unsigned started = 1;
            try {
                th.reserve(nthreads);
                for (; started < nthreads; ++started) th.emplace_back(guarded, started);
            } catch (...) { /* the shares from `started` on run inline below */ }
            guarded(0u);
            for (unsigned t = started; t < nthreads; ++t) guarded(t);
            for (auto &x : th) x.join();
            if (err) std::rethrow_exception(err);
        };
        run_parallel([&](unsigned t) {
            for (uint64_t v = (uint64_t)g->n * t / nthreads; v < (uint64_t)g->n * (t + 1) / nthreads; ++v)
                std::sort(adj.begin() + adj_off[v], adj.begin() + adj_off[v + 1]);
        });
        auto common = [&](uint32_t a, uint32_t b, uint32_t *out) -> uint32_t {   // out == nullptr: count only
            const uint64_t da = adj_off[a + 1] - adj_off[a], db = adj_off[b + 1] - adj_off[b];
            uint32_t k = 0;
            if (da + db < 2ull * nw) {
                const uint32_t *pa = &adj[adj_off[a]], *ea = pa + da, *pb = &adj[adj_off[b]], *eb = pb + db;
                while (pa < ea && pb < eb) {
                    if (*pa < *pb) ++pa;
                    else if (*pb < *pa) ++pb;
                    else { if (out) out[k] = *pa; ++k; ++pa; ++pb; }
                }
            } else {
                const uint32_t *ra = &und[(size_t)a * g->stride32], *rb = &und[(size_t)b * g->stride32];
                for (uint32_t w = 0; w < nw; ++w) {
                    uint32_t x = ra[w] & rb[w];
                    while (x) { if (out) out[k] = w * 32 + (uint32_t)__builtin_ctz(x); ++k; x &= x - 1; }
                }
            }
            return k;
        };
        run_parallel([&](unsigned t) {   // pass 1: sizes
            for (uint64_t e = U * t / nthreads; e < U * (t + 1) / nthreads; ++e) {
                etab[e].big = s->ue[2 * e]; etab[e].small = s->ue[2 * e + 1];
                etab[e].k = common(etab[e].big, etab[e].small, nullptr);
            }
        });
        for (uint64_t e = 0; e < U; ++e) { kmax = std::max(kmax, etab[e].k); ksum += etab[e].k; }
        // Sparse state: rows much longer than what a build reads of them -- more than 1024 vertices, local sets
        // of at most 11 vertices, two common neighbours on average at most -- and the simple moves (the multi-wave kernel's
        // domain; whether that kernel runs at all is settled below, with the count entries known).  FCM_SPARSE=0 / 1 overrides.
        {
            const bool simple = !(cfg->move_weights[2] > 0.0 || cfg->move_weights[3] > 0.0);
            bool want = g->stride32 > 32 && U > 0 && (double)ksum <= 2.0 * (double)U;
            if (const char *e = getenv("FCM_SPARSE")) want = atoi(e) != 0;
            s->sparse = want && simple && U > 0 && kmax + 2 <= 11;
        }
        uint64_t total = 0;
        for (uint64_t e = 0; e < U; ++e) {
            if (total > 0xFFFFFF00ull) return fail(FCM_ERR_UNSUPPORTED, "neighbourhood table exceeds 2^32 entries");
            etab[e].nb_off = (uint32_t)total;
            total += etab[e].k;
            if (s->sparse && etab[e].k) {   // behind the list, from an even word on: two words per local pair
                const uint64_t sl = (uint64_t)etab[e].k + 2;
                total += (total & 1) + sl * (sl - 1);
            }
            if (g->has(etab[e].big, etab[e].small) && g->has(etab[e].small, etab[e].big)) dbl0.push_back((uint32_t)e);
        }
        if (total > 0xFFFFFF00ull) return fail(FCM_ERR_UNSUPPORTED, "neighbourhood table exceeds 2^32 entries");
        nb.assign((size_t)total, 0u);
        auto pair_id = [&](uint32_t a, uint32_t b) -> uint32_t {   // etab index of {a, b}, 0xFFFFFFFF if not adjacent (ue is ascending (big, small))
            const uint32_t big = std::max(a, b), small = std::min(a, b);
            uint64_t lo = 0, hi = U;
            while (lo < hi) {
                const uint64_t mid = (lo + hi) / 2;
                if (s->ue[2 * mid] < big || (s->ue[2 * mid] == big && s->ue[2 * mid + 1] < small)) lo = mid + 1; else hi = mid;
            }
            return (lo < U && s->ue[2 * lo] == big && s->ue[2 * lo + 1] == small) ? (uint32_t)lo : 0xFFFFFFFFu;
        };
        run_parallel([&](unsigned t) {   // pass 2: the lists (ascending, like the bitmap scan gives them)
            for (uint64_t e = U * t / nthreads; e < U * (t + 1) / nthreads; ++e) {
                const uint32_t k = etab[e].k;
                if (!k) continue;
                uint32_t *L = &nb[etab[e].nb_off];
                common(etab[e].big, etab[e].small, L);
                if (!s->sparse) continue;
                // the local pairs (i < j, lexicographic; local indices: the list, then big, then small): pair id and i | j << 8 | swap << 16
                // (swap: L[i] is the smaller vertex of that pair, i.e. its bit 2 id is L[j] -> L[i])
                uint32_t *out = L + k + ((etab[e].nb_off + k) & 1u);
                const uint32_t sl = k + 2;
                auto vert = [&](uint32_t i) { return i < k ? L[i] : (i == k ? etab[e].big : etab[e].small); };
                for (uint32_t i = 0; i < sl; ++i)
                    for (uint32_t j = i + 1; j < sl; ++j) {
                        const uint32_t a = vert(i), b = vert(j);
                        *out++ = (i == k && j == k + 1) ? (uint32_t)e : pair_id(a, b);
                        *out++ = i | (j << 8) | ((a < b ? 1u : 0u) << 16);
                    }
            }
        });
    }
    if (kmax + 2 > FCM_MAX_LOCAL)
        return fail(FCM_ERR_UNSUPPORTED, "an edge has %u common neighbours; this build supports at most %d", kmax, FCM_MAX_LOCAL - 2);
    // at least two mask words of LDS: the one-word path falls back to the wide one when the
    // per-class copies of multi-class vertices do not fit in 64 nodes.  Local sets beyond 256 vertices take the
    // evaluator with its masks in a per-chain workspace (fcm_xwide.hpp), allocated only then.
    s->maxnw_variant = kmax + 2 <= 128 ? 2 : 4;
    const bool need_xw = kmax + 2 > 256;
    const uint64_t D = dbl0.size();

    // --- initial counts and the reachable dimension range ------------------
    std::vector<uint32_t> edges;
    edge_list(*g, edges);
    uint64_t fc[FCM_MAX_COUNTS], ncl[FCM_MAX_COUNTS];
    int fc_len = 0, ncl_len = 0;
    if ((rc = device_count(g->rows.data(), g->n, g->stride32, edges, cfg->device, fc, &fc_len))) return rc;   // lib.rs:51
    if ((rc = clique_counts(*g, cfg->device, ncl, &ncl_len, nullptr))) return rc;
    int want = std::max(std::max(fc_len, ncl_len), std::max((int)bounds->min_len, (int)bounds->max_len));
    want = std::max(want, 2);
    int nc = want;
    bool lossless = true;
    if (cfg->dim_cap > 0 && cfg->dim_cap + 1 < want) { nc = cfg->dim_cap + 1; lossless = false; }
    if (nc > FCM_MAX_COUNTS) return fail(FCM_ERR_UNSUPPORTED, "needs %d count entries, this build tracks at most %d; pass a dim_cap", nc, FCM_MAX_COUNTS);
    nc = std::max(nc, 2);
    s->maxt_variant = nc - 2;   // tracked depth; fcm_launch_step picks the kernel variant

    // --- device buffers -----------------------------------------------------
    // Sparse state only under the multi-wave kernel: simple moves (checked above), 4..8 count entries, not switched off by FCM_MW.
    // (Withdrawn here, the lists still carry the local pair ids behind them: nobody reads those then.)
    {
        bool mw_off = false;
        if (const char *e = getenv("FCM_MW")) { const int v = atoi(e); mw_off = !(v == 2 || v == 4 || v == 8 || v == 16); }
        if (!(nc - 2 >= 2 && nc - 2 <= 6) || mw_off) s->sparse = false;
    }
    if (s->sparse) {   // the chain's record: two bits per adjacent pair (bit 2e: big -> small, 2e + 1: small -> big), padded to whole 128-B lines
        s->bits_stride = (uint32_t)((2 * U + 31) / 32);
        s->bits_stride = (s->bits_stride + 31u) / 32u * 32u;
    }
    const uint64_t rows_per_chain = s->sparse ? (uint64_t)s->bits_stride : (uint64_t)g->n * g->stride32;
    if (rows_per_chain * 4ull >= (1ull << 32))   // (the kernels address a chain's record through a buffer descriptor with 32-bit offsets)
        return fail(FCM_ERR_UNSUPPORTED, "a chain's bitmap would take %llu bytes; this build addresses at most 4 GiB per chain", (unsigned long long)(rows_per_chain * 4ull));
    const uint32_t dbl_stride = (uint32_t)((D + 31) / 32 * 32);
    const uint32_t C = cfg->n_chains;
    if ((rc = s->d_etab.alloc(std::max<size_t>(1, etab.size()) * sizeof(FcmEdgeEntry)))) return rc;
    if ((rc = s->d_nb.alloc((nb.size() + 64) * sizeof(uint32_t)))) return rc;
    if ((rc = s->d_rows.alloc((size_t)C * rows_per_chain * sizeof(uint32_t)))) return rc;
    if ((rc = s->d_dbl.alloc(std::max<size_t>(1, (size_t)C * dbl_stride) * sizeof(uint32_t)))) return rc;
    if ((rc = s->d_counts.alloc((size_t)C * FCM_MAX_COUNTS * sizeof(uint64_t)))) return rc;
    if ((rc = s->d_stats.alloc((size_t)C * FCM_NSTATS * sizeof(uint64_t)))) return rc;
    if (!etab.empty()) HIP_TRY(hipMemcpy(s->d_etab.p, etab.data(), etab.size() * sizeof(FcmEdgeEntry), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(s->d_nb.p, 0, (nb.size() + 64) * sizeof(uint32_t)));
    if (!nb.empty()) HIP_TRY(hipMemcpy(s->d_nb.p, nb.data(), nb.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    {
        DevBuf d_base;
        if ((rc = d_base.alloc(rows_per_chain * sizeof(uint32_t)))) return rc;
        if (s->sparse) {
            std::vector<uint32_t> bits((size_t)rows_per_chain, 0u);
            for (uint64_t e = 0; e < U; ++e) {
                if (g->has(etab[e].big, etab[e].small)) bits[(2 * e) >> 5] |= 1u << ((2 * e) & 31);
                if (g->has(etab[e].small, etab[e].big)) bits[(2 * e + 1) >> 5] |= 1u << ((2 * e + 1) & 31);
            }
            HIP_TRY(hipMemcpy(d_base.p, bits.data(), rows_per_chain * sizeof(uint32_t), hipMemcpyHostToDevice));
        } else {
            HIP_TRY(hipMemcpy(d_base.p, g->rows.data(), rows_per_chain * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        int lrc = fcm_launch_broadcast_rows(s->d_rows.as<uint32_t>(), d_base.as<uint32_t>(), rows_per_chain, C, nullptr);
        if (lrc) return fail(FCM_ERR_HIP, "broadcast launch failed: %s", hipGetErrorString((hipError_t)lrc));
        HIP_TRY(hipDeviceSynchronize());
    }
    {
        std::vector<uint32_t> h((size_t)C * dbl_stride, 0u);
        for (uint32_t c = 0; c < C; ++c) std::copy(dbl0.begin(), dbl0.end(), h.begin() + (size_t)c * dbl_stride);
        if (!h.empty()) HIP_TRY(hipMemcpy(s->d_dbl.p, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        std::vector<uint64_t> hc((size_t)C * FCM_MAX_COUNTS, 0), hs((size_t)C * FCM_NSTATS, 0);
        for (uint32_t c = 0; c < C; ++c) {
            for (int d = 0; d < nc && d < FCM_MAX_COUNTS; ++d) hc[(size_t)c * FCM_MAX_COUNTS + d] = fc[d];
            hs[(size_t)c * FCM_NSTATS + FCM_STAT_COUNT_LEN] = (uint64_t)std::min(fc_len, nc);
        }
        HIP_TRY(hipMemcpy(s->d_counts.p, hc.data(), hc.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_stats.p, hs.data(), hs.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    // --- clique moves: maximal cliques, order thresholds, pair lookup, slot index ---
    CliqueTable ct;
    uint64_t cumo[FCM_MAX_COUNTS] = {0}, clp_base[FCM_MAX_COUNTS] = {0}, clq_pairs_bytes = 0;
    s->clique_moves = cfg->move_weights[2] > 0.0 || cfg->move_weights[3] > 0.0;
    if (s->clique_moves) {
        if (!lossless) return fail(FCM_ERR_UNSUPPORTED, "clique moves need the lossless mode (dim_cap = 0)");
        if ((rc = maximal_cliques(*g, und, ct))) return rc;
        if (ct.orders > 15) return fail(FCM_ERR_UNSUPPORTED, "maximal cliques of %d vertices; the clique moves support up to 15", ct.orders);
        // clique_order_distribution: weights (#cliques of the order)^0.2 (src/bin/sample.rs:87-88)
        double tot = 0, acc = 0;
        int last = 0;
        for (int o = 0; o < ct.orders; ++o) {
            const double wgt = std::pow((double)ct.count[o], 0.2);
            tot += wgt;
            if (wgt > 0.0) last = o;
        }
        for (int o = 0; o < ct.orders; ++o) {
            acc += std::pow((double)ct.count[o], 0.2);
            cumo[o] = (uint64_t)std::floor(4294967296.0 * (acc / tot));
        }
        for (int o = last; o < FCM_MAX_COUNTS; ++o) cumo[o] = 4294967296ull;
        std::vector<uint32_t> efirst((size_t)g->n + 2, 0u);
        for (uint64_t e = 0; e < U; ++e) efirst[etab[e].big + 1]++;
        for (uint32_t v = 0; v < g->n; ++v) efirst[v + 1] += efirst[v];
        if ((rc = s->d_clq.alloc((ct.flat.size() + 64) * sizeof(uint32_t)))) return rc;
        if ((rc = s->d_efirst.alloc(efirst.size() * sizeof(uint32_t)))) return rc;
        if ((rc = s->d_slot_of.alloc(std::max<size_t>(1, (size_t)C * U) * sizeof(uint32_t)))) return rc;
        HIP_TRY(hipMemset(s->d_clq.p, 0, (ct.flat.size() + 64) * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(s->d_clq.p, ct.flat.data(), ct.flat.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_efirst.p, efirst.data(), efirst.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        // pair ids of every maximal clique: for clique positions i < j (vertices ascending, so the
        // pair is (v_j, v_i)) in the order (0,1),(0,2),..,(1,2),..; one coalesced read on the device
        // instead of a table search per changed edge
        {
            std::vector<uint32_t> pairs;
            uint64_t tot = 0;
            for (int o = 1; o <= ct.orders; ++o) { clp_base[o - 1] = tot; tot += ct.count[o - 1] * (uint64_t)(o * (o - 1) / 2); }
            pairs.resize((size_t)tot + 64, 0u);
            for (int o = 2; o <= ct.orders; ++o) {
                const uint32_t *b = ct.flat.data() + ct.base[o - 1];
                uint32_t *dst = pairs.data() + clp_base[o - 1];
                for (uint64_t c = 0; c < ct.count[o - 1]; ++c, b += o)
                    for (int i = 0; i < o; ++i)
                        for (int j = i + 1; j < o; ++j) {
                            const uint32_t big = b[j], small = b[i];
                            // the run of `big` in etab is sorted by `small`
                            uint32_t lo = efirst[big], hi = efirst[big + 1];
                            while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (etab[mid].small < small) lo = mid + 1; else hi = mid; }
                            *dst++ = lo;
                        }
            }
            if ((rc = s->d_clq_pairs.alloc(pairs.size() * sizeof(uint32_t)))) return rc;
            HIP_TRY(hipMemcpy(s->d_clq_pairs.p, pairs.data(), pairs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            clq_pairs_bytes = pairs.size() * sizeof(uint32_t);
        }
        std::vector<uint32_t> so((size_t)U, 0xFFFFFFFFu);
        for (size_t j = 0; j < dbl0.size(); ++j) so[dbl0[j]] = (uint32_t)j;
        for (uint32_t c = 0; c < C && U; ++c)
            HIP_TRY(hipMemcpy(s->d_slot_of.as<uint32_t>() + (size_t)c * U, so.data(), (size_t)U * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (need_xw && (rc = s->d_xw.alloc((size_t)C * FCM_XW_WORDS_HOST * sizeof(uint64_t)))) return rc;
    if ((rc = s->d_dbg.alloc((size_t)C * 8 * sizeof(uint64_t)))) return rc;
    HIP_TRY(hipMemset(s->d_dbg.p, 0, (size_t)C * 8 * sizeof(uint64_t)));
    HIP_TRY(hipStreamCreate(&s->own_stream));
    s->stream = s->own_stream;
    HIP_TRY(hipEventCreate(&s->ev0));
    HIP_TRY(hipEventCreate(&s->ev1));

    // --- kernel parameters ---------------------------------------------------
    FcmStepParams &p = s->params;
    memset(&p, 0, sizeof p);
    p.etab = s->d_etab.as<FcmEdgeEntry>();
    p.nb = s->d_nb.as<uint32_t>();
    p.rows = s->d_rows.as<uint32_t>();
    p.dbl = s->d_dbl.as<uint32_t>();
    p.counts = s->d_counts.as<uint64_t>();
    p.stats = s->d_stats.as<uint64_t>();
    p.dbgbuf = s->d_dbg.as<uint64_t>();
    for (int d = 0; d < FCM_MAX_COUNTS; ++d) {
        // zero padding of the shorter side, src/util.rs:53-57
        p.bmin[d] = d < bounds->min_len ? bounds->flag_count_min[d] : 0;
        p.bmax[d] = d < bounds->max_len ? bounds->flag_count_max[d] : 0;
    }
    uint64_t cum[4];
    move_thresholds(cfg->move_weights, cum);
    p.cum0 = cum[0];
    p.cum1 = cum[1];
    p.cum2 = cum[2];
    if (s->clique_moves) {
        p.clq = s->d_clq.as<uint32_t>();
        p.clq_pairs = s->d_clq_pairs.as<uint32_t>();
        for (int o = 0; o < FCM_MAX_COUNTS; ++o) p.clp_base[o] = clp_base[o];
        p.efirst = s->d_efirst.as<uint32_t>();
        p.slot_of = s->d_slot_of.as<uint32_t>();
        for (int o = 0; o < FCM_MAX_COUNTS; ++o) { p.cl_base[o] = ct.base[o]; p.cl_count[o] = ct.count[o]; p.cumo[o] = cumo[o]; }
        p.cl_orders = ct.orders;
        p.chg_cap = (uint32_t)std::max(32, 2 * ct.orders * ct.orders);   // u64 words: 4 u32 per vertex pair, at most o(o-1) changed pairs (two cliques of a swap)
    }
    p.seed = cfg->seed;
    p.rows_per_chain = rows_per_chain;
    p.n = g->n;
    p.stride32 = g->stride32;
    p.U = (uint32_t)U;
    p.D = (uint32_t)D;
    p.dbl_stride = dbl_stride;
    p.first_chain = cfg->first_chain_id;
    p.nchains = C;
    p.ncounts = nc;
    p.maxnw = s->maxnw_variant;
    p.xw_ws = need_xw ? s->d_xw.as<uint64_t>() : nullptr;
    p.sparse = s->sparse ? 1u : 0u;
    p.guard_limit = 0x7FFFFFFFull;
    if (const char *e = getenv("FCM_TEST_GUARD_LIMIT")) p.guard_limit = strtoull(e, nullptr, 10);   // test hook (tests/test_gpu_parity.py)
    p.commit_words = (uint32_t)rows_per_chain;
    if (const char *e = getenv("FCM_TEST_COMMIT_LIMIT")) p.commit_words = (uint32_t)strtoul(e, nullptr, 10);   // test hook: a commit's word index at or beyond it is "out of range"

    if (s->cfg.sample_distance == 0) s->cfg.sample_distance = fcm_default_sample_distance(fc[1]);  // sample.rs:102

    fcm_sampler_info &I = s->info;
    I.n = g->n;
    I.row_words = g->stride32 / 2;
    I.n_undirected = U;
    I.n_double = D;
    I.k_max = kmax;
    I.k_mean = U ? (double)ksum / (double)U : 0.0;
    I.sparse_state = s->sparse ? 1u : 0u;
    I.bytes_per_chain = rows_per_chain * 4 + (uint64_t)dbl_stride * 4 + FCM_MAX_COUNTS * 8 + FCM_NSTATS * 8 + (s->clique_moves ? U * 4 : 0) + (need_xw ? FCM_XW_WORDS_HOST * 8 : 0);
    I.bytes_static = etab.size() * sizeof(FcmEdgeEntry) + nb.size() * 4 + ct.flat.size() * 4 + clq_pairs_bytes;
    I.ncounts = nc;
    I.lossless = lossless ? 1 : 0;
    I.n_chains = C;
    // Simple moves run with several waves per chain (fcm_step_mw.hpp): W consecutive proposals in flight, decided in order.
    // W is chosen so that chains x W fills the chip's 8192 wave slots without passing them: 8 up to 1024 chains, 4 up to 2048, 2 above; 16 on
    // graphs of more than 1024 vertices whose proposals are memory-heavy, or up to 256 chains (on smaller graphs more than 8 proposals in flight buy nothing: the in-order decisions are the limit by then).
    // FCM_MW=<1|2|4|8|16> overrides (1 = the one-wave kernel).
    {
        uint32_t W = C > 2048 ? 2u : (C > 1024 ? 4u : 8u);   // the largest of 8, 4, 2 with chains x W <= 8192 (all resident)
        if (p.stride32 > 32) {
            // rows longer than a cache line (n > 1024).  If a build touches many lines (config 3: 42 rows x 4 lines) a proposal is
            // mostly memory round trips and 16 of them in flight per chain pay at any chain count, even when the chains then
            // run in several rounds (4096 chains of config 3: +21 % over W = 2); otherwise only while the chip is not full.
            const double rows_touched = I.k_mean + 2.0, lines = rows_touched * std::min(rows_touched, (double)p.stride32 / 32.0);
            if (lines >= 64.0 || C <= 256) W = 16u;
        }
        if (const char *e = getenv("FCM_MW")) {
            const int v = atoi(e);
            W = (v == 2 || v == 4 || v == 8 || v == 16) ? (uint32_t)v : 1u;
        }
        if (s->clique_moves || nc - 2 < 2 || nc - 2 > 6) W = 1u;
        I.waves_per_chain = W;
        p.mw_waves = W >= 2 ? W : 0u;
        // Move mixes with clique moves: the kernel that evaluates a move's pairs on the pre-move bitmap (fcm_step_cq.hpp), for
        // up to 8 count entries like the multi-wave kernel; otherwise, or with FCM_CQ=0, the one-wave kernel's clique path.
        // Measured on configs[2]'s graph (default mix, profiles/r03_default_mix_*): 2048 chains 6.9e7 vs 5.5e7 proposals/s, 1024
        // chains 4.8e7 vs 3.1e7, 256 chains 1.6e7 vs 0.8e7; with 4096 chains one wave per chain already fills the chip and the
        // one-wave kernel, which needs no patches, is 4 % ahead (9.1e7 vs 8.7e7): it stays the choice there.  FCM_CQ=1 / 0 forces
        // the one or the other.
        const char *cq = getenv("FCM_CQ");
        s->use_cq = s->clique_moves && nc - 2 >= 2 && nc - 2 <= 6 && ct.orders <= 8 && (cq ? atoi(cq) != 0 : C <= 2048);   // (<= 8 count entries: cliques of <= 8 vertices, the kernel's tables)
        if (s->use_cq) {
            // W waves per chain share a move's pairs: as many as keep chains x W within the 4096 wave slots of 4 waves per SIMD
            // (128 VGPRs), at most 8 (a move changes about 6 pairs).  FCM_CQW=<1|2|4|8> overrides.
            uint32_t Wc = C > 2048 ? 1u : (C > 1024 ? 2u : (C > 512 ? 4u : 8u));
            if (const char *e = getenv("FCM_CQW")) { const int v = atoi(e); Wc = (v == 2 || v == 4 || v == 8) ? (uint32_t)v : 1u; }
            I.waves_per_chain = Wc;
            p.mw_waves = Wc >= 2 ? Wc : 0u;
        }
        I.cooperative_clique_kernel = s->use_cq ? 1u : 0u;
    }

    guard.s = nullptr;
    *out = s;
    return FCM_OK;
} FCM_CATCH

extern "C" void fcm_sampler_destroy(fcm_sampler *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    delete s;
}

extern "C" int fcm_sampler_set_stream(fcm_sampler *s, void *hip_stream)
try {
    if (!s) return fail(FCM_ERR_INVALID, "sampler is NULL");
    s->stream = hip_stream ? (hipStream_t)hip_stream : s->own_stream;
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_sampler_step(fcm_sampler *s, uint64_t n_proposals)
try {
    if (!s) return fail(FCM_ERR_INVALID, "sampler is NULL");
    int rc = use_device(s->device);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
    uint64_t left = n_proposals;
    while (left > 0) {
        // (samplers with clique moves: launches of at most 4096 proposals -- the clique kernels' per-launch tallies are u32 words in LDS
        //  and a move can add up to 56 pairs x 2 directions x 1022 common neighbours to the sum of k)
        static_assert(4096ull * 56ull * 2ull * 1024ull < (1ull << 32), "per-launch tallies of the clique kernels fit 32 bits");
        const uint64_t chunk = std::min<uint64_t>(left, s->clique_moves ? 4096u : FCM_LAUNCH_CHUNK);
        s->params.nprop = chunk;
        int lrc = fcm_launch_step(&s->params, s->maxt_variant, s->clique_moves ? (s->use_cq ? 3 : 1) : (s->info.waves_per_chain >= 2 ? 2 : 0), s->stream);
        if (lrc) return fail(FCM_ERR_HIP, "step kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
        left -= chunk;
    }
    HIP_TRY(hipEventRecord(s->ev1, s->stream));
    s->timed = true;
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_sampler_sync(fcm_sampler *s)
try {
    if (!s) return fail(FCM_ERR_INVALID, "sampler is NULL");
    int rc = use_device(s->device);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_sampler_next(fcm_sampler *s)
try {
    if (!s) return fail(FCM_ERR_INVALID, "sampler is NULL");
    int rc = fcm_sampler_step(s, s->cfg.sample_distance);
    if (rc) return rc;
    return fcm_sampler_sync(s);
} FCM_CATCH

extern "C" int fcm_sampler_last_step_ms(fcm_sampler *s, float *ms)
try {
    if (!s || !ms) return fail(FCM_ERR_INVALID, "NULL argument");
    if (!s->timed) return fail(FCM_ERR_INVALID, "no step has been launched");
    int rc = use_device(s->device);
    if (rc) return rc;
    HIP_TRY(hipEventSynchronize(s->ev1));
    HIP_TRY(hipEventElapsedTime(ms, s->ev0, s->ev1));
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_sampler_ncounts(const fcm_sampler *s) { return s ? s->params.ncounts : 0; }
extern "C" uint64_t fcm_sampler_sample_distance(const fcm_sampler *s) { return s ? s->cfg.sample_distance : 0; }

static int fetch_stats(fcm_sampler *s, std::vector<uint64_t> &hs)
{
    int rc = fcm_sampler_sync(s);
    if (rc) return rc;
    hs.resize((size_t)s->params.nchains * FCM_NSTATS);
    HIP_TRY(hipMemcpy(hs.data(), s->d_stats.p, hs.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < s->params.nchains; ++c) {
        const uint64_t st = hs[(size_t)c * FCM_NSTATS + FCM_STAT_STATUS];
        if (st) return fail(FCM_ERR_INTERNAL, "chain %u: device-side consistency check failed (status 0x%llx)", c, (unsigned long long)st);
    }
    return FCM_OK;
}

extern "C" int fcm_sampler_get_counts(fcm_sampler *s, uint64_t *out, int32_t *count_len)
try {
    if (!s || !out) return fail(FCM_ERR_INVALID, "NULL argument");
    std::vector<uint64_t> hs;
    int rc = fetch_stats(s, hs);
    if (rc) return rc;
    const uint32_t C = s->params.nchains;
    const int nc = s->params.ncounts;
    std::vector<uint64_t> hc((size_t)C * FCM_MAX_COUNTS);
    HIP_TRY(hipMemcpy(hc.data(), s->d_counts.p, hc.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < C; ++c) {
        for (int d = 0; d < nc; ++d) out[(size_t)c * nc + d] = hc[(size_t)c * FCM_MAX_COUNTS + d];
        if (count_len) count_len[c] = (int32_t)hs[(size_t)c * FCM_NSTATS + FCM_STAT_COUNT_LEN];
    }
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_sampler_get_stats(fcm_sampler *s, uint64_t *out)
try {
    if (!s || !out) return fail(FCM_ERR_INVALID, "NULL argument");
    std::vector<uint64_t> hs;
    int rc = fetch_stats(s, hs);
    if (rc) return rc;
    memcpy(out, hs.data(), hs.size() * sizeof(uint64_t));
    return FCM_OK;
} FCM_CATCH

// sparse state: the chain's record (bits_stride words)
static int fetch_bits(fcm_sampler *s, uint32_t chain, std::vector<uint32_t> &bits)
{
    bits.resize((size_t)s->bits_stride);
    HIP_TRY(hipMemcpy(bits.data(), s->d_rows.as<uint32_t>() + (size_t)chain * s->bits_stride, bits.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return FCM_OK;
}

static int fetch_rows(fcm_sampler *s, uint32_t chain, std::vector<uint32_t> &rows)
{
    if (chain >= s->params.nchains) return fail(FCM_ERR_INVALID, "chain %u out of range", chain);
    int rc = fcm_sampler_sync(s);
    if (rc) return rc;
    if (s->sparse) {   // expand the chain's two bits per pair into row bitmaps
        std::vector<uint32_t> bits;
        if ((rc = fetch_bits(s, chain, bits))) return rc;
        rows.assign((size_t)s->n * s->stride32, 0u);
        const uint64_t U = s->ue.size() / 2;
        for (uint64_t e = 0; e < U; ++e) {
            const uint32_t a = s->ue[2 * e], b = s->ue[2 * e + 1];
            if ((bits[(2 * e) >> 5] >> ((2 * e) & 31)) & 1u) rows[(size_t)a * s->stride32 + (b >> 5)] |= 1u << (b & 31);
            if ((bits[(2 * e + 1) >> 5] >> ((2 * e + 1) & 31)) & 1u) rows[(size_t)b * s->stride32 + (a >> 5)] |= 1u << (a & 31);
        }
        return FCM_OK;
    }
    rows.resize((size_t)s->params.rows_per_chain);
    HIP_TRY(hipMemcpy(rows.data(), s->d_rows.as<uint32_t>() + (size_t)chain * s->params.rows_per_chain,
                      rows.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return FCM_OK;
}

extern "C" int fcm_sampler_get_edges(fcm_sampler *s, uint32_t chain, fcm_node *out, uint64_t cap, uint64_t *m)
try {
    if (!s) return fail(FCM_ERR_INVALID, "sampler is NULL");
    std::vector<uint32_t> rows;
    int rc = fetch_rows(s, chain, rows);
    if (rc) return rc;
    const uint32_t nw = (s->n + 31) / 32;
    uint64_t i = 0;
    for (uint32_t a = 0; a < s->n; ++a)
        for (uint32_t w = 0; w < nw; ++w) {
            uint32_t x = rows[(size_t)a * s->stride32 + w];
            while (x) {
                const uint32_t b = w * 32 + (uint32_t)__builtin_ctz(x);
                x &= x - 1;
                if (out && i < cap) { out[2 * i] = a; out[2 * i + 1] = b; }
                ++i;
            }
        }
    if (m) *m = i;
    return FCM_OK;
} FCM_CATCH

// BitOutput::save record, src/io.rs:152-159 (slot order) and :180-194 (packing)
extern "C" int fcm_sampler_get_edgebits(fcm_sampler *s, uint32_t chain, uint8_t *out, uint64_t cap, uint64_t *nbytes)
try {
    if (!s) return fail(FCM_ERR_INVALID, "sampler is NULL");
    const uint64_t U = s->ue.size() / 2;
    const uint64_t need = (2 * U + 7) / 8;
    if (nbytes) *nbytes = need;
    if (!out) return FCM_OK;
    if (cap < need) return fail(FCM_ERR_INVALID, "edgebits buffer too small: need %llu", (unsigned long long)need);
    if (s->sparse) {   // the record IS the chain's state
        if (chain >= s->params.nchains) return fail(FCM_ERR_INVALID, "chain %u out of range", chain);
        int rcs = fcm_sampler_sync(s);
        if (rcs) return rcs;
        std::vector<uint32_t> bits;
        if ((rcs = fetch_bits(s, chain, bits))) return rcs;
        memcpy(out, bits.data(), (size_t)need);
        return FCM_OK;
    }
    std::vector<uint32_t> rows;
    int rc = fetch_rows(s, chain, rows);
    if (rc) return rc;
    memset(out, 0, (size_t)need);
    auto has = [&](uint32_t a, uint32_t b) { return (rows[(size_t)a * s->stride32 + (b >> 5)] >> (b & 31)) & 1u; };
    for (uint64_t e = 0; e < U; ++e) {
        const uint32_t a = s->ue[2 * e], b = s->ue[2 * e + 1];
        if (has(a, b)) out[(2 * e) >> 3] |= (uint8_t)(1u << ((2 * e) & 7));          // slot [big,small]: a<b false sorts first
        if (has(b, a)) out[(2 * e + 1) >> 3] |= (uint8_t)(1u << ((2 * e + 1) & 7));  // slot [small,big]
    }
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_sampler_get_double_slots(fcm_sampler *s, uint32_t chain, uint32_t *out, uint64_t cap, uint64_t *n)
try {
    if (!s) return fail(FCM_ERR_INVALID, "sampler is NULL");
    if (chain >= s->params.nchains) return fail(FCM_ERR_INVALID, "chain %u out of range", chain);
    int rc = fcm_sampler_sync(s);
    if (rc) return rc;
    const uint64_t D = s->params.D;
    if (n) *n = D;
    if (out && D) {
        const uint64_t k = std::min<uint64_t>(D, cap);
        HIP_TRY(hipMemcpy(out, s->d_dbl.as<uint32_t>() + (size_t)chain * s->params.dbl_stride, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_sampler_get_info(const fcm_sampler *s, fcm_sampler_info *out)
try {
    if (!s || !out) return fail(FCM_ERR_INVALID, "NULL argument");
    *out = s->info;
    return FCM_OK;
} FCM_CATCH


extern "C" int fcm_sampler_get_bounds(const fcm_sampler *s, fcm_bounds *out)
try {
    if (!s || !out) return fail(FCM_ERR_INVALID, "NULL argument");
    *out = s->bounds;
    return FCM_OK;
} FCM_CATCH

// ---------------------------------------------------------------------------
// State API on one chain: edgeset_neighborhood (src/lib.rs:99-111), apply_transition (:61-79), revert_transition
// (:81-95), Transition::single_edge_flip (:292-299).  What the reference's search tools call between proposals
// (src/bin/seo_search_counterexample.rs:51-89, seo_bt_flip_only_once.rs:65-69, all_cxs.rs:54-86).  (pre, post) are the
// reference's own vectors -- flagser_count of the induced subgraph on the edge set's neighbourhood before and after --
// counted by the HIP counting kernel; callers look at their lengths as well as at post[2] - pre[2].
// Host-orchestrated, a few launches and small copies per call: not the stepping path.
// ---------------------------------------------------------------------------
static int ensure_host_tables(fcm_sampler *s)
{
    if (s->h_tables) return FCM_OK;
    const size_t U = s->ue.size() / 2;
    s->h_etab.resize(U);
    if (U) HIP_TRY(hipMemcpy(s->h_etab.data(), s->d_etab.p, U * sizeof(FcmEdgeEntry), hipMemcpyDeviceToHost));
    size_t total = 0;
    if (U) total = (size_t)s->h_etab[U - 1].nb_off + s->h_etab[U - 1].k;
    s->h_nb.resize(total);
    if (total) HIP_TRY(hipMemcpy(s->h_nb.data(), s->d_nb.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost));
    s->h_tables = true;
    return FCM_OK;
}

// etab index of the adjacent pair {a, b}, or -1 (the reference indexes a HashMap with it: a missing key panics)
static int64_t pair_index(const fcm_sampler *s, uint32_t a, uint32_t b)
{
    const uint32_t big = std::max(a, b), small = std::min(a, b);
    size_t lo = 0, hi = s->ue.size() / 2;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        const uint32_t mb = s->ue[2 * mid], ms = s->ue[2 * mid + 1];
        if (mb < big || (mb == big && ms < small)) lo = mid + 1; else hi = mid;
    }
    if (lo < s->ue.size() / 2 && s->ue[2 * lo] == big && s->ue[2 * lo + 1] == small) return (int64_t)lo;
    return -1;
}

static int check_transition_args(const fcm_sampler *s, uint32_t chain, const fcm_node *edges, const int32_t *add, uint32_t n)
{
    if (!s) return fail(FCM_ERR_INVALID, "sampler is NULL");
    if (chain >= s->params.nchains) return fail(FCM_ERR_INVALID, "chain %u out of range", chain);
    if (n && (!edges || !add)) return fail(FCM_ERR_INVALID, "NULL argument");
    if (n > 4096) return fail(FCM_ERR_UNSUPPORTED, "%u change edges; at most 4096 per transition", n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t a = edges[2 * i], b = edges[2 * i + 1];
        if (a >= s->n || b >= s->n || a == b) return fail(FCM_ERR_INVALID, "change edge %u = (%u,%u) out of range or a loop", i, a, b);
        if (pair_index(s, a, b) < 0)
            return fail(FCM_ERR_PANIC, "change edge %u = (%u,%u): the pair is not adjacent in pr(G); the reference indexes edge_neighborhood with it (src/lib.rs:104)", i, a, b);
    }
    return FCM_OK;
}

static int edgeset_neighborhood(fcm_sampler *s, const fcm_node *edges, uint32_t n, std::vector<uint32_t> &out)
{
    int rc = ensure_host_tables(s);
    if (rc) return rc;
    out.clear();
    for (uint32_t i = 0; i < n; ++i) {
        const int64_t e = pair_index(s, edges[2 * i], edges[2 * i + 1]);
        if (e < 0) return fail(FCM_ERR_PANIC, "edge (%u,%u) is not in edge_neighborhood (src/lib.rs:104)", edges[2 * i], edges[2 * i + 1]);
        const FcmEdgeEntry &t = s->h_etab[(size_t)e];
        out.insert(out.end(), s->h_nb.begin() + t.nb_off, s->h_nb.begin() + t.nb_off + t.k);
        out.push_back(edges[2 * i]);
        out.push_back(edges[2 * i + 1]);
    }
    std::sort(out.begin(), out.end());                       // sort_unstable, dedup (:108-109)
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return FCM_OK;
}

extern "C" int fcm_sampler_edgeset_neighborhood(fcm_sampler *s, const fcm_node *edges, uint32_t n, fcm_node *out, uint64_t cap, uint64_t *k)
try {
    if (!s || (n && !edges)) return fail(FCM_ERR_INVALID, "NULL argument");
    int rc = use_device(s->device);
    if (rc) return rc;
    for (uint32_t i = 0; i < n; ++i)
        if (edges[2 * i] >= s->n || edges[2 * i + 1] >= s->n) return fail(FCM_ERR_INVALID, "edge %u out of range", i);
    std::vector<uint32_t> nb;
    if ((rc = edgeset_neighborhood(s, edges, n, nb))) return rc;
    if (k) *k = nb.size();
    if (out) memcpy(out, nb.data(), sizeof(uint32_t) * (size_t)std::min<uint64_t>(cap, nb.size()));
    return FCM_OK;
} FCM_CATCH

static inline bool sub_has(const std::vector<uint32_t> &sub, uint32_t nlw, uint32_t i, uint32_t j) { return (sub[(size_t)i * nlw + (j >> 5)] >> (j & 31)) & 1u; }
static inline void sub_set(std::vector<uint32_t> &sub, uint32_t nlw, uint32_t i, uint32_t j, bool present)
{
    uint32_t &w = sub[(size_t)i * nlw + (j >> 5)];
    const uint32_t b = 1u << (j & 31);
    w = present ? (w | b) : (w & ~b);
}

// induced adjacency of `list` in the chain's current graph: sub[i * nlw + w], bit j = list[i] -> list[j]
static int gather_sub(fcm_sampler *s, uint32_t chain, const std::vector<uint32_t> &list, std::vector<uint32_t> &sub, uint32_t &nlw)
{
    const uint32_t nl = (uint32_t)list.size();
    nlw = (nl + 63) / 64 * 2;
    sub.assign((size_t)nl * nlw, 0u);
    if (nl == 0) return FCM_OK;
    int rc;
    if (s->sparse) {   // read off the chain's record on the host
        std::vector<uint32_t> bits;
        if ((rc = fetch_bits(s, chain, bits))) return rc;
        for (uint32_t i = 0; i < nl; ++i)
            for (uint32_t j = 0; j < i; ++j) {
                const int64_t e = pair_index(s, list[i], list[j]);
                if (e < 0) continue;
                const bool ibig = list[i] > list[j];
                const bool bs = (bits[(2 * e) >> 5] >> ((2 * e) & 31)) & 1u, sb = (bits[(2 * e + 1) >> 5] >> ((2 * e + 1) & 31)) & 1u;
                sub_set(sub, nlw, i, j, ibig ? bs : sb);
                sub_set(sub, nlw, j, i, ibig ? sb : bs);
            }
        return FCM_OK;
    }
    if (s->tr_list_cap < nl) { DevBuf nb_; if ((rc = nb_.alloc((size_t)nl * 4))) return rc; std::swap(s->d_tr_list.p, nb_.p); s->tr_list_cap = nl; }
    if (s->tr_out_cap < sub.size()) { DevBuf nb_; if ((rc = nb_.alloc(sub.size() * 4))) return rc; std::swap(s->d_tr_out.p, nb_.p); s->tr_out_cap = sub.size(); }
    HIP_TRY(hipMemcpyAsync(s->d_tr_list.p, list.data(), (size_t)nl * 4, hipMemcpyHostToDevice, s->stream));
    const uint32_t *rows = s->d_rows.as<uint32_t>() + (size_t)chain * s->params.rows_per_chain;
    int lrc = fcm_launch_gather_sub(rows, s->stride32, s->d_tr_list.as<uint32_t>(), nl, nlw, s->d_tr_out.as<uint32_t>(), s->stream);
    if (lrc) return fail(FCM_ERR_HIP, "gather launch failed: %s", hipGetErrorString((hipError_t)lrc));
    HIP_TRY(hipMemcpyAsync(sub.data(), s->d_tr_out.p, sub.size() * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return FCM_OK;
}

// flagser_count of the graph on `nl` vertices whose adjacency is `sub` (as gather_sub makes it)
static int count_sub(const fcm_sampler *s, const std::vector<uint32_t> &sub, uint32_t nl, uint32_t nlw, uint64_t counts[FCM_MAX_COUNTS], int *len)
{
    const uint32_t st = stride_for(nl);
    std::vector<uint32_t> rows((size_t)nl * st, 0u), edges;
    for (uint32_t i = 0; i < nl; ++i)
        for (uint32_t w = 0; w < nlw; ++w) {
            uint32_t x = sub[(size_t)i * nlw + w];
            rows[(size_t)i * st + w] = x;
            while (x) { edges.push_back(i); edges.push_back(w * 32 + (uint32_t)__builtin_ctz(x)); x &= x - 1; }
        }
    return device_count(rows.data(), nl, st, edges, s->device, counts, len);
}


// The chain's reciprocal-pair slot list after set_edge calls: the pairs (ascending id) that stopped being reciprocal
// hand their slots to the pairs that became reciprocal, in order (the rule of the clique moves).  The
// kernels draw double-edge moves over a fixed number D of slots, so a transition that changes the number of
// reciprocal pairs is refused before anything is written.
struct SlotPlan { std::vector<uint32_t> lost, gained; };
static int plan_slots(const fcm_sampler *s, const std::vector<uint32_t> &list, const std::vector<uint32_t> &before, const std::vector<uint32_t> &after,
                      uint32_t nlw, const fcm_node *edges, uint32_t n, SlotPlan &plan)
{
    auto idx = [&](uint32_t v) { return (uint32_t)(std::lower_bound(list.begin(), list.end(), v) - list.begin()); };
    std::vector<uint32_t> ids;
    for (uint32_t i = 0; i < n; ++i) ids.push_back((uint32_t)pair_index(s, edges[2 * i], edges[2 * i + 1]));
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    for (uint32_t e : ids) {
        const uint32_t ia = idx(s->ue[2 * (size_t)e]), ib = idx(s->ue[2 * (size_t)e + 1]);
        const bool rb = sub_has(before, nlw, ia, ib) && sub_has(before, nlw, ib, ia), ra = sub_has(after, nlw, ia, ib) && sub_has(after, nlw, ib, ia);
        if (rb && !ra) plan.lost.push_back(e);
        if (ra && !rb) plan.gained.push_back(e);
        if (!sub_has(after, nlw, ia, ib) && !sub_has(after, nlw, ib, ia))
            return fail(FCM_ERR_UNSUPPORTED, "the transition leaves the pair (%u,%u) without an edge: pr(G), and with it the static tables, would change",
                        s->ue[2 * (size_t)e], s->ue[2 * (size_t)e + 1]);
    }
    if (plan.lost.size() != plan.gained.size())
        return fail(FCM_ERR_UNSUPPORTED, "the transition changes the number of reciprocal pairs (%zu lost, %zu gained); the sampler draws over a fixed number of them",
                    plan.lost.size(), plan.gained.size());
    return FCM_OK;
}
static int commit_slots(fcm_sampler *s, uint32_t chain, const SlotPlan &plan)
{
    if (plan.lost.empty()) return FCM_OK;
    const uint64_t D = s->params.D;
    std::vector<uint32_t> dbl((size_t)D);
    uint32_t *d_dbl = s->d_dbl.as<uint32_t>() + (size_t)chain * s->params.dbl_stride;
    HIP_TRY(hipMemcpy(dbl.data(), d_dbl, (size_t)D * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < plan.lost.size(); ++i) {
        const auto it = std::find(dbl.begin(), dbl.end(), plan.lost[i]);
        if (it == dbl.end()) return fail(FCM_ERR_INTERNAL, "chain %u: reciprocal pair %u is not in the slot list", chain, plan.lost[i]);
        const uint32_t slot = (uint32_t)(it - dbl.begin()), ge = plan.gained[i];
        *it = ge;
        HIP_TRY(hipMemcpy(d_dbl + slot, &ge, 4, hipMemcpyHostToDevice));
        if (s->clique_moves) {
            uint32_t *so = s->d_slot_of.as<uint32_t>() + (size_t)chain * s->params.U;
            const uint32_t none = 0xFFFFFFFFu;
            HIP_TRY(hipMemcpy(so + plan.lost[i], &none, 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(so + ge, &slot, 4, hipMemcpyHostToDevice));
        }
    }
    return FCM_OK;
}

static int set_edges_on_device(fcm_sampler *s, uint32_t chain, const fcm_node *edges, const int32_t *add, uint32_t n, bool invert)
{
    if (n == 0) return FCM_OK;
    if (s->sparse) {
        std::vector<uint32_t> bits;
        int rcs = fetch_bits(s, chain, bits);
        if (rcs) return rcs;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t a = edges[2 * i], b = edges[2 * i + 1];
            const uint64_t bit = 2 * (uint64_t)pair_index(s, a, b) + (a > b ? 0 : 1);
            if ((add[i] != 0) != invert) bits[bit >> 5] |= 1u << (bit & 31); else bits[bit >> 5] &= ~(1u << (bit & 31));
        }
        HIP_TRY(hipMemcpy(s->d_rows.as<uint32_t>() + (size_t)chain * s->bits_stride, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        return FCM_OK;
    }
    std::vector<uint32_t> chg((size_t)n * 3);
    for (uint32_t i = 0; i < n; ++i) { chg[3 * i] = edges[2 * i]; chg[3 * i + 1] = edges[2 * i + 1]; chg[3 * i + 2] = ((add[i] != 0) != invert) ? 1u : 0u; }
    int rc;
    if (s->tr_chg_cap < chg.size()) { DevBuf nb_; if ((rc = nb_.alloc(chg.size() * 4))) return rc; std::swap(s->d_tr_chg.p, nb_.p); s->tr_chg_cap = chg.size(); }
    HIP_TRY(hipMemcpyAsync(s->d_tr_chg.p, chg.data(), chg.size() * 4, hipMemcpyHostToDevice, s->stream));
    uint32_t *rows = s->d_rows.as<uint32_t>() + (size_t)chain * s->params.rows_per_chain;
    int lrc = fcm_launch_set_edges(rows, s->stride32, s->d_tr_chg.as<uint32_t>(), n, s->stream);
    if (lrc) return fail(FCM_ERR_HIP, "set_edge launch failed: %s", hipGetErrorString((hipError_t)lrc));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return FCM_OK;
}

// flag_count -= sub; (resize) += add -- src/lib.rs:64-67,72-77 and :85-94.  The reference asserts while it subtracts; here
// the check comes first and a failing one changes nothing.
static int update_counts(fcm_sampler *s, uint32_t chain, const uint64_t *sub, int sub_len, const uint64_t *addv, int add_len)
{
    const int nc = s->params.ncounts;
    uint64_t c[FCM_MAX_COUNTS];
    uint64_t *d_c = s->d_counts.as<uint64_t>() + (size_t)chain * FCM_MAX_COUNTS;
    uint64_t *d_len = s->d_stats.as<uint64_t>() + (size_t)chain * FCM_NSTATS + FCM_STAT_COUNT_LEN;
    uint64_t len = 0;
    HIP_TRY(hipMemcpy(c, d_c, sizeof c, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&len, d_len, 8, hipMemcpyDeviceToHost));
    for (int d = 0; d < sub_len && d < nc && d < (int)len; ++d)      // zip: up to the shorter of the two
        if (c[d] < sub[d]) return fail(FCM_ERR_PANIC, "flag_count[%d] = %llu < %llu: the reference's assert!(*s >= *p) fires (src/lib.rs:65,86)", d,
                                       (unsigned long long)c[d], (unsigned long long)sub[d]);
    for (int d = 0; d < sub_len && d < nc && d < (int)len; ++d) c[d] -= sub[d];
    if ((uint64_t)std::min(add_len, nc) > len) len = (uint64_t)std::min(add_len, nc);   // flag_count.resize(post.len(), 0): never shrinks
    for (int d = 0; d < add_len && d < nc; ++d) c[d] += addv[d];
    HIP_TRY(hipMemcpy(d_c, c, sizeof c, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_len, &len, 8, hipMemcpyHostToDevice));
    return FCM_OK;
}

extern "C" int fcm_sampler_apply_transition(fcm_sampler *s, uint32_t chain, const fcm_node *edges, const int32_t *add, uint32_t n,
                                            uint64_t *pre, int32_t *pre_len, uint64_t *post, int32_t *post_len)
try {
    int rc = check_transition_args(s, chain, edges, add, n);
    if (rc) return rc;
    if (!pre || !pre_len || !post || !post_len) return fail(FCM_ERR_INVALID, "NULL argument");
    if ((rc = fcm_sampler_sync(s))) return rc;
    std::vector<uint32_t> list, before, after;
    if ((rc = edgeset_neighborhood(s, edges, n, list))) return rc;               // :62
    uint32_t nlw = 0;
    if ((rc = gather_sub(s, chain, list, before, nlw))) return rc;               // Graph::subgraph, :63
    uint64_t cpre[FCM_MAX_COUNTS], cpost[FCM_MAX_COUNTS];
    int lpre = 0, lpost = 0;
    if ((rc = count_sub(s, before, (uint32_t)list.size(), nlw, cpre, &lpre))) return rc;
    after = before;
    auto idx = [&](uint32_t v) { return (uint32_t)(std::lower_bound(list.begin(), list.end(), v) - list.begin()); };
    for (uint32_t i = 0; i < n; ++i) sub_set(after, nlw, idx(edges[2 * i]), idx(edges[2 * i + 1]), add[i] != 0);   // set_edge, :68-70
    if ((rc = count_sub(s, after, (uint32_t)list.size(), nlw, cpost, &lpost))) return rc;                            // :71
    SlotPlan plan;
    if ((rc = plan_slots(s, list, before, after, nlw, edges, n, plan))) return rc;
    if ((rc = update_counts(s, chain, cpre, lpre, cpost, lpost))) return rc;     // :64-67, :72-77
    if ((rc = set_edges_on_device(s, chain, edges, add, n, false))) return rc;
    if ((rc = commit_slots(s, chain, plan))) return rc;
    for (int d = 0; d < FCM_MAX_COUNTS; ++d) { pre[d] = d < lpre ? cpre[d] : 0; post[d] = d < lpost ? cpost[d] : 0; }
    *pre_len = lpre; *post_len = lpost;
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_sampler_revert_transition(fcm_sampler *s, uint32_t chain, const fcm_node *edges, const int32_t *add, uint32_t n,
                                             const uint64_t *pre, int32_t pre_len, const uint64_t *post, int32_t post_len)
try {
    int rc = check_transition_args(s, chain, edges, add, n);
    if (rc) return rc;
    if ((pre_len && !pre) || (post_len && !post) || pre_len < 0 || post_len < 0 || pre_len > FCM_MAX_COUNTS || post_len > FCM_MAX_COUNTS)
        return fail(FCM_ERR_INVALID, "bad (pre, post)");
    if ((rc = fcm_sampler_sync(s))) return rc;
    // the pairs' bits now and after set_edge(a, b, !add) (:82-84): for the slot list only -- a revert counts nothing
    std::vector<uint32_t> list, before, after;
    for (uint32_t i = 0; i < n; ++i) { list.push_back(edges[2 * i]); list.push_back(edges[2 * i + 1]); }
    std::sort(list.begin(), list.end());
    list.erase(std::unique(list.begin(), list.end()), list.end());
    uint32_t nlw = 0;
    if ((rc = gather_sub(s, chain, list, before, nlw))) return rc;
    after = before;
    auto idx = [&](uint32_t v) { return (uint32_t)(std::lower_bound(list.begin(), list.end(), v) - list.begin()); };
    for (uint32_t i = 0; i < n; ++i) sub_set(after, nlw, idx(edges[2 * i]), idx(edges[2 * i + 1]), add[i] == 0);
    SlotPlan plan;
    if ((rc = plan_slots(s, list, before, after, nlw, edges, n, plan))) return rc;
    if ((rc = update_counts(s, chain, post, post_len, pre, pre_len))) return rc;   // :85-94
    if ((rc = set_edges_on_device(s, chain, edges, add, n, true))) return rc;
    return commit_slots(s, chain, plan);
} FCM_CATCH

// ---------------------------------------------------------------------------
// The batched State API: one transition per chain, every chain in one launch (fcm_apply_batch_kernel, fcm_count.hip).
// ---------------------------------------------------------------------------
static int ensure_batch_buffers(fcm_sampler *s)
{
    if (s->d_bt_pair.p) return FCM_OK;
    const size_t C = s->params.nchains;
    int rc;
    if ((rc = s->d_bt_pair.alloc(C * 4)) || (rc = s->d_bt_ops.alloc(C * 4)) || (rc = s->d_bt_pre.alloc(C * FCM_MAX_COUNTS * 8)) || (rc = s->d_bt_post.alloc(C * FCM_MAX_COUNTS * 8))
        || (rc = s->d_bt_lens.alloc(C * 8)) || (rc = s->d_bt_status.alloc(C * 4)) || (rc = s->d_bt_x.alloc(C * 8)) || (rc = s->d_bt_out.alloc(C * 12))) return rc;
    return FCM_OK;
}

// The change edges of one transition folded into what the kernel takes: the one adjacent pair they lie on and, per direction,
// keep / set / clear (set_edge calls in order: the last one on a direction decides; `invert`: a revert sets !add).  Returns
// 0 = folded, 1 = the edges lie on several pairs (one-chain path), or a failing status (out of range, not adjacent).
static int fold_transition(const fcm_sampler *s, const fcm_node *edges, const int32_t *add, uint32_t n, bool invert, uint32_t &pair, uint32_t &ops)
{
    pair = FCM_TR_SKIP; ops = 0u;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t a = edges[2 * i], b = edges[2 * i + 1];
        if (a >= s->n || b >= s->n || a == b) return -FCM_ERR_INVALID;
        const int64_t e = pair_index(s, a, b);
        if (e < 0) return -FCM_ERR_PANIC;                              // the reference indexes edge_neighborhood with it (src/lib.rs:104)
        if (pair != FCM_TR_SKIP && pair != (uint32_t)e) return 1;
        pair = (uint32_t)e;
        const uint32_t code = ((add[i] != 0) != invert) ? 1u : 2u, sh = a > b ? 0u : 2u;   // a > b: the direction big -> small
        ops = (ops & ~(3u << sh)) | (code << sh);
    }
    return 0;
}

static int batch_transitions(fcm_sampler *s, const fcm_node *edges, const int32_t *add, const uint32_t *m, uint32_t m_cap, bool revert,
                             uint64_t *pre, int32_t *pre_len, uint64_t *post, int32_t *post_len, int32_t *status)
{
    if (!s || !m || !pre || !pre_len || !post || !post_len) return fail(FCM_ERR_INVALID, "NULL argument");
    const uint32_t C = s->params.nchains;
    int rc = fcm_sampler_sync(s);
    if (rc) return rc;
    if ((rc = ensure_batch_buffers(s))) return rc;
    std::vector<uint32_t> pair(C, FCM_TR_SKIP), ops(C, 0u), lens(2 * (size_t)C, 0u), dst(C, 0u);
    std::vector<int32_t> code(C, FCM_OK);
    std::vector<uint8_t> host_path(C, 0);
    bool any_kernel = false;
    for (uint32_t c = 0; c < C; ++c) {
        if (m[c] > m_cap || (m[c] && (!edges || !add))) return fail(FCM_ERR_INVALID, "chain %u: %u change edges, room for %u", c, m[c], m_cap);
        if (revert && (pre_len[c] < 0 || post_len[c] < 0 || pre_len[c] > FCM_MAX_COUNTS || post_len[c] > FCM_MAX_COUNTS)) return fail(FCM_ERR_INVALID, "chain %u: bad (pre, post)", c);
        const int f = fold_transition(s, edges + (size_t)c * m_cap * 2, add + (size_t)c * m_cap, m[c], revert, pair[c], ops[c]);
        if (f < 0) { code[c] = -f; pair[c] = FCM_TR_SKIP; continue; }
        if (f == 1 || s->sparse) { host_path[c] = 1; pair[c] = FCM_TR_SKIP; continue; }
        if (pair[c] != FCM_TR_SKIP) any_kernel = true;
        if (revert) { lens[2 * (size_t)c] = (uint32_t)pre_len[c]; lens[2 * (size_t)c + 1] = (uint32_t)post_len[c]; }
    }
    if (any_kernel) {
        HIP_TRY(hipMemcpyAsync(s->d_bt_pair.p, pair.data(), (size_t)C * 4, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(s->d_bt_ops.p, ops.data(), (size_t)C * 4, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemsetAsync(s->d_bt_status.p, 0, (size_t)C * 4, s->stream));
        if (revert) {
            HIP_TRY(hipMemcpyAsync(s->d_bt_lens.p, lens.data(), (size_t)C * 8, hipMemcpyHostToDevice, s->stream));
            HIP_TRY(hipMemcpyAsync(s->d_bt_pre.p, pre, (size_t)C * FCM_MAX_COUNTS * 8, hipMemcpyHostToDevice, s->stream));
            HIP_TRY(hipMemcpyAsync(s->d_bt_post.p, post, (size_t)C * FCM_MAX_COUNTS * 8, hipMemcpyHostToDevice, s->stream));
        }
        FcmApplyParams ap;
        memset(&ap, 0, sizeof ap);
        ap.etab = s->params.etab; ap.nb = s->params.nb; ap.rows = s->params.rows; ap.counts = s->params.counts; ap.stats = s->params.stats;
        ap.pair = s->d_bt_pair.as<uint32_t>(); ap.ops = s->d_bt_ops.as<uint32_t>(); ap.pre = s->d_bt_pre.as<uint64_t>(); ap.post = s->d_bt_post.as<uint64_t>();
        ap.lens = s->d_bt_lens.as<uint32_t>(); ap.status = s->d_bt_status.as<uint32_t>();
        ap.rows_per_chain = s->params.rows_per_chain; ap.stride32 = s->params.stride32; ap.nchains = C; ap.ncounts = (uint32_t)s->params.ncounts; ap.revert = revert ? 1u : 0u;
        int lrc = fcm_launch_apply_batch(&ap, s->stream);
        if (lrc) return fail(FCM_ERR_HIP, "batched transition launch failed: %s", hipGetErrorString((hipError_t)lrc));
        HIP_TRY(hipMemcpyAsync(dst.data(), s->d_bt_status.p, (size_t)C * 4, hipMemcpyDeviceToHost, s->stream));
        if (!revert) {
            HIP_TRY(hipMemcpyAsync(lens.data(), s->d_bt_lens.p, (size_t)C * 8, hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipMemcpyAsync(pre, s->d_bt_pre.p, (size_t)C * FCM_MAX_COUNTS * 8, hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipMemcpyAsync(post, s->d_bt_post.p, (size_t)C * FCM_MAX_COUNTS * 8, hipMemcpyDeviceToHost, s->stream));
        }
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    int first = FCM_OK;
    for (uint32_t c = 0; c < C; ++c) {
        uint64_t *pc = pre + (size_t)c * FCM_MAX_COUNTS, *qc = post + (size_t)c * FCM_MAX_COUNTS;
        if (code[c] == FCM_OK && pair[c] != FCM_TR_SKIP) {
            switch (dst[c]) {
            case FCM_TRS_OK: break;
            case FCM_TRS_HOST: host_path[c] = 1; break;                // a local set beyond 64 vertices
            case FCM_TRS_ASSERT: code[c] = FCM_ERR_PANIC; fail(FCM_ERR_PANIC, "chain %u: the reference's assert!(*s >= *p) fires (src/lib.rs:65,86)", c); break;
            case FCM_TRS_UNSUPPORTED: code[c] = FCM_ERR_UNSUPPORTED; fail(FCM_ERR_UNSUPPORTED, "chain %u: the transition changes the number of reciprocal pairs or leaves the pair without an edge", c); break;
            case FCM_TRS_DEEP: code[c] = FCM_ERR_UNSUPPORTED; fail(FCM_ERR_UNSUPPORTED, "chain %u: simplices beyond dimension %d in the neighbourhood", c, FCM_MAX_COUNTS - 1); break;
            default: code[c] = FCM_ERR_INTERNAL; fail(FCM_ERR_INTERNAL, "chain %u: the pair is adjacent in the table and absent from the bitmap", c); break;
            }
            if (!revert && dst[c] != FCM_TRS_HOST) { pre_len[c] = (int32_t)lens[2 * (size_t)c]; post_len[c] = (int32_t)lens[2 * (size_t)c + 1]; }
        } else if (code[c] == FCM_OK && !host_path[c] && !revert) {    // the empty transition: both vectors empty (the subgraph on no vertices)
            for (int d = 0; d < FCM_MAX_COUNTS; ++d) pc[d] = qc[d] = 0;
            pre_len[c] = post_len[c] = 0;
        }
        if (host_path[c]) {                                            // several pairs, a wide neighbourhood, or the sparse state: the one-chain path
            const fcm_node *ec = edges + (size_t)c * m_cap * 2;
            const int32_t *ac = add + (size_t)c * m_cap;
            code[c] = revert ? fcm_sampler_revert_transition(s, c, ec, ac, m[c], pc, pre_len[c], qc, post_len[c])
                             : fcm_sampler_apply_transition(s, c, ec, ac, m[c], pc, &pre_len[c], qc, &post_len[c]);
        }
        if (code[c] == FCM_ERR_INVALID) fail(FCM_ERR_INVALID, "chain %u: a change edge is out of range or a loop", c);
        if (code[c] == FCM_ERR_PANIC && pair[c] == FCM_TR_SKIP && !host_path[c]) fail(FCM_ERR_PANIC, "chain %u: a change edge is not on an adjacent pair of pr(G) (src/lib.rs:104)", c);
        if (status) status[c] = code[c];
        if (first == FCM_OK && code[c] != FCM_OK) first = code[c];
    }
    return status ? FCM_OK : first;
}

extern "C" int fcm_sampler_apply_transitions(fcm_sampler *s, const fcm_node *edges, const int32_t *add, const uint32_t *m, uint32_t m_cap,
                                             uint64_t *pre, int32_t *pre_len, uint64_t *post, int32_t *post_len, int32_t *status)
try {
    return batch_transitions(s, edges, add, m, m_cap, false, pre, pre_len, post, post_len, status);
} FCM_CATCH

extern "C" int fcm_sampler_revert_transitions(fcm_sampler *s, const fcm_node *edges, const int32_t *add, const uint32_t *m, uint32_t m_cap,
                                              const uint64_t *pre, const int32_t *pre_len, const uint64_t *post, const int32_t *post_len, int32_t *status)
try {
    return batch_transitions(s, edges, add, m, m_cap, true, (uint64_t *)pre, (int32_t *)pre_len, (uint64_t *)post, (int32_t *)post_len, status);
} FCM_CATCH

// Transition::single_edge_flip on every chain at once: x[c] = one uniform 64-bit number per chain.
extern "C" int fcm_sampler_single_edge_flips(fcm_sampler *s, const uint64_t *x, fcm_node *edges /* [n_chains][2][2] */, int32_t *add /* [n_chains][2] */, uint32_t *n /* [n_chains] */)
try {
    if (!s || !x || !edges || !add || !n) return fail(FCM_ERR_INVALID, "NULL argument");
    const uint32_t C = s->params.nchains;
    int rc = fcm_sampler_sync(s);
    if (rc) return rc;
    if ((rc = ensure_batch_buffers(s))) return rc;
    HIP_TRY(hipMemcpyAsync(s->d_bt_x.p, x, (size_t)C * 8, hipMemcpyHostToDevice, s->stream));
    FcmFlipDrawParams fp;
    memset(&fp, 0, sizeof fp);
    fp.etab = s->params.etab; fp.rows = s->params.rows; fp.x = s->d_bt_x.as<uint64_t>(); fp.out = s->d_bt_out.as<uint32_t>();
    fp.rows_per_chain = s->params.rows_per_chain; fp.stride32 = s->params.stride32; fp.nchains = C; fp.U = s->params.U; fp.D = s->params.D; fp.sparse = s->sparse ? 1u : 0u;
    int lrc = fcm_launch_flip_draw(&fp, s->stream);
    if (lrc) return fail(FCM_ERR_HIP, "flip draw launch failed: %s", hipGetErrorString((hipError_t)lrc));
    std::vector<uint32_t> out(3 * (size_t)C);
    HIP_TRY(hipMemcpyAsync(out.data(), s->d_bt_out.p, out.size() * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (uint32_t c = 0; c < C; ++c) {
        if (out[3 * (size_t)c] == 0xFFFFFFFEu) return fail(FCM_ERR_INTERNAL, "chain %u: a pair is adjacent in the table and absent from the bitmap", c);
        n[c] = 0;
        if (out[3 * (size_t)c] == FCM_TR_SKIP) continue;                // empty transition (:297-298)
        const uint32_t from = out[3 * (size_t)c + 1], to = out[3 * (size_t)c + 2];
        fcm_node *e = edges + 4 * (size_t)c;
        e[0] = from; e[1] = to; add[2 * (size_t)c] = 0;                 // ([from,to], false), ([to,from], true) (:295)
        e[2] = to; e[3] = from; add[2 * (size_t)c + 1] = 1;
        n[c] = 2;
    }
    return FCM_OK;
} FCM_CATCH

// Transition::single_edge_flip (src/lib.rs:292-299) on the chain's current graph.  The reference draws a directed edge
// with the caller's rng; here the caller hands over one uniform 64-bit number x and the draw is the sampler's own
// (DESIGN.md 3): r = mulhi64(x, U + D) names a directed edge; a reciprocal pair gives the empty transition.
extern "C" int fcm_sampler_single_edge_flip(fcm_sampler *s, uint32_t chain, uint64_t x, fcm_node *edges /* [2][2] */, int32_t *add /* [2] */, uint32_t *n)
try {
    if (!s || !edges || !add || !n) return fail(FCM_ERR_INVALID, "NULL argument");
    if (chain >= s->params.nchains) return fail(FCM_ERR_INVALID, "chain %u out of range", chain);
    int rc = fcm_sampler_sync(s);
    if (rc) return rc;
    *n = 0;
    const uint64_t U = s->params.U, M = U + s->params.D;
    if (M == 0) return FCM_OK;                                                  // sample_edge -> None (:293)
    const uint64_t r = (uint64_t)(((unsigned __int128)x * M) >> 64);
    if (r >= U) return FCM_OK;                                                  // the second direction of a reciprocal pair: reverse present (:294)
    const uint32_t big = s->ue[2 * (size_t)r], small = s->ue[2 * (size_t)r + 1];
    const uint32_t *rows = s->d_rows.as<uint32_t>() + (size_t)chain * s->params.rows_per_chain;
    uint32_t wbs = 0, wsb = 0;
    bool bs, sb;
    if (s->sparse) {
        HIP_TRY(hipMemcpy(&wbs, rows + ((2 * r) >> 5), 4, hipMemcpyDeviceToHost));
        bs = (wbs >> ((2 * r) & 31)) & 1u; sb = (wbs >> ((2 * r + 1) & 31)) & 1u;
    } else {
        HIP_TRY(hipMemcpy(&wbs, rows + (size_t)big * s->stride32 + (small >> 5), 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&wsb, rows + (size_t)small * s->stride32 + (big >> 5), 4, hipMemcpyDeviceToHost));
        bs = (wbs >> (small & 31)) & 1u; sb = (wsb >> (big & 31)) & 1u;
    }
    if (bs == sb) {
        if (!bs) return fail(FCM_ERR_INTERNAL, "chain %u: pair (%u,%u) is adjacent in the table and absent from the bitmap", chain, big, small);
        return FCM_OK;                                                          // reciprocal: empty transition (:297-298)
    }
    const uint32_t from = bs ? big : small, to = bs ? small : big;
    edges[0] = from; edges[1] = to; add[0] = 0;                                 // ([from,to], false), ([to,from], true) (:295)
    edges[2] = to; edges[3] = from; add[1] = 1;
    *n = 2;
    return FCM_OK;
} FCM_CATCH

// ---------------------------------------------------------------------------
// Checkpoint / resume (role of src/io.rs:51-62)
// ---------------------------------------------------------------------------
static const char FCM_STATE_MAGIC[8] = {'F', 'C', 'M', 'S', 'T', 'A', 'T', '4'};

struct StateHeader {
    char magic[8];
    uint64_t sample_number;
    uint32_t n, n_chains;
    uint64_t U, D;
    int32_t ncounts, nstats;
    // the set this file belongs to when the chains of a run are saved as one file per handle (fcm_sampler_save_state_shard):
    // which shard of how many, the chains of the whole run and a number every file of one save carries (cfg.first_chain_id
    // says where this shard's chains sit).  A single-handle save is shard 0 of 1.
    uint32_t shard_index, shard_count;
    uint64_t total_chains, set_id;
    fcm_sampler_config cfg;
    fcm_bounds bounds;
};

template <class T> static bool wr(FILE *f, const T *p, size_t cnt) { return cnt == 0 || fwrite(p, sizeof(T), cnt, f) == cnt; }
template <class T> static bool rd(FILE *f, T *p, size_t cnt) { return cnt == 0 || fread(p, sizeof(T), cnt, f) == cnt; }

extern "C" int fcm_sampler_save_state(fcm_sampler *s, const char *path, uint64_t sample_number)
{
    return fcm_sampler_save_state_shard(s, path, sample_number, 0u, 1u, s ? (uint64_t)s->cfg.first_chain_id + s->params.nchains : 0u, 0u);   // (a set of one: chains first_chain_id .. of a run that ends there)
}

extern "C" int fcm_sampler_save_state_shard(fcm_sampler *s, const char *path, uint64_t sample_number, uint32_t shard_index, uint32_t shard_count,
                                            uint64_t total_chains, uint64_t set_id)
try {
    if (!s || !path) return fail(FCM_ERR_INVALID, "NULL argument");
    if (shard_count == 0 || shard_index >= shard_count || (uint64_t)s->cfg.first_chain_id + s->params.nchains > total_chains)
        return fail(FCM_ERR_INVALID, "shard %u of %u holding chains %u..%llu of %llu: not a shard of that run", shard_index, shard_count, s->cfg.first_chain_id,
                    (unsigned long long)s->cfg.first_chain_id + s->params.nchains, (unsigned long long)total_chains);
    std::vector<uint64_t> hs;
    int rc = fetch_stats(s, hs);   // syncs; refuses to save a chain whose device-side checks failed
    if (rc) return rc;
    const uint32_t C = s->params.nchains;
    const uint64_t U = s->ue.size() / 2, D = s->params.D;
    const uint64_t rec = (2 * U + 7) / 8;
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(FCM_ERR_IO, "cannot write %s", tmp.c_str());
    StateHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, FCM_STATE_MAGIC, 8);
    h.sample_number = sample_number;
    h.n = s->n; h.n_chains = C; h.U = U; h.D = D; h.ncounts = s->params.ncounts; h.nstats = FCM_NSTATS;
    h.shard_index = shard_index; h.shard_count = shard_count; h.total_chains = total_chains; h.set_id = set_id;
    h.cfg = s->cfg; h.bounds = s->bounds;
    bool ok = wr(f, &h, 1) && wr(f, s->ue.data(), s->ue.size());
    std::vector<uint64_t> hc((size_t)C * FCM_MAX_COUNTS);
    if (hipMemcpy(hc.data(), s->d_counts.p, hc.size() * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) ok = false;
    ok = ok && wr(f, hc.data(), hc.size()) && wr(f, hs.data(), hs.size());
    std::vector<uint8_t> bits((size_t)rec);
    std::vector<uint32_t> dbl((size_t)D);
    for (uint32_t c = 0; ok && c < C; ++c) {
        uint64_t nb = 0;
        if (fcm_sampler_get_edgebits(s, c, bits.data(), rec, &nb) != FCM_OK) { fclose(f); remove(tmp.c_str()); return FCM_ERR_HIP; }
        if (fcm_sampler_get_double_slots(s, c, dbl.data(), D, nullptr) != FCM_OK) { fclose(f); remove(tmp.c_str()); return FCM_ERR_HIP; }
        ok = wr(f, bits.data(), bits.size()) && wr(f, dbl.data(), dbl.size());
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { remove(tmp.c_str()); return fail(FCM_ERR_IO, "write to %s failed", tmp.c_str()); }
    if (rename(tmp.c_str(), path) != 0) return fail(FCM_ERR_IO, "moving temp state file to %s failed", path);
    return FCM_OK;
} FCM_CATCH

extern "C" int fcm_sampler_load_state(const char *path, int device, fcm_sampler **out, uint64_t *sample_number)
try {
    if (!path || !out) return fail(FCM_ERR_INVALID, "NULL argument");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(FCM_ERR_IO, "unable to load state %s", path);
    struct Closer { FILE *f; ~Closer() { if (f) fclose(f); } } closer{f};
    StateHeader h;
    if (!rd(f, &h, 1) || memcmp(h.magic, FCM_STATE_MAGIC, 8) != 0) return fail(FCM_ERR_IO, "%s is not a libfcm state file", path);
    if (h.n_chains == 0 || h.ncounts < 2 || h.ncounts > FCM_MAX_COUNTS || h.nstats != FCM_NSTATS || h.shard_count == 0 || h.shard_index >= h.shard_count
        || (uint64_t)h.cfg.first_chain_id + h.n_chains > h.total_chains || h.cfg.n_chains != h.n_chains)
        return fail(FCM_ERR_IO, "%s: corrupt header", path);
    // Every size below comes from the file: hold the header against the graph it claims and against the
    // file's length before anything is allocated from it.
    const uint64_t U = h.U, D = h.D;
    if (U > (uint64_t)h.n * (h.n ? h.n - 1 : 0) / 2 || D > U) return fail(FCM_ERR_IO, "%s: corrupt header (pair counts)", path);
    const uint64_t rec = (2 * U + 7) / 8;
    {
        if (fseek(f, 0, SEEK_END) != 0) return fail(FCM_ERR_IO, "%s: seek failed", path);
        const long fsz = ftell(f);
        if (fsz < 0 || fseek(f, (long)sizeof h, SEEK_SET) != 0) return fail(FCM_ERR_IO, "%s: seek failed", path);
        const unsigned __int128 want = (unsigned __int128)sizeof h + (unsigned __int128)U * 8u
            + (unsigned __int128)h.n_chains * ((FCM_MAX_COUNTS + FCM_NSTATS) * 8u)
            + (unsigned __int128)h.n_chains * ((unsigned __int128)rec + (unsigned __int128)D * 4u);
        if (want != (unsigned __int128)(uint64_t)fsz)
            return fail(FCM_ERR_IO, "%s: truncated or corrupt (length %ld does not match its header)", path, fsz);
    }
    std::vector<uint32_t> ue((size_t)U * 2);
    std::vector<uint64_t> hc((size_t)h.n_chains * FCM_MAX_COUNTS), hs((size_t)h.n_chains * FCM_NSTATS);
    if (!rd(f, ue.data(), ue.size()) || !rd(f, hc.data(), hc.size()) || !rd(f, hs.data(), hs.size()))
        return fail(FCM_ERR_IO, "%s: truncated", path);
    for (uint64_t e = 0; e < U; ++e)
        if (ue[2 * e] >= h.n || ue[2 * e + 1] >= ue[2 * e]) return fail(FCM_ERR_IO, "%s: corrupt pair list", path);

    // a graph with the right pr(G) (orientation of chain 0) to rebuild the static tables from
    std::vector<uint8_t> bits((size_t)rec);
    std::vector<uint32_t> dbl((size_t)D);
    const long chain_pos = ftell(f);
    if (!rd(f, bits.data(), bits.size())) return fail(FCM_ERR_IO, "%s: truncated", path);
    fcm_graph *g = nullptr;
    int rc = fcm_graph_new_disconnected(h.n, &g);
    if (rc) return rc;
    struct GG { fcm_graph *g; ~GG() { delete g; } } gg{g};
    auto apply_bits = [&](fcm_graph &gr) {
        std::fill(gr.rows.begin(), gr.rows.end(), 0u);
        gr.m = 0;
        for (uint64_t e = 0; e < U; ++e) {
            if ((bits[(2 * e) >> 3] >> ((2 * e) & 7)) & 1) gr.set(ue[2 * e], ue[2 * e + 1], true);
            if ((bits[(2 * e + 1) >> 3] >> ((2 * e + 1) & 7)) & 1) gr.set(ue[2 * e + 1], ue[2 * e], true);
        }
    };
    apply_bits(*g);
    fcm_sampler_config cfg = h.cfg;
    cfg.device = device;
    fcm_sampler *s = nullptr;
    if ((rc = fcm_sampler_create(g, &h.bounds, &cfg, &s))) return rc;
    struct SG { fcm_sampler *s; ~SG() { if (s) fcm_sampler_destroy(s); } } sg{s};
    if (s->params.U != U || s->params.ncounts != h.ncounts || s->params.nchains != h.n_chains)
        return fail(FCM_ERR_IO, "%s: state does not match the tables rebuilt from it", path);
    // per-chain state
    std::vector<uint8_t> seen((size_t)U);
    if (fseek(f, chain_pos, SEEK_SET) != 0) return fail(FCM_ERR_IO, "%s: seek failed", path);
    for (uint32_t c = 0; c < h.n_chains; ++c) {
        if (!rd(f, bits.data(), bits.size()) || !rd(f, dbl.data(), dbl.size())) return fail(FCM_ERR_IO, "%s: truncated", path);
        apply_bits(*g);
        if (s->sparse) {
            std::vector<uint32_t> w((size_t)s->bits_stride, 0u);
            memcpy(w.data(), bits.data(), bits.size());
            HIP_TRY(hipMemcpy(s->d_rows.as<uint32_t>() + (size_t)c * s->bits_stride, w.data(), w.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        } else {
            HIP_TRY(hipMemcpy(s->d_rows.as<uint32_t>() + (size_t)c * s->params.rows_per_chain, g->rows.data(),
                              (size_t)s->params.rows_per_chain * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        uint64_t nd = 0;
        for (uint64_t e = 0; e < U; ++e) nd += g->has(ue[2 * e], ue[2 * e + 1]) && g->has(ue[2 * e + 1], ue[2 * e]);
        if (nd != D) return fail(FCM_ERR_IO, "%s: chain %u has %llu reciprocal pairs, header says %llu", path, c,
                                 (unsigned long long)nd, (unsigned long long)D);
        // the slot list must name exactly the chain's reciprocal pairs, each once (the kernels index etab with it)
        std::fill(seen.begin(), seen.end(), (uint8_t)0);
        for (uint64_t j = 0; j < D; ++j) {
            const uint32_t e = dbl[j];
            if (e >= U || seen[e] || !(g->has(ue[2 * (size_t)e], ue[2 * (size_t)e + 1]) && g->has(ue[2 * (size_t)e + 1], ue[2 * (size_t)e])))
                return fail(FCM_ERR_IO, "%s: chain %u: corrupt slot list (entry %llu)", path, c, (unsigned long long)j);
            seen[e] = 1;
        }
        if (D) HIP_TRY(hipMemcpy(s->d_dbl.as<uint32_t>() + (size_t)c * s->params.dbl_stride, dbl.data(), (size_t)D * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (s->clique_moves && U) {   // inverse of the slot list
            std::vector<uint32_t> so((size_t)U, 0xFFFFFFFFu);
            for (uint64_t j = 0; j < D; ++j) so[dbl[j]] = (uint32_t)j;
            HIP_TRY(hipMemcpy(s->d_slot_of.as<uint32_t>() + (size_t)c * U, so.data(), (size_t)U * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
    HIP_TRY(hipMemcpy(s->d_counts.p, hc.data(), hc.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_stats.p, hs.data(), hs.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (sample_number) *sample_number = h.sample_number;
    sg.s = nullptr;
    *out = s;
    return FCM_OK;
} FCM_CATCH


// The header of a state file, without loading it: what a caller that saved one file per handle needs to put the set back together.
extern "C" int fcm_state_file_info(const char *path, fcm_state_info *out)
try {
    if (!path || !out) return fail(FCM_ERR_INVALID, "NULL argument");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(FCM_ERR_IO, "unable to load state %s", path);
    StateHeader h;
    const bool ok = rd(f, &h, 1);
    fclose(f);
    if (!ok || memcmp(h.magic, FCM_STATE_MAGIC, 8) != 0) return fail(FCM_ERR_IO, "%s is not a libfcm state file", path);
    if (h.n_chains == 0 || h.shard_count == 0 || h.shard_index >= h.shard_count || (uint64_t)h.cfg.first_chain_id + h.n_chains > h.total_chains)
        return fail(FCM_ERR_IO, "%s: corrupt header", path);
    out->sample_number = h.sample_number; out->total_chains = h.total_chains; out->set_id = h.set_id; out->seed = h.cfg.seed;
    out->n = h.n; out->n_chains = h.n_chains; out->first_chain_id = h.cfg.first_chain_id; out->shard_index = h.shard_index; out->shard_count = h.shard_count;
    return FCM_OK;
} FCM_CATCH

// Diagnostic: the per-chain cycle sums a -DFCM_STAMP build accumulates (zeros in the product build).
extern "C" int fcm_sampler_debug_stamps(fcm_sampler *s, uint64_t *out /* [n_chains][8] */)
try {
    if (!s || !out) return fail(FCM_ERR_INVALID, "NULL argument");
    int rc = fcm_sampler_sync(s);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, s->d_dbg.p, (size_t)s->params.nchains * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return FCM_OK;
} FCM_CATCH
