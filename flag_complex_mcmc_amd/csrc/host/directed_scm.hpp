// directed_scm.hpp — C++ host-side mirror of the reference crate's interface
// for the hot path (reference `directed-scm`: src/lib.rs, src/io.rs), over the
// C ABI of libfcm.so.  Header-only; names and argument meaning follow the
// reference so that code written against it reads the same:
//
//   reference (Rust)                         here
//   flag_complex::Graph                      fcm::Graph
//   Bounds{flag_count_min, flag_count_max}   fcm::Bounds           (src/lib.rs:113-161)
//   MCMCSampler<R>                           fcm::MCMCSampler      (src/lib.rs:163-198), a batch of chains
//   io::read_flag_file / save_flag_file      fcm::io::...          (src/io.rs:18-48)
//   io::save_state / load_state              MCMCSampler::save_state / load_state (src/io.rs:51-62)
//   io::BitOutput                            fcm::io::BitOutput    (src/io.rs:128-212)
//   Transition{change_edges}                 fcm::Transition       (src/lib.rs:200-204, 292-299)
//   State::{apply,revert}_transition, ...    fcm::State            (src/lib.rs:61-111), a view of one chain
//   (one process, many OS threads: all_cxs.rs:33-38)   fcm::MultiDeviceSampler: one handle per device, one host thread each
//
// Rust panics (unwrap/expect/assert!) become fcm::Error exceptions.
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <ctime>
#include <unistd.h>
#include <thread>
#include <exception>
#include <memory>
#include <algorithm>
#include <utility>
#include <vector>

#include "../../../include/fcm.h"

namespace fcm {

using Node = fcm_node;                 // u32
using Edge = std::pair<Node, Node>;    // [from, to]

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc)
{
    if (rc != FCM_OK) throw Error(rc, fcm_last_error());
}

class Graph {
public:
    explicit Graph(fcm_graph *h) : h_(h) {}
    Graph(const Graph &) = delete;
    Graph &operator=(const Graph &) = delete;
    Graph(Graph &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~Graph() { fcm_graph_destroy(h_); }

    static Graph new_disconnected(size_t nnodes) { fcm_graph *h; check(fcm_graph_new_disconnected((uint32_t)nnodes, &h)); return Graph(h); }
    size_t nnodes() const { return fcm_graph_nnodes(h_); }
    bool has_edge(Node a, Node b) const { return fcm_graph_has_edge(h_, a, b) != 0; }
    void set_edge(Node a, Node b, bool present) { check(fcm_graph_set_edge(h_, a, b, present)); }
    void add_edge(Node a, Node b) { check(fcm_graph_add_edge(h_, a, b)); }
    void remove_edge(Node a, Node b) { check(fcm_graph_remove_edge(h_, a, b)); }
    std::vector<Edge> edges() const { return pairs(&fcm_graph_edges); }
    std::vector<Edge> undirected_edges() const { return pairs(&fcm_graph_undirected_edges); }
    /// directed flag complex cell counts, on the GPU
    std::vector<size_t> flagser_count(int device = 0) const
    {
        uint64_t c[FCM_MAX_COUNTS];
        int len = 0;
        check(fcm_graph_flagser_count(h_, device, c, FCM_MAX_COUNTS, &len));
        return std::vector<size_t>(c, c + len);
    }
    fcm_graph *raw() const { return h_; }

private:
    std::vector<Edge> pairs(int (*fn)(const fcm_graph *, fcm_node *, uint64_t, uint64_t *)) const
    {
        uint64_t m = 0;
        check(fn(h_, nullptr, 0, &m));
        std::vector<fcm_node> flat(2 * m + 2);
        check(fn(h_, flat.data(), m, &m));
        std::vector<Edge> out(m);
        for (uint64_t i = 0; i < m; ++i) out[i] = {flat[2 * i], flat[2 * i + 1]};
        return out;
    }
    fcm_graph *h_;
};

struct Bounds {
    std::vector<size_t> flag_count_min, flag_count_max;

    fcm_bounds c() const
    {
        fcm_bounds b{};
        if (flag_count_min.size() > FCM_MAX_COUNTS + 1 || flag_count_max.size() > FCM_MAX_COUNTS + 1)
            throw Error(FCM_ERR_UNSUPPORTED, "bounds longer than FCM_MAX_COUNTS+1");
        for (size_t i = 0; i < flag_count_min.size(); ++i) b.flag_count_min[i] = flag_count_min[i];
        for (size_t i = 0; i < flag_count_max.size(); ++i) b.flag_count_max[i] = flag_count_max[i];
        b.min_len = (int32_t)flag_count_min.size();
        b.max_len = (int32_t)flag_count_max.size();
        return b;
    }
    static Bounds from_c(const fcm_bounds &b)
    {
        Bounds r;
        r.flag_count_min.assign(b.flag_count_min, b.flag_count_min + b.min_len);
        r.flag_count_max.assign(b.flag_count_max, b.flag_count_max + b.max_len);
        return r;
    }
    /// src/bin/sample.rs:89-95
    static Bounds target(const std::vector<size_t> &flag_count, double target_relaxation)
    {
        std::vector<uint64_t> fc(flag_count.begin(), flag_count.end());
        fcm_bounds b;
        check(fcm_target_bounds(fc.data(), (int)fc.size(), target_relaxation, &b));
        return from_c(b);
    }
    /// Bounds::calculate (src/lib.rs:119-156)
    static Bounds calculate(const Graph &initial_graph, const std::vector<size_t> &initial_flag_count,
                            const Bounds &target_bounds, int device = 0)
    {
        std::vector<uint64_t> fc(initial_flag_count.begin(), initial_flag_count.end());
        const fcm_bounds t = target_bounds.c();
        fcm_bounds out;
        check(fcm_bounds_calculate(initial_graph.raw(), fc.data(), (int)fc.size(), &t, device, &out, nullptr, nullptr));
        return from_c(out);
    }
    /// Bounds::check (src/lib.rs:157-160)
    bool check_counts(const std::vector<size_t> &flag_count) const
    {
        std::vector<uint64_t> fc(flag_count.begin(), flag_count.end());
        const fcm_bounds b = c();
        return fcm_bounds_check(&b, fc.data(), (int)fc.size()) != 0;
    }
};

class State;

/// A batch of independent MCMCSampler chains on one GPU.
class MCMCSampler {
public:
    MCMCSampler(const Graph &g, const Bounds &bounds, uint32_t n_chains, uint64_t seed, const double (&move_weights)[4],
                size_t sample_distance = 0, int device = 0, int dim_cap = 0, uint32_t first_chain_id = 0)
    {
        fcm_sampler_config cfg{};
        cfg.n_chains = n_chains;
        cfg.first_chain_id = first_chain_id;
        cfg.seed = seed;
        for (int i = 0; i < 4; ++i) cfg.move_weights[i] = move_weights[i];
        cfg.sample_distance = sample_distance;
        cfg.dim_cap = dim_cap;
        cfg.device = device;
        const fcm_bounds b = bounds.c();
        check(fcm_sampler_create(g.raw(), &b, &cfg, &h_));
        refresh();
    }
    explicit MCMCSampler(fcm_sampler *h) : h_(h) { refresh(); }
    MCMCSampler(const MCMCSampler &) = delete;
    MCMCSampler(MCMCSampler &&o) noexcept : h_(o.h_), info_(o.info_) { o.h_ = nullptr; }
    ~MCMCSampler() { fcm_sampler_destroy(h_); }

    /// MCMCSampler::next (src/lib.rs:181-194): sample_distance proposals on every chain
    void next() { check(fcm_sampler_next(h_)); }
    void step(uint64_t n_proposals) { check(fcm_sampler_step(h_, n_proposals)); check(fcm_sampler_sync(h_)); }

    uint32_t n_chains() const { return info_.n_chains; }
    size_t sample_distance() const { return fcm_sampler_sample_distance(h_); }
    Bounds bounds() const { fcm_bounds b; check(fcm_sampler_get_bounds(h_, &b)); return Bounds::from_c(b); }

    /// state.flag_count of every chain, as the reference would print it
    std::vector<std::vector<size_t>> flag_counts()
    {
        const int nc = fcm_sampler_ncounts(h_);
        std::vector<uint64_t> flat((size_t)n_chains() * nc);
        std::vector<int32_t> len(n_chains());
        check(fcm_sampler_get_counts(h_, flat.data(), len.data()));
        std::vector<std::vector<size_t>> out(n_chains());
        for (uint32_t c = 0; c < n_chains(); ++c) out[c].assign(flat.begin() + (size_t)c * nc, flat.begin() + (size_t)c * nc + len[c]);
        return out;
    }
    struct Metrics { uint64_t sampled, accepted; double acceptance_ratio() const { return (double)accepted / (double)sampled; } };
    std::vector<Metrics> metrics()
    {
        std::vector<uint64_t> st((size_t)n_chains() * FCM_NSTATS);
        check(fcm_sampler_get_stats(h_, st.data()));
        std::vector<Metrics> out(n_chains());
        for (uint32_t c = 0; c < n_chains(); ++c) out[c] = {st[(size_t)c * FCM_NSTATS + FCM_STAT_SAMPLED], st[(size_t)c * FCM_NSTATS + FCM_STAT_ACCEPTED]};
        return out;
    }
    std::vector<uint8_t> edgebits(uint32_t chain)
    {
        uint64_t n = 0;
        check(fcm_sampler_get_edgebits(h_, chain, nullptr, 0, &n));
        std::vector<uint8_t> out(n);
        check(fcm_sampler_get_edgebits(h_, chain, out.data(), n, &n));
        return out;
    }
    Graph graph(uint32_t chain)
    {
        uint64_t m = 0;
        check(fcm_sampler_get_edges(h_, chain, nullptr, 0, &m));
        std::vector<fcm_node> flat(2 * m + 2);
        check(fcm_sampler_get_edges(h_, chain, flat.data(), m, &m));
        fcm_graph *g;
        check(fcm_graph_from_edges(info_.n, m, flat.data(), &g));
        return Graph(g);
    }
    /// io::save_state / io::load_state (src/io.rs:51-62)
    void save_state(const std::string &fname, size_t sample_number) { check(fcm_sampler_save_state(h_, fname.c_str(), sample_number)); }
    static std::pair<size_t, MCMCSampler> load_state(const std::string &fname, int device = 0)
    {
        fcm_sampler *h;
        uint64_t n = 0;
        check(fcm_sampler_load_state(fname.c_str(), device, &h, &n));
        return {(size_t)n, MCMCSampler(h)};
    }
    const fcm_sampler_info &info() const { return info_; }
    fcm_sampler *raw() const { return h_; }
    State state(uint32_t chain);   // the chain's `State` (MCMCSampler::state, src/lib.rs:166)

private:
    void refresh() { check(fcm_sampler_get_info(h_, &info_)); }
    fcm_sampler *h_ = nullptr;
    fcm_sampler_info info_{};
};

/// `Transition { change_edges: Vec<([Node; 2], bool)> }` (src/lib.rs:200-204); true = add the edge
struct Transition {
    std::vector<std::pair<Edge, bool>> change_edges;
    void flat(std::vector<fcm_node> &e, std::vector<int32_t> &a) const
    {
        e.clear(); a.clear();
        for (const auto &c : change_edges) { e.push_back(c.first.first); e.push_back(c.first.second); a.push_back(c.second ? 1 : 0); }
    }
};
using Counters = std::pair<std::vector<size_t>, std::vector<size_t>>;   // (pre, post) of apply_transition

/// The reference's `State` (src/lib.rs:29-112) of one chain of a sampler: a view, the data stays on the GPU.
class State {
public:
    State(MCMCSampler &s, uint32_t chain) : s_(s), chain_(chain) {}
    std::vector<size_t> flag_count() { return s_.flag_counts()[chain_]; }
    Graph graph() { return s_.graph(chain_); }
    /// State::edgeset_neighborhood (src/lib.rs:99-111)
    std::vector<Node> edgeset_neighborhood(const std::vector<Edge> &edges)
    {
        std::vector<fcm_node> flat;
        for (const auto &e : edges) { flat.push_back(e.first); flat.push_back(e.second); }
        uint64_t k = 0;
        check(fcm_sampler_edgeset_neighborhood(s_.raw(), flat.data(), (uint32_t)edges.size(), nullptr, 0, &k));
        std::vector<Node> out(k);
        check(fcm_sampler_edgeset_neighborhood(s_.raw(), flat.data(), (uint32_t)edges.size(), out.data(), k, &k));
        return out;
    }
    /// State::apply_transition (src/lib.rs:61-79)
    Counters apply_transition(const Transition &t)
    {
        std::vector<fcm_node> e; std::vector<int32_t> a;
        t.flat(e, a);
        uint64_t pre[FCM_MAX_COUNTS], post[FCM_MAX_COUNTS];
        int32_t pl = 0, ql = 0;
        check(fcm_sampler_apply_transition(s_.raw(), chain_, e.data(), a.data(), (uint32_t)a.size(), pre, &pl, post, &ql));
        return {std::vector<size_t>(pre, pre + pl), std::vector<size_t>(post, post + ql)};
    }
    /// State::revert_transition (src/lib.rs:81-95)
    void revert_transition(const Transition &t, const Counters &c)
    {
        std::vector<fcm_node> e; std::vector<int32_t> a;
        t.flat(e, a);
        std::vector<uint64_t> pre(c.first.begin(), c.first.end()), post(c.second.begin(), c.second.end());
        check(fcm_sampler_revert_transition(s_.raw(), chain_, e.data(), a.data(), (uint32_t)a.size(), pre.data(), (int32_t)pre.size(),
                                            post.data(), (int32_t)post.size()));
    }
    /// Transition::single_edge_flip (src/lib.rs:292-299); x = one uniform 64-bit number from the caller's rng
    Transition single_edge_flip(uint64_t x)
    {
        fcm_node e[4]; int32_t a[2]; uint32_t n = 0;
        check(fcm_sampler_single_edge_flip(s_.raw(), chain_, x, e, a, &n));
        Transition t;
        for (uint32_t i = 0; i < n; ++i) t.change_edges.push_back({{e[2 * i], e[2 * i + 1]}, a[i] != 0});
        return t;
    }

private:
    MCMCSampler &s_;
    uint32_t chain_;
};

/// Chains sharded over several devices of one node: one handle per entry of `devices`, each driven by its own host
/// thread (the reference's only multi-chain precedent runs its States on OS threads: src/bin/all_cxs.rs:33-38).
/// Global chain c belongs to shard c / ceil(C/G) and draws from stream (seed, c), so every result equals the
/// single-handle run's, chain for chain, whatever the device list is (a device may be named more than once).
class MultiDeviceSampler {
public:
    MultiDeviceSampler(const Graph &g, const Bounds &bounds, uint32_t n_chains, uint64_t seed, const double (&move_weights)[4],
                       size_t sample_distance, const std::vector<int> &devices, int dim_cap = 0)
    {
        const uint32_t G = (uint32_t)devices.size();
        if (G == 0) throw Error(FCM_ERR_INVALID, "no devices");
        const uint32_t per = (n_chains + G - 1) / G;
        std::vector<std::pair<uint32_t, uint32_t>> ranges;
        for (uint32_t r = 0; r < G; ++r) {
            const uint32_t lo = std::min(r * per, n_chains), hi = std::min(lo + per, n_chains);
            if (hi > lo) { ranges.push_back({lo, hi}); devices_.push_back(devices[r]); }
        }
        shards_.resize(ranges.size());
        parallel([&](size_t r) {
            shards_[r].reset(new MCMCSampler(g, bounds, ranges[r].second - ranges[r].first, seed, move_weights, sample_distance,
                                             devices_[r], dim_cap, ranges[r].first));
        });
        for (auto &r : ranges) first_.push_back(r.first);
        n_chains_ = n_chains;
    }
    /// resumed shards (load_state of each shard's file)
    explicit MultiDeviceSampler(std::vector<std::unique_ptr<MCMCSampler>> shards, std::vector<int> devices) : shards_(std::move(shards)), devices_(std::move(devices))
    {
        uint32_t at = 0;
        for (auto &s : shards_) { first_.push_back(at); at += s->n_chains(); }
        n_chains_ = at;
    }
    uint32_t n_chains() const { return n_chains_; }
    size_t n_shards() const { return shards_.size(); }
    MCMCSampler &shard(size_t r) { return *shards_[r]; }
    int device(size_t r) const { return devices_[r]; }
    size_t sample_distance() const { return shards_[0]->sample_distance(); }
    /// MCMCSampler::next on every shard at once
    void next() { parallel([&](size_t r) { shards_[r]->next(); }); }
    void step(uint64_t n) { parallel([&](size_t r) { shards_[r]->step(n); }); }
    /// the host-side gather: per-chain vectors of all shards in global chain order
    std::vector<std::vector<size_t>> flag_counts()
    {
        std::vector<std::vector<std::vector<size_t>>> part(shards_.size());
        parallel([&](size_t r) { part[r] = shards_[r]->flag_counts(); });
        std::vector<std::vector<size_t>> out;
        for (auto &p : part) out.insert(out.end(), p.begin(), p.end());
        return out;
    }
    std::vector<MCMCSampler::Metrics> metrics()
    {
        std::vector<std::vector<MCMCSampler::Metrics>> part(shards_.size());
        parallel([&](size_t r) { part[r] = shards_[r]->metrics(); });
        std::vector<MCMCSampler::Metrics> out;
        for (auto &p : part) out.insert(out.end(), p.begin(), p.end());
        return out;
    }
    /// (shard, local chain) of a global chain id
    std::pair<size_t, uint32_t> locate(uint32_t chain) const
    {
        size_t r = 0;
        while (r + 1 < first_.size() && first_[r + 1] <= chain) ++r;
        return {r, chain - first_[r]};
    }
    std::vector<uint8_t> edgebits(uint32_t chain) { const auto l = locate(chain); return shards_[l.first]->edgebits(l.second); }
    Graph graph(uint32_t chain) { const auto l = locate(chain); return shards_[l.first]->graph(l.second); }
    /// One state file per shard: <fname> for a single shard (the single-device layout), <fname>.shard<r> otherwise.  Every file
    /// carries its place in the set (shard r of G, the run's chain count, the save's sample number as the set's id:
    /// fcm_sampler_save_state_shard), so that a resume can check it has ONE complete save in hand.
    static std::string shard_file(const std::string &fname, size_t r, size_t n_shards) { return n_shards == 1 ? fname : fname + ".shard" + std::to_string(r); }
    void save_state(const std::string &fname, size_t sample_number)
    {
        const size_t G = shards_.size();
        const uint64_t set_id = ((uint64_t)sample_number << 32) ^ (uint64_t)time(nullptr) ^ ((uint64_t)getpid() << 20);   // one number per save
        // all shards to <file>.new first, moved into place only once every one of them is written: a failure half-way leaves
        // the previous save whole (the moves themselves are not atomic as a set: the loader checks the set's id)
        parallel([&](size_t r) {
            check(fcm_sampler_save_state_shard(shards_[r]->raw(), (shard_file(fname, r, G) + ".new").c_str(), sample_number, (uint32_t)r, (uint32_t)G, n_chains_, set_id));
        });
        for (size_t r = 0; r < G; ++r) {
            const std::string f = shard_file(fname, r, G);
            if (rename((f + ".new").c_str(), f.c_str()) != 0) throw Error(FCM_ERR_IO, "moving " + f + ".new into place failed");
        }
    }
    /// Resume: the number of shards comes from the FILES, not from the device list -- shard r goes to devices[r % devices.size()]
    /// (a device may hold several shards, each with its own handle and host thread; a state saved on 4 handles resumes on 2
    /// devices or on 1, one saved on 2 resumes on 3 with the third device idle).  Fails loudly if a shard is missing, if the
    /// files are not of one save (set id, sample number, seed) or if their chain ranges do not tile 0 .. total-1.
    static std::pair<size_t, MultiDeviceSampler> load_state(const std::string &fname, const std::vector<int> &devices)
    {
        if (devices.empty()) throw Error(FCM_ERR_INVALID, "no devices");
        auto exists = [](const std::string &f) { FILE *t = fopen(f.c_str(), "rb"); if (!t) return false; fclose(t); return true; };
        fcm_state_info first{};
        const bool sharded = exists(fname + ".shard0");
        if (sharded && exists(fname)) {
            // both layouts on disk (a run saved on one handle, then on several, or the other way round): take the newer save, by sample number
            fcm_state_info one{}, many{};
            check(fcm_state_file_info(fname.c_str(), &one));
            check(fcm_state_file_info((fname + ".shard0").c_str(), &many));
            first = many.sample_number >= one.sample_number ? many : one;
        } else {
            check(fcm_state_file_info((sharded ? fname + ".shard0" : fname).c_str(), &first));
        }
        const size_t G = first.shard_count;   // (from the file, not from the device list)
        std::vector<fcm_state_info> info(G);
        uint64_t at = first.shard_index == 0 ? first.first_chain_id : 0;   // (a single handle's file may start at any global chain id: first_chain_id of its config)
        for (size_t r = 0; r < G; ++r) {
            const std::string f = shard_file(fname, r, G);
            if (!exists(f)) throw Error(FCM_ERR_IO, f + ": shard " + std::to_string(r) + " of " + std::to_string(G) + " is missing");
            check(fcm_state_file_info(f.c_str(), &info[r]));
            const fcm_state_info &h = info[r];
            if (h.shard_index != r || h.shard_count != G || h.total_chains != first.total_chains || h.set_id != first.set_id || h.sample_number != first.sample_number
                || h.seed != first.seed || h.n != first.n)
                throw Error(FCM_ERR_IO, f + ": not a shard of the same save as " + shard_file(fname, 0, G) + " (shard index / count, chains, set id, sample number, seed or graph differ)");
            if (h.first_chain_id != at) throw Error(FCM_ERR_IO, f + ": its chains start at " + std::to_string(h.first_chain_id) + ", the shards before it end at " + std::to_string(at));
            at += h.n_chains;
        }
        if (at != first.total_chains) throw Error(FCM_ERR_IO, fname + ": the shards hold " + std::to_string(at) + " chains, the run had " + std::to_string(first.total_chains));
        std::vector<std::unique_ptr<MCMCSampler>> shards(G);
        std::vector<int> devs(G);
        for (size_t r = 0; r < G; ++r) devs[r] = devices[r % devices.size()];
        {   // every shard on its own thread, like everything else here
            std::vector<std::exception_ptr> err(G);
            std::vector<std::thread> th;
            auto run = [&](size_t r) {
                try { auto ls = MCMCSampler::load_state(shard_file(fname, r, G), devs[r]); shards[r].reset(new MCMCSampler(std::move(ls.second))); }
                catch (...) { err[r] = std::current_exception(); }
            };
            size_t started = 1;
            try { for (; started < G; ++started) th.emplace_back(run, started); } catch (...) {}
            run(0);
            for (size_t r = started; r < G; ++r) run(r);
            for (auto &t : th) t.join();
            for (auto &e : err) if (e) std::rethrow_exception(e);
        }
        return {(size_t)first.sample_number, MultiDeviceSampler(std::move(shards), std::move(devs))};
    }

private:
    // fn(r) for every shard, each on its own thread; the first exception is rethrown here once all have been joined
    template <class F> void parallel(F fn)
    {
        const size_t G = shards_.size();
        std::vector<std::exception_ptr> err(G);
        std::vector<std::thread> th;
        auto run = [&](size_t r) { try { fn(r); } catch (...) { err[r] = std::current_exception(); } };
        size_t started = 1;
        try { for (; started < G; ++started) th.emplace_back(run, started); } catch (...) {}
        run(0);
        for (size_t r = started; r < G; ++r) run(r);
        for (auto &t : th) t.join();
        for (auto &e : err) if (e) std::rethrow_exception(e);
    }
    std::vector<std::unique_ptr<MCMCSampler>> shards_;
    std::vector<int> devices_;
    std::vector<uint32_t> first_;
    uint32_t n_chains_ = 0;
};

namespace io {
inline Graph read_flag_file(const std::string &fname) { fcm_graph *g; check(fcm_read_flag_file(fname.c_str(), &g)); return Graph(g); }
inline void save_flag_file(const std::string &fname, const Graph &g) { check(fcm_save_flag_file(fname.c_str(), g.raw())); }

/// io::BitOutput (src/io.rs:128-212): <dir>/graph.flag and <dir>/N.edgebits
class BitOutput {
public:
    BitOutput(const Graph &graph, const std::string &dir) : dir_(dir)
    {
        mkdir(dir.c_str(), 0777);  // AlreadyExists is fine (io.rs:143-146)
        save_flag_file(dir + "/graph.flag", graph);
        const size_t nslots = 2 * graph.undirected_edges().size();
        if (nslots / 8 == 0) throw Error(FCM_ERR_PANIC, "fewer than 8 edge slots: reference BitOutput::new divides by zero (src/io.rs:161)");
        chunk_size_ = std::max<size_t>(2000000000 / (nslots / 8), 1);
    }
    ~BitOutput() { if (f_) fclose(f_); }
    template <class Sampler> void save(Sampler &s, uint32_t chain)
    {
        if (index_in_file_ == 0) {
            f_ = fopen((dir_ + "/" + std::to_string(index_in_dir_) + ".edgebits").c_str(), "wb");
            if (!f_) throw Error(FCM_ERR_IO, "cannot create edgebits file in " + dir_);
        }
        const std::vector<uint8_t> rec = s.edgebits(chain);
        if (fwrite(rec.data(), 1, rec.size(), f_) != rec.size()) throw Error(FCM_ERR_IO, "edgebits write failed");
        if (++index_in_file_ == chunk_size_) { fclose(f_); f_ = nullptr; index_in_file_ = 0; ++index_in_dir_; }
    }
    void flush() { if (f_) fflush(f_); }

private:
    std::string dir_;
    size_t chunk_size_ = 1, index_in_file_ = 0, index_in_dir_ = 0;
    FILE *f_ = nullptr;
};
}  // namespace io

inline State MCMCSampler::state(uint32_t chain) { return State(*this, chain); }

static const double MOVE_DISTRIBUTION_SIMPLE[4] = {0.5, 0.5, 0.0, 0.0};  // src/bin/sample.rs:16
static const double MOVE_DISTRIBUTION[4] = {0.1, 0.1, 0.6, 0.2};         // src/bin/sample.rs:17

}  // namespace fcm
