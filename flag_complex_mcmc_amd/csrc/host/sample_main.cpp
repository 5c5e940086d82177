// sample_main.cpp — the `sample` binary of the reference (src/bin/sample.rs),
// driving a batch of chains on one GPU through libfcm.so.
//
// Same options as the reference's clap Args (src/bin/sample.rs:21-78) where
// they apply, plus --chains / --device / --devices / --dim-cap.  --devices 0,1,..: the chains are sharded over
// the listed devices, one handle and one host thread each (fcm::MultiDeviceSampler); output is laid out as with one
// device, chain for chain.  Differences, all forced
// by scope: samples are written as edgebits only (--save-bits is
// implied; HDF5 is out of scope).
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "directed_scm.hpp"

struct Args {
    std::string input, label, continue_from, samples_store_dir = "./samples/", state_store_dir = "./state/";
    double target_relaxation = 0.01;
    size_t number_of_samples = 1000, sample_distance = 0, state_save_interval = 100;
    uint64_t seed = 0;
    bool simple = false;
    uint32_t chains = 1;
    int device = 0, dim_cap = 0;
    std::vector<int> devices;   // --devices a,b,..: empty = the one --device
};

static void usage()
{
    fprintf(stderr,
            "MCMC sampler for flag complexes of a directed graph (GPU batch)\n"
            "  -i, --input <flag file>         -l, --label <label>\n"
            "  -t, --target-relaxation <r>     [0.01]     -n, --number-of-samples <n> [1000]\n"
            "  -s, --seed <seed> [0]           --sample-distance <d> [0 = 2 E log2 E]\n"
            "  -c, --continue-from <state>     --samples-store-dir <dir> [./samples/]\n"
            "  --state-store-dir <dir> [./state/]   --state-save-interval <k> [100]\n"
            "  --simple   only single edge flips and double edge moves     --save-bits (implied)\n"
            "  --chains <n> [1]   --device <d> [0]   --devices <d0,d1,..> (chains sharded over them)   --dim-cap <d> [0 = lossless]\n");
}

static bool parse(int argc, char **argv, Args &a)
{
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", k.c_str()); exit(2); } return argv[++i]; };
        if (k == "-i" || k == "--input") a.input = val();
        else if (k == "-l" || k == "--label") a.label = val();
        else if (k == "-t" || k == "--target-relaxation" || k == "--target_relaxation") a.target_relaxation = atof(val());
        else if (k == "-n" || k == "--number-of-samples" || k == "--number_of_samples") a.number_of_samples = strtoull(val(), nullptr, 10);
        else if (k == "-s" || k == "--seed") a.seed = strtoull(val(), nullptr, 10);
        else if (k == "--sample-distance" || k == "--sample_distance") a.sample_distance = strtoull(val(), nullptr, 10);
        else if (k == "-c" || k == "--continue-from" || k == "--continue_from") a.continue_from = val();
        else if (k == "--samples-store-dir" || k == "--samples_store_dir") a.samples_store_dir = val();
        else if (k == "--state-store-dir" || k == "--state_store_dir") a.state_store_dir = val();
        else if (k == "--state-save-interval" || k == "--state_save_interval") a.state_save_interval = strtoull(val(), nullptr, 10);
        else if (k == "--save-bits" || k == "--save_bits") {}
        else if (k == "--simple") a.simple = true;
        else if (k == "--chains") a.chains = (uint32_t)strtoul(val(), nullptr, 10);
        else if (k == "--device") a.device = atoi(val());
        else if (k == "--devices") {
            const std::string v = val();
            a.devices.clear();
            for (size_t i = 0; i < v.size();) {
                size_t j = v.find(',', i);
                if (j == std::string::npos) j = v.size();
                if (j > i) a.devices.push_back(atoi(v.substr(i, j - i).c_str()));
                i = j + 1;
            }
            if (a.devices.empty()) { fprintf(stderr, "--devices needs a list\n"); return false; }
        }
        else if (k == "--dim-cap") a.dim_cap = atoi(val());
        else if (k == "-h" || k == "--help") { usage(); exit(0); }
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return false; }
    }
    return true;
}

// std::fs::create_dir_all (src/bin/sample.rs:109-110): every missing component, existing ones are fine
static void create_dir_all(const std::string &path)
{
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') mkdir(path.substr(0, i).c_str(), 0777);
}

static void print_counts(const char *what, const std::vector<size_t> &v)
{
    printf("%s[", what);
    for (size_t i = 0; i < v.size(); ++i) printf("%s%zu", i ? ", " : "", v[i]);
    printf("]\n");
}

// initialize_new_sampler, src/bin/sample.rs:80-105
static fcm::MultiDeviceSampler initialize_new_sampler(const Args &args, const std::vector<int> &devices)
{
    fcm::Graph g = fcm::io::read_flag_file(args.input);
    printf("initial flagser\n");
    const std::vector<size_t> flag_count = g.flagser_count(devices[0]);           // State::new, src/lib.rs:51
    const fcm::Bounds target = fcm::Bounds::target(flag_count, args.target_relaxation);
    const fcm::Bounds bounds = fcm::Bounds::calculate(g, flag_count, target, devices[0]);
    print_counts("  s^--: ", bounds.flag_count_min);
    print_counts("   s^-: ", target.flag_count_min);
    print_counts("  s(G): ", flag_count);
    print_counts("   s^+: ", target.flag_count_max);
    print_counts("  s^++: ", bounds.flag_count_max);
    // src/bin/sample.rs:101: --simple selects [0.5,0.5,0,0], the default is [0.1,0.1,0.6,0.2]
    fcm::MultiDeviceSampler s(g, bounds, args.chains, args.seed, args.simple ? fcm::MOVE_DISTRIBUTION_SIMPLE : fcm::MOVE_DISTRIBUTION,
                              args.sample_distance, devices, args.dim_cap);
    printf("The sampling distance was set to %zu.\n", s.sample_distance());
    return s;
}

int main(int argc, char **argv)
{
    Args args;
    if (!parse(argc, argv, args)) { usage(); return 2; }
    if (args.continue_from.empty() && (args.input.empty() || args.label.empty())) { usage(); return 2; }
    const std::vector<int> devices = args.devices.empty() ? std::vector<int>{args.device} : args.devices;
    try {
        create_dir_all(args.state_store_dir);
        create_dir_all(args.samples_store_dir);
        size_t sample_index_start = 0;
        std::unique_ptr<fcm::MultiDeviceSampler> sampler;
        if (!args.continue_from.empty()) {                                       // src/bin/sample.rs:114-115
            auto ls = fcm::MultiDeviceSampler::load_state(args.continue_from, devices);
            sample_index_start = ls.first;
            sampler.reset(new fcm::MultiDeviceSampler(std::move(ls.second)));
        } else {
            sampler.reset(new fcm::MultiDeviceSampler(initialize_new_sampler(args, devices)));
        }
        if (sampler->n_shards() > 1) {
            printf("%u chains on %zu handles:", sampler->n_chains(), sampler->n_shards());
            for (size_t r = 0; r < sampler->n_shards(); ++r) printf(" device %d: %u", sampler->device(r), sampler->shard(r).n_chains());
            printf("\n");
        }
        char seedbuf[32];
        snprintf(seedbuf, sizeof seedbuf, "%03" PRIu64, args.seed);
        const std::string state_file = args.state_store_dir + "/sampler-" + args.label + "-" + seedbuf + ".state";
        const std::string run_dir = args.samples_store_dir + "/" + args.label + "-" + seedbuf;
        mkdir(run_dir.c_str(), 0777);
        std::vector<std::unique_ptr<fcm::io::BitOutput>> outs;
        for (uint32_t c = 0; c < sampler->n_chains(); ++c) {
            char cb[32];
            snprintf(cb, sizeof cb, "/chain%05u", c);
            fcm::Graph g0 = sampler->graph(c);
            outs.emplace_back(new fcm::io::BitOutput(g0, run_dir + cb));
        }
        const size_t sample_index_end = sample_index_start + args.number_of_samples;
        for (size_t i = sample_index_start; i < sample_index_end; ++i) {          // src/bin/sample.rs:128-144
            if (args.state_save_interval && i % args.state_save_interval == 0) {
                printf("saving state in step %zu\n", i);
                sampler->save_state(state_file, i);
            }
            sampler->next();                                                      // every shard at once, one host thread each
            const auto counts = sampler->flag_counts();                           // gathered on the host, global chain order
            const auto met = sampler->metrics();
            for (uint32_t c = 0; c < sampler->n_chains(); ++c) outs[c]->save(*sampler, c);
            print_counts("flag count: ", counts[0]);
            printf("[sample %zu] chain 0 acceptance_ratio = %.6f (%" PRIu64 "/%" PRIu64 "), %u chains\n", i,
                   met[0].acceptance_ratio(), met[0].accepted, met[0].sampled, sampler->n_chains());
        }
        for (auto &o : outs) o->flush();
        sampler->save_state(state_file, sample_index_end);                        // src/bin/sample.rs:146
    } catch (const fcm::Error &e) {
        fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
