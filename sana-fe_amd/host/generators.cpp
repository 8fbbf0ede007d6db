// generators.cpp -- helpers of include/sanafe_host.h that are not part of the chip runtime: the synthetic-network
// edge generator of the benchmark configurations and the YAML-subset -> JSON test hook.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sanafe_host.h"

namespace sanafe_amd
{
int host_fail(int code, const std::string &msg); // chip.cpp: records the message for sanafe_last_error()
}
namespace
{
int fail(int code, const std::string &msg) { return sanafe_amd::host_fail(code, msg); }
} // namespace

// ---------------------------------------------------------------------------------------------
// Synthetic-network generator for the benchmark configs (after
// scripts/tcad2025/random_network.py:63-105): every neuron gets `out_degree` distinct targets
// drawn uniformly (rejection against a per-thread bitmap) and an integer weight in
// {-8..8}\{0}; neuron i uses its own std::mt19937_64 stream seeded from (seed, i), so the
// result does not depend on the thread count.
// ---------------------------------------------------------------------------------------------
extern "C" int sanafe_generate_random_edges(int64_t n_neurons, int64_t out_degree, uint64_t seed, int n_threads,
        int64_t src_base, int64_t dst_base, int64_t *src, int64_t *dst, double *weight)
{
    if (n_neurons <= 0 || out_degree < 0 || out_degree > n_neurons || !src || !dst || !weight)
        return fail(SANAFE_HIP_ERR_INVALID, "bad generator arguments");
    n_threads = std::max(1, n_threads);
    auto work = [&](int tid) {
        std::vector<uint64_t> seen((n_neurons + 63) / 64, 0);
        for (int64_t i = tid; i < n_neurons; i += n_threads)
        {
            std::mt19937_64 gen(seed * 0x9E3779B97F4A7C15ull + static_cast<uint64_t>(i) + 1);
            int64_t *d = dst + i * out_degree;
            for (int64_t k = 0; k < out_degree; k++)
            {
                uint64_t r;
                do
                {
                    r = gen() % static_cast<uint64_t>(n_neurons);
                } while (seen[r >> 6] & (1ull << (r & 63)));
                seen[r >> 6] |= 1ull << (r & 63);
                d[k] = dst_base + static_cast<int64_t>(r);
                src[i * out_degree + k] = src_base + i;
                const uint64_t w = gen();
                const double mag = static_cast<double>(1 + (w % 8));
                weight[i * out_degree + k] = (w & (1ull << 40)) ? mag : -mag;
            }
            for (int64_t k = 0; k < out_degree; k++)
            {
                const uint64_t r = static_cast<uint64_t>(d[k] - dst_base);
                seen[r >> 6] &= ~(1ull << (r & 63));
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    return 0;
}

// Sharded variant for tile-sharded (multi-GPU) runs: keeps only the edges a rank needs, i.e. those
// whose source OR destination neuron lies in [lo, hi).  The per-neuron streams are the same as in
// sanafe_generate_random_edges, so every rank sees a consistent slice of one global network
// without ever holding all of it.
struct sanafe_edge_set
{
    std::vector<std::vector<int64_t>> src, dst;
    std::vector<std::vector<double>> w;
};

extern "C" int sanafe_generate_random_edges_sharded(int64_t n_neurons, int64_t out_degree, uint64_t seed, int n_threads,
        int64_t lo, int64_t hi, sanafe_edge_set **out, int64_t *count)
{
    if (n_neurons <= 0 || out_degree < 0 || out_degree > n_neurons || !out || !count)
        return fail(SANAFE_HIP_ERR_INVALID, "bad generator arguments");
    n_threads = std::max(1, n_threads);
    auto set = std::make_unique<sanafe_edge_set>();
    set->src.resize(n_threads);
    set->dst.resize(n_threads);
    set->w.resize(n_threads);
    auto work = [&](int tid) {
        std::vector<uint64_t> seen((n_neurons + 63) / 64, 0);
        std::vector<int64_t> targets(out_degree);
        const int64_t b = n_neurons * tid / n_threads, e = n_neurons * (tid + 1) / n_threads; // contiguous: output stays sorted by source
        {
            // expected number of kept edges (+2 %): no doubling slack in vectors that reach tens of GB
            const int64_t local_src = std::max<int64_t>(0, std::min(e, hi) - std::max(b, lo));
            const double frac_in = static_cast<double>(hi - lo) / static_cast<double>(n_neurons);
            const double expect = static_cast<double>(out_degree) * (local_src + (e - b - local_src) * frac_in);
            const size_t cap = static_cast<size_t>(expect * 1.02) + 4096;
            set->src[tid].reserve(cap);
            set->dst[tid].reserve(cap);
            set->w[tid].reserve(cap);
        }
        for (int64_t i = b; i < e; i++)
        {
            std::mt19937_64 gen(seed * 0x9E3779B97F4A7C15ull + static_cast<uint64_t>(i) + 1);
            const bool src_local = (i >= lo && i < hi);
            for (int64_t k = 0; k < out_degree; k++)
            {
                uint64_t r;
                do
                {
                    r = gen() % static_cast<uint64_t>(n_neurons);
                } while (seen[r >> 6] & (1ull << (r & 63)));
                seen[r >> 6] |= 1ull << (r & 63);
                targets[k] = static_cast<int64_t>(r);
                const uint64_t wv = gen();
                if (src_local || (targets[k] >= lo && targets[k] < hi))
                {
                    const double mag = static_cast<double>(1 + (wv % 8));
                    set->src[tid].push_back(i);
                    set->dst[tid].push_back(targets[k]);
                    set->w[tid].push_back((wv & (1ull << 40)) ? mag : -mag);
                }
            }
            for (int64_t k = 0; k < out_degree; k++) seen[targets[k] >> 6] &= ~(1ull << (targets[k] & 63));
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    int64_t total = 0;
    for (auto &v : set->src) total += static_cast<int64_t>(v.size());
    *count = total;
    *out = set.release();
    return 0;
}

// Locally connected variant -- the weak-scaling form of the recipe: every neuron draws its `out_degree` distinct
// targets uniformly from the `window` neurons centred on itself (ids wrap around), so the fan-in statistics of a core
// (sources per core, synapses per axon) do not change when the chip grows by whole windows; with window == n_neurons
// it is the uniform recipe.  Only sources within window / 2 of [lo, hi) can reach it, so a rank generates its slice
// of a chip of any size in time proportional to its own share.
extern "C" int sanafe_generate_random_edges_windowed(int64_t n_neurons, int64_t out_degree, uint64_t seed, int n_threads,
        int64_t window, int64_t lo, int64_t hi, sanafe_edge_set **out, int64_t *count)
{
    if (n_neurons <= 0 || window <= 0 || window > n_neurons || out_degree < 0 || out_degree > window || lo < 0 || hi > n_neurons || lo > hi ||
            !out || !count)
        return fail(SANAFE_HIP_ERR_INVALID, "bad generator arguments");
    n_threads = std::max(1, n_threads);
    auto set = std::make_unique<sanafe_edge_set>();
    set->src.resize(n_threads);
    set->dst.resize(n_threads);
    set->w.resize(n_threads);
    // candidate sources: [lo - window/2 - 1, hi + window/2 + 1), clipped to one lap of the ring
    const int64_t half = window / 2 + 1;
    const int64_t span = std::min<int64_t>(n_neurons, (hi - lo) + 2 * half);
    const int64_t first = ((lo - half) % n_neurons + n_neurons) % n_neurons;
    auto work = [&](int tid) {
        std::vector<uint64_t> seen((window + 63) / 64, 0);
        std::vector<int64_t> draws(out_degree);
        const int64_t b = span * tid / n_threads, e = span * (tid + 1) / n_threads;
        const size_t cap = static_cast<size_t>(static_cast<double>(out_degree) * static_cast<double>(e - b) * 0.75) + 4096;
        set->src[tid].reserve(cap);
        set->dst[tid].reserve(cap);
        set->w[tid].reserve(cap);
        for (int64_t q = b; q < e; q++)
        {
            const int64_t i = (first + q) % n_neurons;
            std::mt19937_64 gen(seed * 0x9E3779B97F4A7C15ull + static_cast<uint64_t>(i) + 1);
            const bool src_local = (i >= lo && i < hi);
            const int64_t base = i - window / 2;
            for (int64_t k = 0; k < out_degree; k++)
            {
                uint64_t r;
                do
                {
                    r = gen() % static_cast<uint64_t>(window);
                } while (seen[r >> 6] & (1ull << (r & 63)));
                seen[r >> 6] |= 1ull << (r & 63);
                draws[k] = static_cast<int64_t>(r);
                const int64_t t = ((base + static_cast<int64_t>(r)) % n_neurons + n_neurons) % n_neurons;
                const uint64_t wv = gen();
                if (src_local || (t >= lo && t < hi))
                {
                    const double mag = static_cast<double>(1 + (wv % 8));
                    set->src[tid].push_back(i);
                    set->dst[tid].push_back(t);
                    set->w[tid].push_back((wv & (1ull << 40)) ? mag : -mag);
                }
            }
            for (int64_t k = 0; k < out_degree; k++) seen[draws[k] >> 6] &= ~(1ull << (draws[k] & 63));
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    int64_t total = 0;
    for (auto &v : set->src) total += static_cast<int64_t>(v.size());
    *count = total;
    *out = set.release();
    return 0;
}

extern "C" int sanafe_edge_set_copy(sanafe_edge_set *set, int64_t *src, int64_t *dst, double *weight)
{
    if (!set || !src || !dst || !weight) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    int64_t o = 0;
    for (size_t t = 0; t < set->src.size(); t++)
    {
        const int64_t n = static_cast<int64_t>(set->src[t].size());
        std::copy(set->src[t].begin(), set->src[t].end(), src + o);
        std::copy(set->dst[t].begin(), set->dst[t].end(), dst + o);
        std::copy(set->w[t].begin(), set->w[t].end(), weight + o);
        // the set is consumed by the copy: give each part back as soon as it has been copied
        std::vector<int64_t>().swap(set->src[t]);
        std::vector<int64_t>().swap(set->dst[t]);
        std::vector<double>().swap(set->w[t]);
        o += n;
    }
    return 0;
}

extern "C" void sanafe_edge_set_free(sanafe_edge_set *set) { delete set; }

// ---------------------------------------------------------------------------------------------
// YAML subset reader: canonical JSON of a description file (tests compare it with PyYAML)
// ---------------------------------------------------------------------------------------------
#include "yaml_subset.hpp"
extern "C" char *sanafe_yaml_file_to_json(const char *path)
{
    try
    {
        const std::string js = sanafe_amd::yaml_to_json(sanafe_amd::yaml_parse_file(path));
        char *out = static_cast<char *>(std::malloc(js.size() + 1));
        std::memcpy(out, js.c_str(), js.size() + 1);
        return out;
    }
    catch (const std::exception &e)
    {
        (void) fail(SANAFE_HIP_ERR_INVALID, e.what());
        return nullptr;
    }
}
extern "C" void sanafe_free(void *p) { std::free(p); }
