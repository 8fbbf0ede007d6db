// sanafe_oracle.cpp -- CPU oracle for the SANA-FE per-timestep loop.
//
// TEST INFRASTRUCTURE ONLY (see sanafe_oracle.h).  This is a clean-room,
// single-threaded restatement of the reference's algorithm written from its
// observable behaviour; every function cites the reference file:line it
// follows.  It deliberately keeps the reference's object-per-core structure
// (units, mapped neurons, message lists) so that it shares NO mapping or
// lowering code with the MI355X product path it checks.
//
// Pinning (see DESIGN.md "Oracle"):
//  * unit models (a18-a25) are checked against the reference's own model TUs
//    compiled unmodified into oracle/_ref (tests/test_oracle_vs_ref_models.py)
//    and against the known answers of the reference's unit tests;
//  * the chip-level loop is checked against the run outputs the reference
//    holds (tutorial_5_dvs.ipynb: neurons_fired == 365277) and the survey's
//    recorded probe of the reference for example_chip/example_snn;
//  * the detailed NoC scheduler has no reference-held vector: parity unpinned.
#include <algorithm>
#include <atomic>
#include <exception>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <queue>
#include <random>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "sanafe_oracle.h"

namespace
{
constexpr double NEG_INF = -std::numeric_limits<double>::infinity();

enum Status : uint8_t { UNSET = 0, IDLE = 1, UPDATED = 2, FIRED = 3 };
enum ResetMode { RESET_NONE = 0, RESET_SOFT = 1, RESET_HARD = 2, RESET_SATURATE = 3 };

struct Opt
{
    bool has{false};
    double v{0.0};
    Opt() = default;
    Opt(double x) : has(true), v(x) {}
    double value_or(double d) const { return has ? v : d; }
};

// src/pipeline.hpp:59-67
struct Result
{
    Opt current;
    Status status{UNSET};
    Opt energy;
    Opt latency;
};

// src/attribute.hpp:41-176
struct Attr
{
    std::string key;
    int type{SANAFE_ATTR_DOUBLE};
    double num{0.0};
    std::string str;
    std::vector<double> list;
    int fwd{SANAFE_FWD_SYNAPSE | SANAFE_FWD_DENDRITE | SANAFE_FWD_SOMA};

    double as_double() const
    {
        if (type == SANAFE_ATTR_DOUBLE || type == SANAFE_ATTR_INT) return num;
        throw std::runtime_error("Error: Attribute " + key + " cannot be cast to a double");
    }
    int as_int() const
    {
        if (type != SANAFE_ATTR_INT) throw std::runtime_error("bad variant access (int): " + key);
        return static_cast<int>(num);
    }
    bool as_bool() const
    {
        if (type == SANAFE_ATTR_BOOL) return num != 0.0;
        if (type == SANAFE_ATTR_INT) return num != 0.0;
        throw std::runtime_error("Error: Attribute " + key + " cannot be cast to a bool ()");
    }
    const std::string &as_string() const
    {
        if (type != SANAFE_ATTR_STRING) throw std::runtime_error("bad variant access (string): " + key);
        return str;
    }
};

ResetMode parse_reset_mode(const std::string &s) // src/models.cpp:905-931
{
    if (s == "none") return RESET_NONE;
    if (s == "soft") return RESET_SOFT;
    if (s == "hard") return RESET_HARD;
    if (s == "saturate") return RESET_SATURATE;
    throw std::invalid_argument("Reset mode not recognized");
}

// ---------------------------------------------------------------------------
// Pipeline unit base: src/pipeline.hpp:69-233, src/pipeline.cpp
// ---------------------------------------------------------------------------
struct Unit
{
    std::string name, model;
    bool impl_syn{false}, impl_dend{false}, impl_soma{false};
    bool log_energy{false}, log_latency{false}, update_every_timestep{false};
    Opt def_energy_process_spike, def_latency_process_spike;
    Opt def_energy_update, def_latency_update;
    bool has_soma_energy{false}, has_soma_latency{false};
    double e_access{0}, e_update{0}, e_spike{0};
    double l_access{0}, l_update{0}, l_spike{0};
    double energy{0.0}, latency{0.0};
    long spikes_processed{0}, neurons_updated{0}, neurons_fired{0};
    long neuron_count{0}, connection_count{0};
    bool is_used{false};

    virtual ~Unit() = default;
    virtual void set_attr_hw(const Attr &) {}
    virtual void set_attr_neuron(size_t, const Attr &) {}
    virtual void set_attr_edge(size_t, const Attr &) {}
    virtual void reset() {}
    virtual Result update_syn(size_t, bool, long) { throw std::logic_error("Error: Synapse input not implemented"); }
    virtual Result update_dend(size_t, Opt, bool, size_t, long) { throw std::logic_error("Error: Dendrite input not implemented"); }
    virtual Result update_soma(size_t, Opt, long) { throw std::logic_error("Error: Soma input not implemented"); }
    virtual double get_potential(size_t) { return 0.0; }
    virtual std::map<std::string, double> get_traces(size_t) { return {}; }

    // src/pipeline.cpp:151-266
    void set_attributes_hw(const std::vector<Attr> &attrs)
    {
        std::map<std::string, const Attr *> m;
        for (const Attr &a : attrs) m[a.key] = &a;
        auto has = [&](const char *k) { return m.count(k) > 0; };
        if (has("energy_process_spike")) def_energy_process_spike = m["energy_process_spike"]->as_double();
        if (has("latency_process_spike")) def_latency_process_spike = m["latency_process_spike"]->as_double();
        if (has("energy_update")) def_energy_update = m["energy_update"]->as_double();
        if (has("latency_update")) def_latency_update = m["latency_update"]->as_double();
        const char *en[3] = {"energy_access_neuron", "energy_update_neuron", "energy_spike_out"};
        if (has(en[0]) || has(en[1]) || has(en[2]))
        {
            for (const char *k : en)
                if (!has(k)) throw std::invalid_argument(std::string("Metric not defined: ") + k);
            has_soma_energy = true;
            e_access = m[en[0]]->as_double();
            e_update = m[en[1]]->as_double();
            e_spike = m[en[2]]->as_double();
        }
        const char *ln[3] = {"latency_access_neuron", "latency_update_neuron", "latency_spike_out"};
        if (has(ln[0]) || has(ln[1]) || has(ln[2]))
        {
            for (const char *k : ln)
                if (!has(k)) throw std::invalid_argument(std::string("Missing metric: ") + k);
            has_soma_latency = true;
            l_access = m[ln[0]]->as_double();
            l_update = m[ln[1]]->as_double();
            l_spike = m[ln[2]]->as_double();
        }
        for (const auto &kv : m) set_attr_hw(*kv.second); // key order, like std::map
    }
};

struct MappedNeuron;
struct MappedConnection;

// --------------------------- built-in models -------------------------------
// src/models.cpp:29-68
struct CurrentBasedSynapse : Unit
{
    std::vector<double> weights;
    CurrentBasedSynapse() { impl_syn = true; }
    Result update_syn(size_t addr, bool read, long) override
    {
        Result r;
        r.current = read ? Opt(weights.at(addr)) : Opt(0.0);
        return r;
    }
    void set_attr_edge(size_t addr, const Attr &a) override
    {
        if (weights.size() <= addr) weights.resize(std::max(weights.size() * 2, addr + 1));
        if (a.key == "w" || a.key == "weight") weights.at(addr) = a.as_double();
    }
};

constexpr size_t LOIHI_MAX_CX = 1024; // src/models.hpp:29

// src/models.cpp:71-94, src/models.hpp:58-102
struct Accumulator : Unit
{
    std::vector<Opt> q;
    std::vector<long> last_ts;
    Accumulator() : q(LOIHI_MAX_CX), last_ts(LOIHI_MAX_CX, 0) { impl_dend = true; }
    Result update_dend(size_t n, Opt current, bool, size_t, long t) override
    {
        if (last_ts.at(n) < t)
        {
            q.at(n) = Opt(0.0);
            last_ts.at(n) = t;
        }
        if (current.has) q.at(n) = Opt(q.at(n).value_or(0.0) + current.v);
        Result r;
        r.current = q.at(n);
        return r;
    }
    void reset() override { q.assign(LOIHI_MAX_CX, Opt()); }
};

// src/models.cpp:96-165, src/models.hpp:104-163
struct AccumulatorWithDelay : Unit
{
    static constexpr size_t max_delay = 5;
    std::vector<Opt> acc;
    std::vector<std::vector<Opt>> next;
    std::vector<long> ts_sim;
    std::vector<size_t> delays;
    AccumulatorWithDelay()
            : acc(LOIHI_MAX_CX), next(max_delay + 1, std::vector<Opt>(LOIHI_MAX_CX)), ts_sim(LOIHI_MAX_CX, 0)
    {
        impl_dend = true;
    }
    Result update_dend(size_t n, Opt current, bool has_syn, size_t syn, long t) override
    {
        while (ts_sim[n] < t)
        {
            ++ts_sim[n];
            acc[n] = next[0][n];
            for (size_t i = 0; i + 1 < next.size(); i++) next[i][n] = next[i + 1][n];
            next[next.size() - 1][n] = Opt();
        }
        if (current.has)
        {
            const size_t s = has_syn ? syn : 0;
            const size_t d = (s < delays.size()) ? delays[s] : 0;
            next[d][n] = Opt(next[d][n].value_or(0.0) + current.v);
        }
        Result r;
        r.current = acc[n];
        return r;
    }
    void set_attr_edge(size_t addr, const Attr &a) override
    {
        if (delays.size() <= addr) delays.resize(addr + 1, 0);
        if (a.key == "delay" || a.key == "d")
        {
            const int d = a.as_int();
            if (static_cast<size_t>(d) > max_delay) throw std::runtime_error("Error: delay > max delay\n");
            delays[addr] = static_cast<size_t>(d);
        }
    }
    void reset() override
    {
        acc.assign(LOIHI_MAX_CX, Opt());
        for (auto &v : next) v.assign(LOIHI_MAX_CX, Opt());
    }
};

// src/models.cpp:167-348: one dendrite per unit instance
struct MultiTap : Unit
{
    std::vector<double> v{0.0}, nv{0.0}, space, tc{0.0};
    std::vector<int> syn_to_tap;
    long ts_sim{0};
    MultiTap() { impl_dend = true; }
    void next_state()
    {
        const size_t taps = v.size();
        for (size_t t = 0; t < taps; t++) nv[t] = v[t] * tc[t];
        for (size_t s = 0; s < taps; s++)
        {
            if (s > 0)
            {
                const double c = v[s] * space[s - 1];
                nv[s - 1] += c;
                nv[s] -= c;
            }
            if (s < taps - 1)
            {
                const double c = v[s] * space[s];
                nv[s + 1] += c;
                nv[s] -= c;
            }
        }
        for (size_t t = 0; t < taps; t++) v[t] = nv[t];
    }
    Result update_dend(size_t, Opt current, bool has_syn, size_t syn, long t) override
    {
        while (ts_sim < t)
        {
            ++ts_sim;
            next_state();
        }
        if (current.has)
        {
            int tap = 0;
            if (has_syn && syn < syn_to_tap.size()) tap = syn_to_tap[syn];
            if (tap < 0 || static_cast<size_t>(tap) >= v.size())
                throw std::logic_error("Tap should be >= 0 and less than taps.\n");
            v[tap] += current.v;
        }
        Result r;
        r.current = Opt(v[0]);
        return r;
    }
    void set_attr_neuron(size_t, const Attr &a) override
    {
        if (a.key == "taps")
        {
            const size_t n = a.as_int();
            if (n == 0) throw std::invalid_argument("Number of taps must be > 0\n");
            v.resize(n);
            nv.resize(n);
            tc.resize(n);
            space.resize(n - 1);
        }
        else if (a.key == "time_constants")
        {
            const size_t n = v.size();
            tc = a.list;
            if (tc.size() < n) throw std::invalid_argument("too few time constants");
        }
        else if (a.key == "space_constants")
        {
            const size_t n = v.size();
            space = a.list;
            if (space.size() < n - 1) throw std::invalid_argument("too few space constants");
        }
    }
    void set_attr_edge(size_t addr, const Attr &a) override
    {
        if (a.key == "tap")
        {
            if (syn_to_tap.size() <= addr) syn_to_tap.resize(addr + 1, 0);
            syn_to_tap[addr] = a.as_int();
        }
    }
    void reset() override
    {
        for (size_t i = 0; i < v.size(); i++) v[i] = nv[i] = 0.0;
    }
};

// src/models.cpp:351-662, src/models.hpp:200-282
struct LoihiLif : Unit
{
    struct Cx
    {
        double bias{0.0};
        bool force_update{false};
        double input_current{0.0}, input_decay{0.0}, leak_decay{1.0};
        bool log_current{false};
        double potential{0.0};
        int refractory_delay{0}, refractory_count{0};
        double reset{0.0};
        int reset_mode{RESET_HARD};
        double reverse_reset{0.0};
        int reverse_reset_mode{RESET_NONE};
        double reverse_threshold{0.0}, threshold{0.0};
        long ts_sim{0};
    };
    std::vector<Cx> cx;
    bool noise_file{false};
    std::ifstream noise_stream;
    long sign_mask{0x100}, random_mask{0x7f};
    LoihiLif() : cx(LOIHI_MAX_CX) { impl_soma = true; }
    void set_attr_hw(const Attr &a) override
    {
        if (a.key == "noise")
        {
            noise_file = true;
            noise_stream.open(a.as_string());
            if (!noise_stream.is_open()) throw std::runtime_error("Failed to open noise stream");
        }
        else if (a.key == "noise_bits")
        {
            random_mask = (1L << a.as_int()) - 1L;
        }
    }
    void set_attr_neuron(size_t n, const Attr &a) override
    {
        Cx &c = cx.at(n);
        const std::string &k = a.key;
        if (k == "threshold") c.threshold = a.as_double();
        else if (k == "reverse_threshold") c.reverse_threshold = a.as_double();
        else if (k == "reset") c.reset = a.as_double();
        else if (k == "reverse_reset") c.reverse_reset = a.as_double();
        else if (k == "reset_mode") c.reset_mode = parse_reset_mode(a.as_string());
        else if (k == "reverse_reset_mode") c.reverse_reset_mode = parse_reset_mode(a.as_string());
        else if (k == "leak_decay") c.leak_decay = a.as_double();
        else if (k == "log_u") c.log_current = a.as_bool();
        else if (k == "input_decay") c.input_decay = a.as_double();
        else if (k == "bias") c.bias = a.as_double();
        else if (k == "force_update" || k == "force_update_every_timestep") c.force_update = a.as_bool();
        else if (k == "refractory_delay") c.refractory_delay = a.as_int();
        else if (k == "potential") c.potential = a.as_double();
    }
    double noise() // src/models.cpp:589-651
    {
        if (!noise_stream.is_open()) throw std::runtime_error("Noise stream is not open");
        if (noise_stream.eof() || noise_stream.peek() == std::ifstream::traits_type::eof())
        {
            noise_stream.clear();
            noise_stream.seekg(0, std::ios::beg);
        }
        std::string line;
        int rv = 0;
        if (std::getline(noise_stream, line))
        {
            std::istringstream iss(line);
            iss >> rv;
        }
        else
        {
            throw std::runtime_error("Couldn't read noise entry from file");
        }
        long r = rv;
        const long sign = r & sign_mask;
        r &= random_mask;
        if (sign != 0) r |= ~random_mask;
        return static_cast<double>(r);
    }
    Result update_soma(size_t n, Opt in, long t) override // src/models.cpp:497-567
    {
        Cx &c = cx[n];
        if (c.ts_sim == t)
            throw std::runtime_error("This model does not support multiple updates to the same compartment in one time-step.");
        if (c.ts_sim < t - 1) throw std::runtime_error("This model must update every time-step.\n");
        Status st = IDLE;
        if (std::fabs(c.potential) > 0.0 || in.has || std::fabs(c.bias) > 0.0 || c.force_update) st = UPDATED;
        if (c.ts_sim > 0)
        {
            c.input_current *= c.input_decay;
            c.potential *= c.leak_decay;
        }
        c.potential = static_cast<int>(c.potential * 64.0) / 64.0;
        if (noise_file) c.potential += noise();
        if (!(c.refractory_count > 0))
        {
            c.potential += c.bias;
            c.input_current += in.value_or(0.0);
            c.potential += c.input_current;
            bool fired = false;
            if (c.potential > c.threshold)
            {
                if (c.reset_mode == RESET_HARD) c.potential = c.reset;
                else if (c.reset_mode == RESET_SOFT) c.potential -= c.threshold;
                c.refractory_count = c.refractory_delay;
                fired = true;
            }
            if (c.potential < c.reverse_threshold)
            {
                if (c.reverse_reset_mode == RESET_SOFT) c.potential -= c.reverse_threshold;
                else if (c.reverse_reset_mode == RESET_HARD) c.potential = c.reverse_reset;
                else if (c.reverse_reset_mode == RESET_SATURATE) c.potential = c.reverse_threshold;
            }
            if (fired) st = FIRED;
        }
        ++c.ts_sim;
        c.refractory_count = std::max(0, c.refractory_count - 1);
        Result r;
        r.status = st;
        return r;
    }
    void reset() override
    {
        for (Cx &c : cx)
        {
            c.input_current = 0.0;
            c.potential = 0.0;
        }
    }
    double get_potential(size_t n) override { return cx[n].potential; }
    std::map<std::string, double> get_traces(size_t n) override
    {
        if (cx.at(n).log_current) return {{"u", cx.at(n).input_current}};
        return {};
    }
};

constexpr size_t TRUENORTH_MAX = 4096; // src/models.hpp:284
// src/models.cpp:664-830
struct TrueNorth : Unit
{
    struct N
    {
        bool force_update{false};
        unsigned random_range_mask{0};
        int reset_mode{RESET_HARD}, reverse_reset_mode{RESET_NONE};
        bool leak_towards_zero{true};
        double potential{0}, leak{0}, bias{0}, threshold{0}, reverse_threshold{0}, reset{0}, reverse_reset{0};
    };
    std::vector<N> ns;
    TrueNorth() : ns(TRUENORTH_MAX) { impl_soma = true; }
    void set_attr_neuron(size_t i, const Attr &a) override
    {
        N &n = ns.at(i);
        const std::string &k = a.key;
        if (k == "threshold") n.threshold = a.as_double();
        else if (k == "reverse_threshold") n.reverse_threshold = a.as_double();
        else if (k == "reset") n.reset = a.as_double();
        else if (k == "reverse_reset") n.reverse_reset = a.as_double();
        else if (k == "reset_mode") n.reset_mode = parse_reset_mode(a.as_string());
        else if (k == "reverse_reset_mode") n.reverse_reset_mode = parse_reset_mode(a.as_string());
        else if (k == "leak") n.leak = a.as_double();
        else if (k == "bias") n.bias = a.as_double();
        else if (k == "force_update_every_timestep" || k == "force_update") n.force_update = a.as_bool();
        else if (k == "leak_towards_zero") n.leak_towards_zero = a.as_bool();
        else if (k == "random_mask")
        {
            const int m = a.as_int();
            if (m < 0) throw std::invalid_argument("random_mask < 0; must be unsigned.");
            n.random_range_mask = static_cast<unsigned>(m);
        }
    }
    Result update_soma(size_t i, Opt in, long) override
    {
        N &n = ns[i];
        Status st = IDLE;
        if (std::fabs(n.potential) > 0.0 || in.has || std::fabs(n.bias) > 0.0 || n.force_update) st = UPDATED;
        if (n.leak_towards_zero)
        {
            if (n.potential > 0.0) n.potential -= n.leak;
            else if (n.potential < 0.0) n.potential += n.leak;
        }
        else
        {
            n.potential += n.leak;
        }
        n.potential += n.bias;
        if (in.has) n.potential += in.v;
        double v = n.potential;
        if (n.random_range_mask != 0)
        {
            const unsigned r = std::rand() & n.random_range_mask;
            v += static_cast<double>(r);
        }
        if (v >= n.threshold)
        {
            if (n.reset_mode == RESET_HARD) n.potential = n.reset;
            else if (n.reset_mode == RESET_SOFT) n.potential -= n.threshold;
            else if (n.reset_mode == RESET_SATURATE) n.potential = n.threshold;
            st = FIRED;
        }
        else if (v <= n.reverse_threshold)
        {
            if (n.reverse_reset_mode == RESET_HARD) n.potential = n.reverse_reset;
            else if (n.reverse_reset_mode == RESET_SOFT) n.potential += n.reverse_threshold;
            else if (n.reverse_reset_mode == RESET_SATURATE) n.potential = n.reverse_threshold;
        }
        Result r;
        r.status = st;
        return r;
    }
    void reset() override
    {
        for (N &n : ns) n.potential = 0.0;
    }
    double get_potential(size_t i) override { return ns[i].potential; }
};

// src/models.cpp:832-903, src/models.hpp:344-378: one neuron per instance
struct InputModel : Unit
{
    std::vector<bool> spikes;
    size_t cur{0};
    std::uniform_real_distribution<double> uni{0.0, 1.0};
    std::mt19937 gen;
    double poisson{0.0}, rate{0.0};
    explicit InputModel(unsigned seed) : gen(seed) { impl_soma = true; }
    void set_attr_neuron(size_t, const Attr &a) override
    {
        if (a.key == "spikes")
        {
            spikes.clear();
            for (double d : a.list) spikes.push_back(d != 0.0);
            cur = 0;
        }
        else if (a.key == "poisson") poisson = a.as_double();
        else if (a.key == "rate") rate = a.as_double();
    }
    Result update_soma(size_t, Opt in, long t) override
    {
        if (in.has && in.v != 0.0)
            throw std::runtime_error("Current sent to input neuron which cannot be processed (" + std::to_string(in.v) + ")");
        bool send = false;
        if (cur < spikes.size())
        {
            send = spikes[cur];
            ++cur;
        }
        if (poisson > uni(gen)) send = true;
        if (rate > 0.0 && (t % static_cast<long>(1.0 / rate)) == 0) send = true;
        Result r;
        r.status = send ? FIRED : IDLE;
        return r;
    }
};

// plugins/hodgkin_huxley.cpp:22-170 restated (one neuron per instance)
struct HodgkinHuxley : Unit
{
    double C_m{10.0}, g_Na{1200.0}, g_K{360.0}, g_L{3.0}, V_Na{50.0}, V_K{-77.0}, V_L{54.387}, dt{0.1};
    double V{0.0}, prev_V{0.0}, I{0.0}, m{0.0}, n{0.0}, h{0.0};
    HodgkinHuxley() { impl_soma = true; }
    void set_attr_neuron(size_t, const Attr &a) override
    {
        if (a.key == "m") m = a.as_double();
        else if (a.key == "n") n = a.as_double();
        else if (a.key == "h") h = a.as_double();
        else if (a.key == "current") I = a.as_double();
    }
    Result update_soma(size_t, Opt, long) override
    {
        const double alpha_n = (0.01 * (V + 55)) / (1 - exp(-0.1 * (V + 55)));
        const double alpha_m = (0.1 * (V + 40)) / (1 - exp(-0.1 * (V + 40)));
        const double alpha_h = 0.07 * exp(-0.05 * (V + 65));
        const double beta_n = 0.125 * exp(-0.01125 * (V + 55));
        const double beta_m = 4 * exp(-0.05556 * (V + 65));
        const double beta_h = 1 / (1 + exp(-0.1 * (V + 35)));
        const double tau_n = 1 / (alpha_n + beta_n);
        const double tau_m = 1 / (alpha_m + beta_m);
        const double tau_h = 1 / (alpha_h + beta_h);
        const double pm = alpha_m / (alpha_m + beta_m);
        const double pn = alpha_n / (alpha_n + beta_n);
        const double ph = alpha_h / (alpha_h + beta_h);
        const double den = g_L + g_K * (pow(n, 4)) + g_Na * (pow(m, 3) * h);
        const double tau_V = C_m / den;
        const double Vinf = ((g_L) *V_L + g_K * (pow(n, 4)) * V_K + g_Na * (pow(m, 3)) * h * V_Na + I) / den;
        prev_V = V;
        V = Vinf + (V - Vinf) * exp(-1 * dt / tau_V);
        m = pm + (m - pm) * exp(-1 * dt / tau_m);
        n = pn + (n - pn) * exp(-1 * dt / tau_n);
        h = ph + (h - ph) * exp(-1 * dt / tau_h);
        Result r;
        r.status = ((prev_V < 25) && (V > 25)) ? FIRED : UPDATED;
        return r;
    }
    void reset() override { prev_V = V = m = n = h = 0.0; }
    double get_potential(size_t) override { return V; }
};

// ---------------------------------------------------------------------------
// Mapped structures: src/mapped.hpp, src/core.hpp, src/tile.hpp, src/message.hpp
// ---------------------------------------------------------------------------
struct Core;

struct MappedConnection
{
    MappedNeuron *pre{nullptr}, *post{nullptr};
    Unit *synapse_hw{nullptr};
    std::vector<Unit *> pipeline;
    size_t connection_offset{0}, syn_addr{0};
};

struct MappedNeuron
{
    std::vector<MappedConnection> connections_out;
    std::vector<size_t> axon_out_addresses;
    int64_t gid{0}; // global neuron id in desc order
    int group{0};
    size_t offset{0}, id{0};
    Core *core{nullptr};
    Unit *dendrite_hw{nullptr}, *soma_hw{nullptr};
    std::vector<Unit *> pipeline;
    size_t offset_in_core{0}, dend_addr{0}, soma_addr{0};
    Status status{UNSET};
    bool log_spikes{false}, log_potential{false};
    bool check_synapse_updates{false};
};

struct AxonIn { std::vector<size_t> synapse_addresses; };
struct AxonOut { size_t dest_axon_id{0}, dest_tile_id{0}, dest_core_offset{0}, src_neuron_offset{0}; };
struct AxonInUnit { long spike_messages_in{0}; double energy_msg{0}, latency_msg{0}; };
struct AxonOutUnit { long packets_out{0}; double energy{0}, energy_access{0}, latency_access{0}; };

struct Message : oracle_msg
{
    bool in_noc{false};
};

struct Core
{
    std::vector<AxonInUnit> axon_in_hw;
    std::vector<std::unique_ptr<Unit>> pipeline_hw;
    std::vector<AxonOutUnit> axon_out_hw;
    std::vector<Unit *> in_use;
    std::vector<Message *> messages_in;
    std::vector<AxonIn> axons_in;
    std::vector<MappedNeuron> neurons;
    std::vector<MappedConnection *> connections_in;
    std::vector<AxonOut> axons_out;
    std::vector<Result> buffer;
    int buffer_pos{SANAFE_BUF_BEFORE_SOMA};
    size_t max_neurons{1024};
    double energy{0.0}, next_delay{0.0};
    size_t id{0}, offset{0}, tile{0};
    int tmpl{0};               // core template (include/sanafe_desc.h)
    unsigned input_seed_base{0}; // InputModel instances created before this core's units
};

struct Tile
{
    std::vector<size_t> cores; // global core ids
    double e_hop[4]{}, l_hop[4]{};
    double energy{0};
    size_t hops{0}, north{0}, east{0}, south{0}, west{0};
    long messages_received{0};
    size_t id{0}, x{0}, y{0};
};

struct Timestep
{
    std::vector<std::list<Message>> messages;
    oracle_ts t{};
};

size_t abs_diff(size_t a, size_t b) { return a > b ? a - b : b - a; }

} // namespace

// ---------------------------------------------------------------------------
// The chip: src/chip.cpp
// ---------------------------------------------------------------------------
struct oracle_chip
{
    std::vector<Tile> tiles;
    std::vector<Core> cores;
    std::vector<std::string> strings;
    std::map<size_t, double> sync_table;
    size_t noc_w{1}, noc_h{1}, noc_buf{0}, max_cores_per_tile{0};
    std::vector<std::string> group_names;
    std::vector<int64_t> group_ptr;
    std::vector<MappedNeuron *> by_gid;
    size_t mapped_tiles{0}, mapped_cores{0};
    long total_timesteps{0};
    std::atomic<long> total_messages_sent{0}; // src/chip.hpp: std::atomic<long>, fetch_add per message (src/chip.cpp:815)
    int omp_threads{1}; // > 1: OpenMP over cores like the reference (src/chip.cpp:629-632, 675-678, 1398-1401); message
                        // ids then depend on the thread schedule, exactly as they do there
    unsigned input_instances{0};
    const sanafe_desc *desc{nullptr}; // borrowed: the caller keeps the desc alive with the chip
    Timestep last;

    const std::string &str(int32_t id) const
    {
        static const std::string empty;
        return id < 0 ? empty : strings.at(id);
    }

    static std::vector<Attr> read_attrs(const oracle_chip &c, const sanafe_attr_table &t, int64_t b, int64_t e)
    {
        std::vector<Attr> v;
        for (int64_t i = b; i < e; i++)
        {
            Attr a;
            a.key = c.str(t.key[i]);
            a.type = t.type[i];
            a.fwd = t.fwd ? t.fwd[i] : 7;
            a.num = t.num[i];
            if (a.type == SANAFE_ATTR_STRING) a.str = c.str(t.str[i]);
            if (a.type == SANAFE_ATTR_LIST && t.list_ptr)
                a.list.assign(t.list_num + t.list_ptr[i], t.list_num + t.list_ptr[i + 1]);
            v.push_back(std::move(a));
        }
        return v;
    }

    // src/models.cpp:933-967 (+ the HH plugin by model name)
    std::unique_ptr<Unit> make_unit(const std::string &model, unsigned input_seed = 0)
    {
        if (model == "current_based") return std::make_unique<CurrentBasedSynapse>();
        if (model == "accumulator") return std::make_unique<Accumulator>();
        if (model == "accumulator_with_delay") return std::make_unique<AccumulatorWithDelay>();
        if (model == "taps") return std::make_unique<MultiTap>();
        if (model == "input") return std::make_unique<InputModel>(input_seed ? input_seed : ++input_instances);
        if (model == "leaky_integrate_fire") return std::make_unique<LoihiLif>();
        if (model == "truenorth") return std::make_unique<TrueNorth>();
        if (model == "hodgkin_huxley") return std::make_unique<HodgkinHuxley>();
        throw std::invalid_argument("Pipeline model not supported (" + model + ")\n");
    }

    // SpikingChip::SpikingChip, src/chip.cpp:61-104; Core::create_pipeline_unit src/core.cpp:196-231
    void build_arch(const sanafe_desc &d)
    {
        desc = &d;
        noc_w = d.noc_width;
        noc_h = d.noc_height;
        noc_buf = d.noc_buffer_size;
        for (int i = 0; i < d.n_sync; i++) sync_table[static_cast<size_t>(d.sync_key[i])] = d.sync_val[i];
        tiles.resize(d.n_tiles);
        for (int t = 0; t < d.n_tiles; t++)
        {
            Tile &tile = tiles[t];
            tile.id = t;
            tile.x = t / noc_h; // src/arch.cpp:78-88
            tile.y = t % noc_h;
            for (int k = 0; k < 4; k++)
            {
                tile.e_hop[k] = d.tile_hop_energy[t * 4 + k];
                tile.l_hop[k] = d.tile_hop_latency[t * 4 + k];
            }
        }
        cores.resize(d.n_cores);
        for (int c = 0; c < d.n_cores; c++)
        {
            Core &core = cores[c];
            core.id = c;
            core.tile = d.core_tile[c];
            core.offset = tiles[core.tile].cores.size();
            tiles[core.tile].cores.push_back(c);
            core.buffer_pos = d.core_buffer_pos[c];
            core.max_neurons = d.core_max_neurons[c];
            core.buffer.resize(core.max_neurons);
            core.tmpl = d.core_template[c];
            const int tm = core.tmpl;
            for (int i = d.tmpl_axon_in_ptr[tm]; i < d.tmpl_axon_in_ptr[tm + 1]; i++)
            {
                AxonInUnit u;
                u.energy_msg = d.axon_in_energy[i];
                u.latency_msg = d.axon_in_latency[i];
                core.axon_in_hw.push_back(u);
            }
            // The reference instantiates every unit of every core up front
            // (src/chip.cpp:83-87).  A loihi_large chip has 4096 x 1031 units, so the
            // oracle instantiates a unit on first use instead; the only creation-order
            // side effect, InputModel's seed counter (src/models.hpp:347, 366), is
            // reproduced arithmetically through input_seed_base.
            core.input_seed_base = input_instances;
            core.pipeline_hw.resize(d.tmpl_unit_ptr[tm + 1] - d.tmpl_unit_ptr[tm]);
            for (int u = d.tmpl_unit_ptr[tm]; u < d.tmpl_unit_ptr[tm + 1]; u++)
                if (str(d.unit_model[u]) == "input") ++input_instances;
            for (int i = d.tmpl_axon_out_ptr[tm]; i < d.tmpl_axon_out_ptr[tm + 1]; i++)
            {
                AxonOutUnit u;
                u.energy_access = d.axon_out_energy[i];
                u.latency_access = d.axon_out_latency[i];
                core.axon_out_hw.push_back(u);
            }
        }
        for (const Tile &t : tiles) max_cores_per_tile = std::max(max_cores_per_tile, t.cores.size());
    }

    // Core::create_pipeline_unit, src/core.cpp:196-231 (performed on first use, see build_arch)
    Unit *instantiate(Core &core, int slot)
    {
        if (core.pipeline_hw[slot]) return core.pipeline_hw[slot].get();
        const sanafe_desc &d = *desc;
        const int u = d.tmpl_unit_ptr[core.tmpl] + slot;
        unsigned seed = 0;
        if (str(d.unit_model[u]) == "input")
        {
            seed = core.input_seed_base + 1;
            for (int k = d.tmpl_unit_ptr[core.tmpl]; k < u; k++)
                if (str(d.unit_model[k]) == "input") ++seed;
        }
        std::unique_ptr<Unit> unit = make_unit(str(d.unit_model[u]), seed);
        unit->name = str(d.unit_name[u]);
        unit->model = str(d.unit_model[u]);
        unit->log_energy = d.unit_flags[u] & SANAFE_UNIT_LOG_ENERGY;
        unit->log_latency = d.unit_flags[u] & SANAFE_UNIT_LOG_LATENCY;
        unit->update_every_timestep = d.unit_flags[u] & SANAFE_UNIT_UPDATE_EVERY_TIMESTEP;
        unit->set_attributes_hw(read_attrs(*this, d.unit_attrs, d.unit_attr_ptr[u], d.unit_attr_ptr[u + 1]));
        const bool sy = d.unit_implements[u] & SANAFE_IMPL_SYNAPSE;
        const bool de = d.unit_implements[u] & SANAFE_IMPL_DENDRITE;
        const bool so = d.unit_implements[u] & SANAFE_IMPL_SOMA;
        if (sy != unit->impl_syn || de != unit->impl_dend || so != unit->impl_soma)
            throw std::runtime_error("Unit '" + unit->name + "' is listed in a section it does not implement");
        core.pipeline_hw[slot] = std::move(unit);
        return core.pipeline_hw[slot].get();
    }

    // Core::get_hw, src/core.cpp:61-97
    Unit *get_hw(Core &core, const std::string &name, bool syn, bool dend, bool soma)
    {
        const sanafe_desc &d = *desc;
        const int b = d.tmpl_unit_ptr[core.tmpl], e = d.tmpl_unit_ptr[core.tmpl + 1];
        for (int u = b; u < e; u++)
        {
            const int impl = d.unit_implements[u];
            if ((syn && !(impl & SANAFE_IMPL_SYNAPSE)) || (dend && !(impl & SANAFE_IMPL_DENDRITE)) ||
                    (soma && !(impl & SANAFE_IMPL_SOMA)))
                continue;
            if (name.empty() || name == str(d.unit_name[u])) return instantiate(core, u - b);
        }
        throw std::runtime_error("Could not find h/w (with name:" + name + ")");
    }

    // MappedNeuron::set_attributes, src/mapped.cpp:113-166
    static void neuron_set_attr(MappedNeuron &n, const Attr &a)
    {
        static const std::set<std::string> reserved = {"soma_hw_name", "default_synapse_hw_name", "dendrite_hw_name",
                "log_spikes", "log_potential", "log_v"};
        if (reserved.count(a.key))
            throw std::invalid_argument("Reserved neuron attribute '" + a.key + "' cannot be used as a model attribute.");
        if ((a.fwd & SANAFE_FWD_DENDRITE) && n.dendrite_hw) n.dendrite_hw->set_attr_neuron(n.dend_addr, a);
        if ((a.fwd & SANAFE_FWD_SOMA) && n.soma_hw) n.soma_hw->set_attr_neuron(n.soma_addr, a);
    }

    // SpikingChip::load, src/chip.cpp:129-408
    void load(const sanafe_desc &d)
    {
        group_ptr.assign(d.group_ptr, d.group_ptr + d.n_groups + 1);
        for (int g = 0; g < d.n_groups; g++) group_names.push_back(str(d.group_name[g]));
        std::vector<int> group_lex(d.n_groups); // std::map<std::string,...> order
        for (int g = 0; g < d.n_groups; g++) group_lex[g] = g;
        std::sort(group_lex.begin(), group_lex.end(), [&](int a, int b) { return group_names[a] < group_names[b]; });
        std::vector<int> group_of(d.n_neurons);
        for (int g = 0; g < d.n_groups; g++)
            for (int64_t n = group_ptr[g]; n < group_ptr[g + 1]; n++) group_of[n] = g;

        // map_neurons: src/chip.cpp:186-234
        std::vector<int64_t> order;
        order.reserve(d.n_neurons);
        for (int g : group_lex)
            for (int64_t n = group_ptr[g]; n < group_ptr[g + 1]; n++) order.push_back(n);
        std::stable_sort(order.begin(), order.end(),
                [&](int64_t a, int64_t b) { return d.neuron_map_order[a] < d.neuron_map_order[b]; });
        // Count per-core populations first so the neuron vectors never reallocate
        std::vector<size_t> per_core(cores.size(), 0);
        for (int64_t n : order)
        {
            if (d.neuron_core[n] < 0)
                throw std::runtime_error("Neuron: " + group_names[group_of[n]] + "." +
                        std::to_string(n - group_ptr[group_of[n]]) + " not mapped.");
            per_core.at(d.neuron_core[n])++;
        }
        for (size_t c = 0; c < cores.size(); c++) cores[c].neurons.reserve(per_core[c]);
        by_gid.assign(d.n_neurons, nullptr);
        size_t total_mapped = 0;
        for (int64_t gid : order)
        {
            Core &core = cores.at(d.neuron_core[gid]);
            // Core::map_neuron, src/core.cpp:116-168
            if (core.neurons.size() >= core.max_neurons)
                throw std::runtime_error("Error: Exceeded maximum neurons per core.");
            if (core.pipeline_hw.empty()) throw std::runtime_error("Error: No units defined");
            Unit *dend = get_hw(core, str(d.neuron_dendrite_hw[gid]), false, true, false);
            Unit *soma = get_hw(core, str(d.neuron_soma_hw[gid]), false, false, true);
            if (core.axon_out_hw.empty()) throw std::runtime_error("Error: No axon out units defined");
            core.neurons.emplace_back();
            MappedNeuron &mn = core.neurons.back();
            mn.gid = gid;
            mn.group = group_of[gid];
            mn.offset = gid - group_ptr[mn.group];
            mn.id = total_mapped++;
            mn.core = &core;
            mn.dendrite_hw = dend;
            mn.soma_hw = soma;
            mn.offset_in_core = core.neurons.size() - 1;
            mn.log_spikes = d.neuron_log_spikes[gid];
            mn.log_potential = d.neuron_log_potential[gid];
            // build_neuron_processing_pipeline, src/mapped.cpp:168-188
            bool dend_added = false;
            if (core.buffer_pos <= SANAFE_BUF_INSIDE_DENDRITE)
            {
                mn.pipeline.push_back(dend);
                dend_added = true;
            }
            if (core.buffer_pos <= SANAFE_BUF_INSIDE_SOMA)
                if (soma != dend || !dend_added) mn.pipeline.push_back(soma);
            dend->is_used = true;
            mn.dend_addr = dend->neuron_count++;
            if (soma != dend)
            {
                soma->is_used = true;
                mn.soma_addr = soma->neuron_count++;
            }
            else
            {
                mn.soma_addr = mn.dend_addr;
            }
            for (const Attr &a : read_attrs(*this, d.neuron_attrs, d.neuron_attr_ptr[gid], d.neuron_attr_ptr[gid + 1]))
                neuron_set_attr(mn, a);
            by_gid[gid] = &mn;
        }
        // track_mapped_tiles_and_cores, src/chip.cpp:283-306
        mapped_tiles = mapped_cores = 0;
        for (const Tile &t : tiles)
        {
            bool used = false;
            for (size_t c : t.cores)
                if (!cores[c].neurons.empty())
                {
                    used = true;
                    ++mapped_cores;
                }
            if (used) ++mapped_tiles;
        }

        // map_connections, src/chip.cpp:334-380: groups in lexicographic order,
        // neurons by offset, edges in creation order.
        std::vector<int64_t> eorder(d.n_edges);
        for (int64_t e = 0; e < d.n_edges; e++) eorder[e] = e;
        std::vector<int> lex_rank(d.n_groups);
        for (int i = 0; i < d.n_groups; i++) lex_rank[group_lex[i]] = i;
        std::stable_sort(eorder.begin(), eorder.end(), [&](int64_t a, int64_t b) {
            const int64_t sa = d.edge_src[a], sb = d.edge_src[b];
            const int ra = lex_rank[group_of[sa]], rb = lex_rank[group_of[sb]];
            if (ra != rb) return ra < rb;
            return sa < sb;
        });
        std::vector<size_t> out_count(d.n_neurons, 0);
        for (int64_t e = 0; e < d.n_edges; e++) out_count[d.edge_src[e]]++;
        for (int64_t n = 0; n < d.n_neurons; n++)
            if (by_gid[n]) by_gid[n]->connections_out.reserve(out_count[n]);
        for (int64_t e : eorder)
        {
            MappedNeuron &pre = *by_gid.at(d.edge_src[e]);
            MappedNeuron &post = *by_gid.at(d.edge_dst[e]);
            Core &post_core = *post.core;
            // get_synapse_hw_name, src/chip.cpp:308-332
            std::string hw_name = str(d.edge_synapse_hw[e]);
            if (hw_name.empty()) hw_name = str(d.neuron_synapse_hw[d.edge_dst[e]]);
            // Core::map_connection, src/core.cpp:170-184
            pre.connections_out.emplace_back();
            MappedConnection &con = pre.connections_out.back();
            con.pre = &pre;
            con.post = &post;
            con.synapse_hw = get_hw(post_core, hw_name, true, false, false);
            con.syn_addr = con.synapse_hw->connection_count++;
            con.synapse_hw->is_used = true;
            // build_message_processing_pipeline, src/mapped.cpp:27-58
            post.check_synapse_updates |= con.synapse_hw->update_every_timestep;
            con.pipeline.push_back(con.synapse_hw);
            if (post_core.buffer_pos > SANAFE_BUF_BEFORE_DENDRITE && post.dendrite_hw != con.synapse_hw)
                con.pipeline.push_back(post.dendrite_hw);
            if (post_core.buffer_pos > SANAFE_BUF_BEFORE_SOMA && post.soma_hw != post.dendrite_hw)
                con.pipeline.push_back(post.soma_hw);
            // MappedConnection::set_attributes, src/mapped.cpp:60-89 (synapse_attributes only)
            std::vector<Attr> attrs;
            {
                Attr w;
                w.key = "weight";
                w.type = SANAFE_ATTR_DOUBLE;
                w.num = d.edge_weight[e];
                attrs.push_back(w);
            }
            if (d.edge_delay && d.edge_delay[e] >= 0)
            {
                Attr dl;
                dl.key = d.edge_delay[e] >= 64 ? "tap" : "delay"; // include/sanafe_desc.h: 64 + tap index
                dl.type = SANAFE_ATTR_INT;
                dl.num = d.edge_delay[e] >= 64 ? d.edge_delay[e] - 64 : d.edge_delay[e];
                attrs.push_back(dl);
            }
            if (d.edge_attr_ptr)
                for (Attr &a : read_attrs(*this, d.edge_attrs, d.edge_attr_ptr[e], d.edge_attr_ptr[e + 1]))
                    attrs.push_back(a);
            for (const Attr &a : attrs)
            {
                if (a.fwd & SANAFE_FWD_SYNAPSE) con.synapse_hw->set_attr_edge(con.syn_addr, a);
                if (a.fwd & SANAFE_FWD_DENDRITE) post.dendrite_hw->set_attr_edge(con.syn_addr, a);
            }
        }
        // map_axons, src/chip.cpp:382-408, 1263-1391
        for (Tile &t : tiles)
            for (size_t c : t.cores)
                for (MappedNeuron &pre : cores[c].neurons)
                {
                    std::set<size_t> cores_out; // std::set<Core*>: ascending core id (quirk 11)
                    for (const MappedConnection &con : pre.connections_out) cores_out.insert(con.post->core->id);
                    for (size_t dc : cores_out)
                    {
                        Core &post_core = cores[dc];
                        post_core.axons_in.emplace_back();
                        AxonOut out;
                        out.dest_axon_id = post_core.axons_in.size() - 1;
                        out.dest_core_offset = post_core.offset;
                        out.dest_tile_id = post_core.tile;
                        out.src_neuron_offset = pre.offset;
                        pre.core->axons_out.push_back(out);
                        pre.axon_out_addresses.push_back(pre.core->axons_out.size() - 1);
                    }
                    for (MappedConnection &con : pre.connections_out)
                    {
                        Core &post_core = *con.post->core;
                        post_core.connections_in.push_back(&con);
                        con.connection_offset = post_core.connections_in.size() - 1;
                        post_core.axons_in.back().synapse_addresses.push_back(con.connection_offset);
                    }
                }
        // NOTE: the reference appends to `post_core.axons_in.back()`, i.e. the newest
        // axon at that core.  Because a pre-neuron first allocates one axon per
        // destination core and then adds its connections, `.back()` is this
        // pre-neuron's axon at that core.  The loop above follows the same rule.
        for (Core &c : cores)
        {
            c.in_use.clear();
            for (auto &hw : c.pipeline_hw)
                if (hw && hw->is_used) c.in_use.push_back(hw.get());
        }
    }

    // ---- default costing: src/pipeline.hpp:511-731 ----
    static void synapse_costs(const MappedConnection &con, Result &r)
    {
        const Unit &u = *con.synapse_hw;
        if (r.energy.has && u.def_energy_process_spike.has)
            throw std::runtime_error("Synapse unit simulates energy and also has default energy metrics set.");
        if (u.def_energy_process_spike.has) r.energy = u.def_energy_process_spike;
        if (r.latency.has && u.def_latency_process_spike.has)
            throw std::runtime_error("Synapse unit simulates latency and also has default latency metrics set.");
        if (u.def_latency_process_spike.has) r.latency = u.def_latency_process_spike;
        if (!r.energy.has) throw std::runtime_error("Synapse unit does not simulate energy or provide a default energy cost");
        if (!r.latency.has) throw std::runtime_error("Synapse unit does not simulate latency or provide a default latency cost");
    }
    static void dendrite_costs(const MappedNeuron &n, Result &r)
    {
        const Unit &u = *n.dendrite_hw;
        if (r.energy.has && u.def_energy_update.has)
            throw std::runtime_error("Dendrite unit simulates energy and also has default energy metrics set.");
        if (u.def_energy_update.has) r.energy = u.def_energy_update;
        if (r.latency.has && u.def_latency_update.has)
            throw std::runtime_error("Dendrite unit simulates latency and also has default latency metrics set.");
        if (u.def_latency_update.has) r.latency = u.def_latency_update;
        if (!r.energy.has) throw std::runtime_error("Dendrite unit does not simulate energy or provide a default energy cost");
        if (!r.latency.has) throw std::runtime_error("Dendrite unit does not simulate latency or provide a default latency cost");
    }
    static void soma_costs(MappedNeuron &n, Result &r)
    {
        Unit &u = *n.soma_hw;
        if (r.energy.has && u.has_soma_energy)
            throw std::runtime_error("Error: Soma unit simulates energy and also has default energy metrics set.");
        if (u.has_soma_energy) r.energy = Opt(u.e_access);
        if (r.latency.has && u.has_soma_latency)
            throw std::runtime_error("Error: Soma unit simulates latency and also has default latency costs set.");
        if (u.has_soma_latency) r.latency = Opt(u.l_access);
        if (r.status == UPDATED || r.status == FIRED)
        {
            if (u.has_soma_energy) r.energy.v += u.e_update;
            if (u.has_soma_latency) r.latency.v += u.l_update;
        }
        if (r.status == FIRED)
        {
            if (u.has_soma_energy) r.energy.v += u.e_spike;
            if (u.has_soma_latency) r.latency.v += u.l_spike;
        }
        if (!r.energy.has) throw std::runtime_error("Soma unit does not simulate energy or provide default energy costs");
        if (!r.latency.has) throw std::runtime_error("Soma unit does not simulate latency or provide default latency costs");
        if (r.status == UPDATED || r.status == FIRED)
        {
            u.neurons_updated++;
            if (r.status == FIRED) u.neurons_fired++;
        }
    }

    // PipelineUnit::process, src/pipeline.cpp:87-105 + adapters src/pipeline.hpp:440-508
    static Result process(Unit &u, long t, MappedNeuron &n, MappedConnection *con, const Result &in)
    {
        Result out;
        if (u.impl_syn)
        {
            const bool read = (con != nullptr);
            out = u.update_syn(read ? con->syn_addr : 0, read, t);
            ++u.spikes_processed;
        }
        else if (u.impl_dend)
        {
            out = u.update_dend(n.dend_addr, in.current, con != nullptr, con ? con->syn_addr : 0, t);
        }
        else
        {
            out = u.update_soma(n.soma_addr, in.current, t);
        }
        if (u.impl_soma) soma_costs(n, out);
        else if (u.impl_dend) dendrite_costs(n, out);
        else synapse_costs(*con, out);
        u.energy += out.energy.value_or(0.0);
        u.latency += out.energy.value_or(0.0); // sic: src/pipeline.cpp:102
        return out;
    }

    // execute_pipeline, src/chip.cpp:766-789
    static Result execute(const std::vector<Unit *> &pipe, long t, MappedNeuron &n, MappedConnection *con, const Result &in)
    {
        double te = 0.0, tl = 0.0;
        Result out = in;
        for (Unit *u : pipe)
        {
            out = process(*u, t, n, con, out);
            te += out.energy.value_or(0.0);
            tl += out.latency.value_or(0.0);
            if (out.status != UNSET) n.status = out.status;
        }
        out.energy = Opt(te);
        out.latency = Opt(tl);
        return out;
    }

    Message make_message(long id, const MappedNeuron &n, long t) // src/message.cpp:39-59
    {
        Message m{};
        m.timestep = t;
        m.mid = id;
        m.src_neuron = n.gid;
        const Core &sc = *n.core;
        const Tile &st = tiles[sc.tile];
        m.src_x = st.x;
        m.src_y = st.y;
        m.src_tile = st.id;
        m.src_core_id = sc.id;
        m.src_core_offset = sc.offset;
        m.placeholder = 1;
        m.sent_timestamp = m.received_timestamp = m.processed_timestamp = NEG_INF;
        return m;
    }

    void reset_measurements() // src/chip.cpp:1393-1445
    {
        for (Tile &t : tiles)
        {
            t.energy = 0;
            t.hops = t.north = t.east = t.south = t.west = 0;
            t.messages_received = 0;
        }
#pragma omp parallel for schedule(dynamic) if (omp_threads > 1) num_threads(omp_threads)
        for (size_t ci = 0; ci < cores.size(); ci++)
        {
            Core &c = cores[ci];
            c.energy = 0.0;
            c.next_delay = 0.0;
            for (auto &a : c.axon_in_hw) a.spike_messages_in = 0;
            for (Unit *u : c.in_use)
            {
                u->energy = u->latency = 0.0;
                u->spikes_processed = u->neurons_updated = u->neurons_fired = 0;
            }
            for (auto &a : c.axon_out_hw)
            {
                a.energy = 0.0;
                a.packets_out = 0;
            }
            c.messages_in.clear();
        }
    }

    void process_neurons(Timestep &ts) // src/chip.cpp:624-654, 710-736, 802-834
    {
        const long t = ts.t.timestep;
        std::exception_ptr failure; // an exception must not leave an OpenMP region
#pragma omp parallel for schedule(dynamic) if (omp_threads > 1) num_threads(omp_threads)
        for (size_t ci = 0; ci < cores.size(); ci++)
        try
        {
            Core &c = cores[ci];
            for (MappedNeuron &n : c.neurons)
            {
                const bool sim_buf = (c.buffer_pos == SANAFE_BUF_BEFORE_DENDRITE) || (c.buffer_pos == SANAFE_BUF_BEFORE_SOMA);
                Result in;
                if (sim_buf)
                {
                    in = c.buffer.at(n.offset_in_core);
                    c.buffer.at(n.offset_in_core) = Result{};
                }
                const Result out = execute(n.pipeline, t, n, nullptr, in);
                c.next_delay += out.latency.value_or(0.0);
                if (n.status == FIRED)
                {
                    for (size_t aa : n.axon_out_addresses)
                    {
                        Message m = make_message(total_messages_sent++, n, t);
                        const AxonOut &ao = c.axons_out[aa];
                        const Tile &dt = tiles[ao.dest_tile_id];
                        const Core &dc = cores[dt.cores[ao.dest_core_offset]];
                        m.placeholder = 0;
                        m.spikes = dc.axons_in[ao.dest_axon_id].synapse_addresses.size();
                        m.dest_x = dt.x;
                        m.dest_y = dt.y;
                        m.dest_tile = dt.id;
                        m.dest_core_id = dc.id;
                        m.dest_core_offset = dc.offset;
                        m.dest_axon_id = ao.dest_axon_id;
                        AxonOutUnit &hw = c.axon_out_hw[0];
                        hw.energy += hw.energy_access;
                        m.generation_delay = c.next_delay + hw.latency_access;
                        c.next_delay = 0.0;
                        ts.messages.at(c.id).push_back(m);
                        ++hw.packets_out;
                    }
                }
            }
            if (c.next_delay != 0.0)
            {
                Message ph = make_message(-1, c.neurons.back(), t);
                ph.generation_delay = c.next_delay;
                ts.messages.at(c.id).push_back(ph);
            }
        }
        catch (...)
        {
#pragma omp critical(oracle_failure)
            if (!failure) failure = std::current_exception();
        }
        if (failure) std::rethrow_exception(failure);
    }

    void process_messages(Timestep &ts) // src/chip.cpp:656-764, 1127-1169
    {
        const long t = ts.t.timestep;
        for (auto &q : ts.messages)
            for (Message &m : q)
            {
                if (m.placeholder) continue;
                const Tile &src = tiles.at(m.src_tile);
                Tile &dest = tiles.at(m.dest_tile);
                const size_t xh = abs_diff(src.x, dest.x), yh = abs_diff(src.y, dest.y);
                double lat = 0.0;
                if (src.x < dest.x)
                {
                    dest.east += xh;
                    lat += static_cast<double>(xh) * src.l_hop[SANAFE_DIR_EAST];
                }
                else
                {
                    dest.west += xh;
                    lat += static_cast<double>(xh) * src.l_hop[SANAFE_DIR_WEST];
                }
                if (src.y < dest.y)
                {
                    dest.north += yh;
                    lat += static_cast<double>(yh) * src.l_hop[SANAFE_DIR_NORTH];
                }
                else
                {
                    dest.south += yh;
                    lat += static_cast<double>(yh) * src.l_hop[SANAFE_DIR_SOUTH];
                }
                dest.hops += xh + yh;
                dest.messages_received++;
                m.min_hop_delay = lat;
                m.hops = xh + yh;
                cores[dest.cores.at(m.dest_core_offset)].messages_in.push_back(&m);
            }
        std::exception_ptr failure;
#pragma omp parallel for schedule(dynamic) if (omp_threads > 1) num_threads(omp_threads)
        for (size_t ci = 0; ci < cores.size(); ci++)
        try
        {
            Core &c = cores[ci];
            for (Message *mp : c.messages_in)
            {
                Message &m = *mp;
                AxonInUnit &au = c.axon_in_hw.at(0);
                au.spike_messages_in++;
                double lat = au.latency_msg;
                const AxonIn &ai = c.axons_in.at(m.dest_axon_id);
                for (size_t sa : ai.synapse_addresses)
                {
                    MappedConnection &con = *c.connections_in.at(sa);
                    MappedNeuron &n = *con.post;
                    const Result out = execute(con.pipeline, t, n, &con, Result{});
                    c.buffer.at(n.offset_in_core) = out;
                    lat += out.latency.value_or(0.0);
                }
                m.processing_delay += lat;
            }
        }
        catch (...)
        {
#pragma omp critical(oracle_failure)
            if (!failure) failure = std::current_exception();
        }
        if (failure) std::rethrow_exception(failure);
    }

    void forced_updates(Timestep &ts) // src/chip.cpp:975-1026
    {
        const long t = ts.t.timestep;
        for (Core &c : cores)
            for (MappedNeuron &n : c.neurons)
            {
                if (n.check_synapse_updates)
                    for (MappedConnection &con : n.connections_out)
                        if (con.synapse_hw->update_every_timestep)
                        {
                            Result r = con.synapse_hw->update_syn(con.syn_addr, false, t);
                            if (r.energy.has) con.synapse_hw->energy += r.energy.v;
                        }
                if (n.dendrite_hw->update_every_timestep)
                {
                    Result r = n.dendrite_hw->update_dend(n.dend_addr, Opt(), false, 0, t);
                    if (r.energy.has) n.dendrite_hw->energy += r.energy.v;
                }
            }
    }

    double sync_delay() const // src/utils.hpp:19-44
    {
        if (sync_table.empty()) throw std::runtime_error("Table is empty");
        auto it = sync_table.upper_bound(mapped_tiles);
        if (it == sync_table.begin()) return sync_table.begin()->second;
        --it;
        return it->second;
    }

    void calc_energy(Timestep &ts) // src/chip.cpp:1171-1261
    {
        oracle_ts &o = ts.t;
        o.synapse_energy = o.dendrite_energy = o.soma_energy = o.network_energy = o.total_energy = 0.0;
        for (Tile &tile : tiles)
        {
            double hop = static_cast<double>(tile.east) * tile.e_hop[SANAFE_DIR_EAST];
            hop += static_cast<double>(tile.west) * tile.e_hop[SANAFE_DIR_WEST];
            hop += static_cast<double>(tile.south) * tile.e_hop[SANAFE_DIR_SOUTH];
            hop += static_cast<double>(tile.north) * tile.e_hop[SANAFE_DIR_NORTH];
            tile.energy = hop;
            o.network_energy += hop;
            for (size_t cid : tile.cores)
            {
                Core &c = cores[cid];
                double ain = 0.0;
                for (const auto &a : c.axon_in_hw) ain = static_cast<double>(a.spike_messages_in) * a.energy_msg;
                o.network_energy += ain;
                double pe = 0.0;
                for (Unit *u : c.in_use)
                {
                    pe += u->energy;
                    if (u->impl_syn) o.synapse_energy += u->energy;
                    if (u->impl_dend) o.dendrite_energy += u->energy;
                    if (u->impl_soma) o.soma_energy += u->energy;
                }
                double aout = 0.0;
                for (const auto &a : c.axon_out_hw) aout = a.energy;
                o.network_energy += aout;
                c.energy = ain;
                c.energy += pe;
                c.energy += aout;
                tile.energy += c.energy;
            }
            o.total_energy += tile.energy;
        }
    }

    void update_counters(Timestep &ts) // src/chip.cpp:1028-1051
    {
        for (const Tile &t : tiles)
        {
            ts.t.total_hops += t.hops;
            for (size_t cid : t.cores)
            {
                const Core &c = cores[cid];
                for (const Unit *u : c.in_use)
                {
                    ts.t.spike_count += u->spikes_processed;
                    ts.t.neurons_updated += u->neurons_updated;
                    ts.t.neurons_fired += u->neurons_fired;
                }
                for (const auto &a : c.axon_out_hw) ts.t.packets_sent += a.packets_out;
            }
        }
    }

    // ---- timing models: src/schedule.cpp ----
    double schedule_simple(Timestep &ts, double sync) // src/schedule.cpp:61-102
    {
        const size_t n = ts.messages.size();
        std::vector<double> np(n, 0.0), mp(n, 0.0);
        for (size_t sc = 0; sc < n; sc++)
            for (Message &m : ts.messages[sc])
            {
                np[sc] += m.generation_delay;
                mp[m.dest_core_id] += m.processing_delay;
                m.blocking_delay = 0.0;
                m.network_delay = m.min_hop_delay;
            }
        const double a = *std::max_element(mp.begin(), mp.end());
        const double b = *std::max_element(np.begin(), np.end());
        return std::max(a, b) + sync;
    }

    struct Noc // src/schedule.hpp:177-204
    {
        std::vector<std::list<Message>> received;
        size_t w, h, max_cpt;
        std::vector<double> density, core_finished;
        double mean_delay{0.0};
        long in_noc{0};
        size_t idx(size_t x, size_t y, size_t link) const
        {
            const size_t lpr = max_cpt + 4;
            return (x * h * lpr) + (y * lpr) + link;
        }
    };

    static void noc_density(Noc &noc, const Message &m, bool entering) // src/schedule.cpp:478-553
    {
        if (static_cast<size_t>(m.src_x) > noc.w || static_cast<size_t>(m.dest_x) > noc.w)
            throw std::runtime_error("Message x > NoC width");
        if (static_cast<size_t>(m.src_y) > noc.h || static_cast<size_t>(m.dest_y) > noc.h)
            throw std::runtime_error("Message y > NoC height");
        double adjust = 1.0 / (2.0 + static_cast<double>(m.hops));
        if (!entering) adjust *= -1.0;
        const int xi = (m.src_x < m.dest_x) ? 1 : -1, yi = (m.src_y < m.dest_y) ? 1 : -1;
        size_t prev = 4 + m.src_core_offset;
        for (int64_t x = m.src_x; x != m.dest_x; x += xi)
        {
            const int dir = (xi > 0) ? 1 : 3; // east : west (src/schedule.hpp:32-39)
            if (x == m.src_x) noc.density[noc.idx(x, m.src_y, 4 + m.src_core_offset)] += adjust;
            else noc.density[noc.idx(x, m.src_y, dir)] += adjust;
            prev = dir;
        }
        for (int64_t y = m.src_y; y != m.dest_y; y += yi)
        {
            const int dir = (yi > 0) ? 0 : 2; // north : south
            if (m.src_x == m.dest_x && y == m.src_y) noc.density[noc.idx(m.dest_x, y, 4 + m.src_core_offset)] += adjust;
            else noc.density[noc.idx(m.dest_x, y, prev)] += adjust;
            prev = dir;
        }
        if (m.src_x == m.dest_x && m.src_y == m.dest_y)
            noc.density[noc.idx(m.dest_x, m.dest_y, 4 + m.src_core_offset)] += adjust;
        else
            noc.density[noc.idx(m.dest_x, m.dest_y, prev)] += adjust;
    }

    static double noc_congestion(const Noc &noc, const Message &m) // src/schedule.cpp:555-611
    {
        const int xi = (m.src_x < m.dest_x) ? 1 : -1, yi = (m.src_y < m.dest_y) ? 1 : -1;
        double flow = 0.0;
        size_t prev = 4 + m.src_core_offset;
        for (int64_t x = m.src_x; x != m.dest_x; x += xi)
        {
            const int dir = (xi > 0) ? 1 : 3;
            if (x == m.src_x) flow += noc.density[noc.idx(x, m.src_y, 4 + m.src_core_offset)];
            else flow += noc.density[noc.idx(x, m.src_y, dir)];
            prev = dir;
        }
        for (int64_t y = m.src_y; y != m.dest_y; y += yi)
        {
            const int dir = (yi > 0) ? 0 : 2;
            if (m.src_x == m.dest_x && y == m.src_y) flow += noc.density[noc.idx(m.dest_x, y, 4 + m.src_core_offset)];
            else flow += noc.density[noc.idx(m.dest_x, y, prev)];
            prev = dir;
        }
        if (m.src_x == m.dest_x && m.src_y == m.dest_y) flow += noc.density[noc.idx(m.dest_x, m.dest_y, 4 + m.src_core_offset)];
        else flow += noc.density[noc.idx(m.dest_x, m.dest_y, prev)];
        return flow;
    }

    static void noc_track(Noc &noc, const Message &m, bool entering) // src/schedule.cpp:402-476
    {
        noc_density(noc, m, entering);
        if (entering)
        {
            noc.mean_delay += (m.processing_delay - noc.mean_delay) / (static_cast<double>(noc.in_noc) + 1.0);
            noc.in_noc++;
        }
        else
        {
            if (noc.in_noc > 1)
                noc.mean_delay += (noc.mean_delay - m.processing_delay) / (static_cast<double>(noc.in_noc) - 1.0);
            else
                noc.mean_delay = 0.0;
            noc.in_noc--;
        }
    }

    struct BySent
    {
        bool operator()(const Message &a, const Message &b) const noexcept { return a.sent_timestamp > b.sent_timestamp; }
    };

    double schedule_detailed(Timestep &ts, double sync) // src/schedule.cpp:208-400
    {
        Noc noc;
        noc.w = noc_w;
        noc.h = noc_h;
        noc.max_cpt = max_cores_per_tile;
        noc.received.resize(cores.size());
        noc.core_finished.assign(cores.size(), 0.0);
        noc.density.assign(noc_h * noc_w * (4 + max_cores_per_tile), 0.0);
        std::vector<std::list<Message>> sent = ts.messages;
        std::vector<std::list<Message>> scheduled(cores.size());
        std::priority_queue<Message, std::vector<Message>, BySent> pq;
        for (auto &q : sent)
            if (!q.empty())
            {
                Message m = q.front();
                q.pop_front();
                m.sent_timestamp = m.generation_delay;
                pq.push(m);
            }
        double last = 0.0;
        while (!pq.empty())
        {
            Message m = pq.top();
            pq.pop();
            last = std::max(last, m.sent_timestamp);
            const double tnow = m.sent_timestamp;
            for (auto &q : noc.received)
                q.remove_if([&](Message &r) {
                    if (r.in_noc && tnow >= r.received_timestamp)
                    {
                        r.in_noc = false;
                        noc_track(noc, r, false);
                        return true;
                    }
                    return false;
                });
            if (!m.placeholder)
            {
                const size_t dc = m.dest_core_id;
                m.messages_along_route = noc_congestion(noc, m);
                const double cap = static_cast<double>((m.hops + 1UL) * noc_buf);
                if (m.messages_along_route > cap)
                {
                    m.blocking_delay = (m.messages_along_route - cap) * noc.mean_delay;
                    m.sent_timestamp += m.blocking_delay;
                }
                else
                {
                    m.blocking_delay = 0.0;
                }
                const double congestion = m.messages_along_route * noc.mean_delay / (static_cast<double>(m.hops) + 1.0);
                m.network_delay = std::max(m.min_hop_delay, congestion);
                const double earliest = m.sent_timestamp + m.network_delay;
                m.received_timestamp = std::max(noc.core_finished[dc], earliest);
                noc.core_finished[dc] = std::max(noc.core_finished[dc] + m.processing_delay, earliest + m.processing_delay);
                m.processed_timestamp = noc.core_finished[dc];
                m.in_noc = true;
                noc.received[dc].push_back(m);
                noc_track(noc, m, true);
                last = std::max(last, m.processed_timestamp);
            }
            const size_t sc = m.src_core_id;
            if (!sent[sc].empty())
            {
                Message &nx = sent[sc].front();
                nx.sent_timestamp = m.sent_timestamp + nx.generation_delay;
                pq.push(nx);
                const double lt = nx.sent_timestamp;
                sent[sc].pop_front();
                last = std::max(last, lt);
            }
            scheduled[sc].push_back(m);
        }
        ts.messages = std::move(scheduled);
        return last + sync;
    }

    void step(int timing, oracle_ts *out) // src/chip.cpp:549-560, 1053-1108
    {
        ++total_timesteps;
        Timestep ts;
        ts.t.timestep = total_timesteps;
        ts.messages.resize(cores.size());
        reset_measurements();
        process_neurons(ts);
        process_messages(ts);
        forced_updates(ts);
        const double sync = sync_delay();
        calc_energy(ts);
        update_counters(ts);
        if (timing == ORACLE_TIMING_SIMPLE) ts.t.sim_time = schedule_simple(ts, sync);
        else ts.t.sim_time = schedule_detailed(ts, sync);
        ts.t.n_messages = 0;
        for (const auto &q : ts.messages) ts.t.n_messages += q.size();
        last = std::move(ts);
        if (out) *out = last.t;
    }
};

// ---------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------
static void set_err(char *err, int errlen, const std::string &s)
{
    if (err && errlen > 0)
    {
        std::snprintf(err, errlen, "%s", s.c_str());
    }
}

extern "C" oracle_chip *oracle_create(const sanafe_desc *desc, char *err, int errlen)
{
    try
    {
        auto chip = std::make_unique<oracle_chip>();
        for (int i = 0; i < desc->n_strings; i++) chip->strings.emplace_back(desc->strings[i]);
        chip->build_arch(*desc);
        chip->load(*desc);
        return chip.release();
    }
    catch (const std::exception &e)
    {
        set_err(err, errlen, e.what());
        return nullptr;
    }
}

extern "C" void oracle_destroy(oracle_chip *chip) { delete chip; }

extern "C" int oracle_step(oracle_chip *chip, int timing_model, oracle_ts *out, char *err, int errlen)
{
    try
    {
        chip->step(timing_model, out);
        return 0;
    }
    catch (const std::exception &e)
    {
        set_err(err, errlen, e.what());
        return -1;
    }
}

extern "C" void oracle_set_threads(oracle_chip *chip, int n_threads) { chip->omp_threads = n_threads > 1 ? n_threads : 1; }

extern "C" void oracle_get_status(const oracle_chip *chip, uint8_t *out)
{
    for (size_t i = 0; i < chip->by_gid.size(); i++) out[i] = chip->by_gid[i] ? chip->by_gid[i]->status : 0;
}

extern "C" void oracle_get_potentials(const oracle_chip *chip, double *out)
{
    for (size_t i = 0; i < chip->by_gid.size(); i++)
    {
        const MappedNeuron *n = chip->by_gid[i];
        out[i] = n ? n->soma_hw->get_potential(n->soma_addr) : 0.0;
    }
}

extern "C" void oracle_get_trace(const oracle_chip *chip, const char *name, double *out)
{
    for (size_t i = 0; i < chip->by_gid.size(); i++)
    {
        const MappedNeuron *n = chip->by_gid[i];
        out[i] = std::numeric_limits<double>::quiet_NaN();
        if (!n) continue;
        auto tr = n->dendrite_hw->get_traces(n->dend_addr);
        auto st = n->soma_hw->get_traces(n->soma_addr);
        tr.merge(st); // dendrite wins on clashes, like std::map::merge in src/chip.cpp:1819-1822
        auto it = tr.find(name);
        if (it != tr.end()) out[i] = it->second;
    }
}

extern "C" int64_t oracle_get_messages(const oracle_chip *chip, oracle_msg *out, int64_t cap)
{
    int64_t k = 0;
    for (const auto &q : chip->last.messages)
        for (const Message &m : q)
        {
            if (out && k < cap) out[k] = static_cast<const oracle_msg &>(m);
            k++;
        }
    return k;
}

// sim_trace_get_optional_traces, src/chip.cpp:1541-1579: the per-tile / per-core / per-unit columns of the perf
// trace, in std::map (lexicographic) order; values are those of the timestep just simulated.  Names are written
// NUL-separated into `names`; returns the number of columns.
extern "C" int64_t oracle_optional_traces(const oracle_chip *chip, char *names, int64_t names_cap, double *values, int64_t cap)
{
    std::map<std::string, double> opt;
    const sanafe_desc &d = *chip->desc;
    for (const Tile &t : chip->tiles)
    {
        const std::string tn = chip->str(d.tile_name[t.id]);
        if (d.tile_log_energy[t.id]) opt[tn + ".energy"] = t.energy;
        for (size_t cid : t.cores)
        {
            const Core &c = chip->cores[cid];
            const std::string cn = chip->str(d.core_name[cid]);
            if (d.core_log_energy[cid]) opt[tn + "." + cn + ".energy"] = c.energy;
            // every unit of the core exists in the reference from construction on (src/chip.cpp:83-87); this oracle
            // instantiates on first use, so a unit nothing was mapped to is listed from the description, at 0.0
            for (size_t slot = 0; slot < c.pipeline_hw.size(); slot++)
            {
                const int u = d.tmpl_unit_ptr[c.tmpl] + static_cast<int>(slot);
                const Unit *hw = c.pipeline_hw[slot].get();
                const std::string base = tn + "." + cn + "." + chip->str(d.unit_name[u]);
                if (d.unit_flags[u] & SANAFE_UNIT_LOG_ENERGY) opt[base + ".energy"] = hw ? hw->energy : 0.0;
                if (d.unit_flags[u] & SANAFE_UNIT_LOG_LATENCY) opt[base + ".latency"] = hw ? hw->latency : 0.0;
            }
        }
    }
    int64_t k = 0, pos = 0;
    for (const auto &kv : opt)
    {
        if (values && k < cap) values[k] = kv.second;
        if (names && pos + static_cast<int64_t>(kv.first.size()) + 1 <= names_cap)
        {
            std::memcpy(names + pos, kv.first.c_str(), kv.first.size() + 1);
            pos += static_cast<int64_t>(kv.first.size()) + 1;
        }
        k++;
    }
    return k;
}

extern "C" void oracle_reset(oracle_chip *chip) // src/chip.cpp:576-600
{
    for (Core &c : chip->cores)
    {
        std::fill(c.buffer.begin(), c.buffer.end(), Result{});
        for (auto &hw : c.pipeline_hw)
            if (hw) hw->reset();
        for (MappedNeuron &n : c.neurons) n.status = UNSET;
    }
}

extern "C" int oracle_set_neuron_attr(oracle_chip *chip, int64_t neuron, const char *key, int type, double num,
        const char *str, const double *list, int64_t list_len, int fwd, char *err, int errlen)
{
    try
    {
        Attr a;
        a.key = key;
        a.type = type;
        a.num = num;
        if (str) a.str = str;
        if (list) a.list.assign(list, list + list_len);
        a.fwd = fwd;
        oracle_chip::neuron_set_attr(*chip->by_gid.at(neuron), a);
        return 0;
    }
    catch (const std::exception &e)
    {
        set_err(err, errlen, e.what());
        return -1;
    }
}

extern "C" int64_t oracle_mapped_tiles(const oracle_chip *chip) { return chip->mapped_tiles; }

// ---------------------------------------------------------------------------
// Unit-level hooks: drive one restated model directly, with the same call
// shapes as oracle/ref_models_driver.cpp drives the reference's model, so the
// tests can compare them call by call.
// ---------------------------------------------------------------------------
struct oracle_unit
{
    std::unique_ptr<Unit> hw;
};
struct oracle_unit_result
{
    int has_current;
    double current;
    int status;
    int has_energy;
    double energy;
    int has_latency;
    double latency;
};
static void fill_result(oracle_unit_result *out, const Result &r)
{
    out->has_current = r.current.has;
    out->current = r.current.value_or(0.0);
    out->status = r.status;
    out->has_energy = r.energy.has;
    out->energy = r.energy.value_or(0.0);
    out->has_latency = r.latency.has;
    out->latency = r.latency.value_or(0.0);
}
static Attr mk_attr(const char *key, int type, double num, const char *str, const double *list, long n)
{
    Attr a;
    a.key = key;
    a.type = type;
    a.num = num;
    if (str) a.str = str;
    if (list) a.list.assign(list, list + n);
    return a;
}
#define OGUARD(body)                        \
    try                                     \
    {                                       \
        body;                               \
        return 0;                           \
    }                                       \
    catch (const std::exception &e)         \
    {                                       \
        set_err(err, errlen, e.what());     \
        return -1;                          \
    }
static unsigned g_unit_input_instances = 0;
extern "C" oracle_unit *oracle_unit_create(const char *model, char *err, int errlen)
{
    try
    {
        oracle_chip tmp;
        tmp.input_instances = g_unit_input_instances;
        auto u = std::make_unique<oracle_unit>();
        u->hw = tmp.make_unit(model);
        g_unit_input_instances = tmp.input_instances;
        return u.release();
    }
    catch (const std::exception &e)
    {
        set_err(err, errlen, e.what());
        return nullptr;
    }
}
extern "C" void oracle_unit_destroy(oracle_unit *u) { delete u; }
extern "C" int oracle_unit_set_attr_hw(oracle_unit *u, const char *key, int type, double num, const char *str,
        const double *list, long n, char *err, int errlen)
{
    OGUARD(u->hw->set_attr_hw(mk_attr(key, type, num, str, list, n)))
}
extern "C" int oracle_unit_set_attr_neuron(oracle_unit *u, long addr, const char *key, int type, double num,
        const char *str, const double *list, long n, char *err, int errlen)
{
    OGUARD(u->hw->set_attr_neuron(addr, mk_attr(key, type, num, str, list, n)))
}
extern "C" int oracle_unit_set_attr_edge(oracle_unit *u, long addr, const char *key, int type, double num,
        const char *str, const double *list, long n, char *err, int errlen)
{
    OGUARD(u->hw->set_attr_edge(addr, mk_attr(key, type, num, str, list, n)))
}
extern "C" int oracle_unit_update_syn(oracle_unit *u, long addr, int read, long t, oracle_unit_result *out, char *err,
        int errlen)
{
    OGUARD(fill_result(out, u->hw->update_syn(addr, read != 0, t)))
}
extern "C" int oracle_unit_update_dend(oracle_unit *u, long naddr, int has_cur, double cur, int has_syn, long syn,
        long t, oracle_unit_result *out, char *err, int errlen)
{
    OGUARD(fill_result(out, u->hw->update_dend(naddr, has_cur ? Opt(cur) : Opt(), has_syn != 0, syn, t)))
}
extern "C" int oracle_unit_update_soma(oracle_unit *u, long naddr, int has_cur, double cur, long t,
        oracle_unit_result *out, char *err, int errlen)
{
    OGUARD(fill_result(out, u->hw->update_soma(naddr, has_cur ? Opt(cur) : Opt(), t)))
}
extern "C" double oracle_unit_get_potential(oracle_unit *u, long addr) { return u->hw->get_potential(addr); }
extern "C" int oracle_unit_get_trace(oracle_unit *u, long addr, const char *name, double *out)
{
    auto tr = u->hw->get_traces(addr);
    auto it = tr.find(name);
    if (it == tr.end()) return 0;
    *out = it->second;
    return 1;
}
extern "C" void oracle_unit_reset(oracle_unit *u) { u->hw->reset(); }
