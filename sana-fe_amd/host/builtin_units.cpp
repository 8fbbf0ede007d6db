// builtin_units.cpp -- the built-in models as host pipeline units, for cores that run on the host.
//
// A core runs on the host when its pipeline cannot run on the device by construction (mapper.hpp: MappedChip::HostCore):
// the soma unit sits in the MESSAGE pipeline and is called once per synaptic event, or a synapse / dendrite unit is a
// plugin.  Such a core may still name built-in models for its other units (`current_based` synapses in front of a plugin
// dendrite, a `truenorth` soma behind the buffer before axon_out ...); those then have to be host objects with the same
// `update()` interface as the plugins they are chained with.  These classes restate the reference's models
// (src/models.cpp) against this build's plugin headers; the device kernels hold the same arithmetic for every core that
// can run on the GPU, and nothing here is ever used for such a core.
#include <algorithm>
#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sanafe_hip.h"
#include "host_cores.hpp"
#include "mapper.hpp"
#include "plugin_abi/pipeline.hpp"

namespace sanafe_amd
{
namespace
{
using sanafe::ModelAttribute;
using sanafe::PipelineResult;

// ModelAttribute -> the attribute record the mapper's soma-parameter code takes (apply_soma_attribute)
SomaAttr to_soma_attr(const std::string &key, const ModelAttribute &a)
{
    SomaAttr s;
    s.key = key;
    if (const bool *b = std::get_if<bool>(&a.value)) s.type = SANAFE_ATTR_BOOL, s.num = *b ? 1.0 : 0.0;
    else if (const int *i = std::get_if<int>(&a.value)) s.type = SANAFE_ATTR_INT, s.num = *i;
    else if (const double *d = std::get_if<double>(&a.value)) s.type = SANAFE_ATTR_DOUBLE, s.num = *d;
    else if (const std::string *t = std::get_if<std::string>(&a.value)) s.type = SANAFE_ATTR_STRING, s.str = *t;
    else s.type = SANAFE_ATTR_LIST;
    return s;
}

// CurrentBasedSynapseModel, src/models.cpp:29-64
class HostCurrentBasedSynapse : public sanafe::SynapseUnit
{
public:
    PipelineResult update(size_t synapse_address, bool read, long int /*timestep*/) override
    {
        PipelineResult out;
        out.current = read ? weights.at(synapse_address) : 0.0;
        return out;
    }
    void set_attribute_hw(const std::string &, const ModelAttribute &) override {}
    void set_attribute_edge(size_t synapse_address, const std::string &key, const ModelAttribute &param) override
    {
        if (weights.size() <= synapse_address) weights.resize(std::max(weights.size() * 2, synapse_address + 1));
        if (key == "w" || key == "weight") weights.at(synapse_address) = static_cast<double>(param);
    }
    void reset() override {}

private:
    std::vector<double> weights;
};

// AccumulatorModel, src/models.cpp:71-94 (1024 neurons per unit, src/models.hpp:58-102)
class HostAccumulator : public sanafe::DendriteUnit
{
public:
    HostAccumulator() : charge(1024), last_step(1024, 0) {}
    PipelineResult update(size_t n, std::optional<double> current, std::optional<size_t>, long int t) override
    {
        if (last_step.at(n) < t)
        {
            charge.at(n) = 0.0;
            last_step.at(n) = t;
        }
        if (current.has_value()) charge.at(n) = charge.at(n).value_or(0.0) + *current;
        PipelineResult out;
        out.current = charge.at(n);
        return out;
    }
    void set_attribute_hw(const std::string &, const ModelAttribute &) override {}
    void set_attribute_neuron(size_t, const std::string &, const ModelAttribute &) override {}
    void set_attribute_edge(size_t, const std::string &, const ModelAttribute &) override {}
    void reset() override { std::fill(charge.begin(), charge.end(), std::optional<double>(0.0)); }

private:
    std::vector<std::optional<double>> charge;
    std::vector<long> last_step;
};

// AccumulatorWithDelayModel, src/models.cpp:96-165 (max_delay 5, src/models.hpp:158)
class HostAccumulatorWithDelay : public sanafe::DendriteUnit
{
public:
    static constexpr size_t max_delay = 5;
    HostAccumulatorWithDelay() : now(1024), line(max_delay + 1, std::vector<std::optional<double>>(1024)), stepped(1024, 0) {}
    PipelineResult update(size_t n, std::optional<double> current, std::optional<size_t> synapse, long int t) override
    {
        while (stepped[n] < t)
        {
            ++stepped[n];
            now[n] = line[0][n];
            for (size_t i = 0; i + 1 < line.size(); i++) line[i][n] = line[i + 1][n];
            line.back()[n] = std::nullopt;
        }
        if (current.has_value())
        {
            const size_t syn = synapse.value_or(0);
            const size_t d = syn < delays.size() ? delays[syn] : 0;
            line[d][n] = line[d][n].value_or(0.0) + *current;
        }
        PipelineResult out;
        out.current = now[n];
        return out;
    }
    void set_attribute_hw(const std::string &, const ModelAttribute &) override {}
    void set_attribute_neuron(size_t, const std::string &, const ModelAttribute &) override {}
    void set_attribute_edge(size_t synapse_address, const std::string &key, const ModelAttribute &param) override
    {
        if (delays.size() <= synapse_address) delays.resize(synapse_address + 1, 0);
        if (key == "delay" || key == "d")
        {
            const int d = static_cast<int>(param);
            if (static_cast<size_t>(d) > max_delay) throw std::runtime_error("Error: delay > max delay\n");
            delays[synapse_address] = static_cast<size_t>(d);
        }
    }
    void reset() override
    {
        std::fill(now.begin(), now.end(), std::nullopt);
        for (auto &row : line) std::fill(row.begin(), row.end(), std::nullopt);
    }

private:
    std::vector<std::optional<double>> now;
    std::vector<std::vector<std::optional<double>>> line;
    std::vector<long> stepped;
    std::vector<size_t> delays;
};

// MultiTapModel1D, src/models.cpp:167-348: ONE RC line per unit instance (its state is not indexed by the neuron address)
class HostMultiTap : public sanafe::DendriteUnit
{
public:
    PipelineResult update(size_t, std::optional<double> current, std::optional<size_t> synapse, long int t) override
    {
        while (stepped < t)
        {
            ++stepped;
            advance();
        }
        if (current.has_value())
        {
            int tap = 0;
            if (synapse.has_value() && *synapse < synapse_to_tap.size()) tap = synapse_to_tap[*synapse];
            if (tap < 0 || static_cast<size_t>(tap) >= v.size()) throw std::logic_error("Tap should be >= 0 and less than taps.\n");
            v[static_cast<size_t>(tap)] += *current;
        }
        PipelineResult out;
        out.current = v[0]; // the most proximal tap
        return out;
    }
    void set_attribute_hw(const std::string &, const ModelAttribute &) override {}
    void set_attribute_neuron(size_t, const std::string &key, const ModelAttribute &param) override
    {
        if (key == "taps")
        {
            const size_t n = static_cast<size_t>(static_cast<int>(param));
            if (n == 0) throw std::invalid_argument("Number of taps must be > 0\n");
            v.resize(n);
            next.resize(n);
            tc.resize(n);
            sc.resize(n - 1);
        }
        else if (key == "time_constants")
        {
            tc = static_cast<std::vector<double>>(param);
            if (tc.size() < v.size())
                throw std::invalid_argument("Expected " + std::to_string(v.size()) + " but received " + std::to_string(tc.size()) + "time constants.");
        }
        else if (key == "space_constants")
        {
            sc = static_cast<std::vector<double>>(param);
            if (sc.size() < v.size() - 1)
                throw std::invalid_argument("Expected " + std::to_string(v.size() - 1) + " but received " + std::to_string(tc.size()) + "time constants.");
        }
    }
    void set_attribute_edge(size_t address, const std::string &key, const ModelAttribute &param) override
    {
        if (key != "tap") return;
        if (synapse_to_tap.size() <= address) synapse_to_tap.resize(address + 1, 0);
        synapse_to_tap[address] = static_cast<int>(param);
    }
    void reset() override
    {
        std::fill(v.begin(), v.end(), 0.0);
        std::fill(next.begin(), next.end(), 0.0);
    }

private:
    void advance() // one time-step of the line: decay, then exchange with both neighbours (src/models.cpp:167-205)
    {
        const size_t n = v.size();
        for (size_t k = 0; k < n; k++) next[k] = v[k] * tc[k];
        for (size_t k = 0; k < n; k++)
        {
            if (k > 0)
            {
                const double towards_soma = v[k] * sc[k - 1];
                next[k - 1] += towards_soma;
                next[k] -= towards_soma;
            }
            if (k + 1 < n)
            {
                const double away = v[k] * sc[k];
                next[k + 1] += away;
                next[k] -= away;
            }
        }
        v = next;
    }
    std::vector<double> v{0.0}, next{0.0}, tc{0.0}, sc;
    std::vector<int> synapse_to_tap;
    long stepped{0};
};

// the soma parameter set of one neuron: the same record, attribute rules (apply_soma_attribute) and arithmetic as the
// device's LIF / TrueNorth updates (csrc/sanafe_kernels.hpp: neuron_kernel)
struct SomaNeuron
{
    sanafe_hip_soma_class p{};
    double bias{0.0}, v{0.0}, input_current{0.0};
    int refractory{0};
    long steps{0};
};

// x86-64 static_cast<int>(double): out of range and NaN give INT_MIN (the reference quantises with it, src/models.cpp:447-455)
int cvt_int_x86(double x)
{
    if (!(x > -2147483649.0 && x < 2147483648.0)) return static_cast<int>(0x80000000u);
    return static_cast<int>(x);
}

// LoihiLifModel, src/models.cpp:351-576 (1024 compartments per unit, src/models.hpp:268)
class HostLoihiLif : public sanafe::SomaUnit
{
public:
    HostLoihiLif() : cx(1024)
    {
        for (SomaNeuron &n : cx)
        {
            n.p.leak_decay = 1.0;
            n.p.reset_mode = SANAFE_RESET_HARD;
            n.p.reverse_reset_mode = SANAFE_RESET_NONE;
        }
    }
    PipelineResult update(size_t address, std::optional<double> current, long int t) override
    {
        SomaNeuron &n = cx[address];
        if (n.steps == t) throw std::runtime_error("This model does not support multiple updates to the same compartment in one time-step.");
        if (n.steps < t - 1) throw std::runtime_error("This model must update every time-step.\n");
        sanafe::NeuronStatus state = sanafe::idle;
        if (std::fabs(n.v) > 0.0 || current.has_value() || std::fabs(n.bias) > 0.0 || n.p.force_update) state = sanafe::updated;
        if (n.steps > 0)
        {
            n.input_current *= n.p.input_decay;
            n.v *= n.p.leak_decay;
        }
        n.v = static_cast<double>(cvt_int_x86(n.v * 64.0)) / 64.0;
        if (!(n.refractory > 0))
        {
            n.v += n.bias;
            n.input_current += current.value_or(0.0);
            n.v += n.input_current;
            bool fired = false;
            if (n.v > n.p.threshold)
            {
                if (n.p.reset_mode == SANAFE_RESET_HARD) n.v = n.p.reset;
                else if (n.p.reset_mode == SANAFE_RESET_SOFT) n.v -= n.p.threshold;
                n.refractory = n.p.refractory_delay;
                fired = true;
            }
            if (n.v < n.p.reverse_threshold)
            {
                if (n.p.reverse_reset_mode == SANAFE_RESET_SOFT) n.v -= n.p.reverse_threshold;
                else if (n.p.reverse_reset_mode == SANAFE_RESET_HARD) n.v = n.p.reverse_reset;
                else if (n.p.reverse_reset_mode == SANAFE_RESET_SATURATE) n.v = n.p.reverse_threshold;
            }
            if (fired) state = sanafe::fired;
        }
        ++n.steps;
        n.refractory = std::max(0, n.refractory - 1);
        PipelineResult out;
        out.status = state;
        return out;
    }
    void set_attribute_hw(const std::string &key, const ModelAttribute &) override
    {
        if (key == "noise") throw std::runtime_error("LIF noise files are not available on a core that runs on the host");
    }
    void set_attribute_neuron(size_t address, const std::string &key, const ModelAttribute &param) override
    {
        SomaNeuron &n = cx.at(address);
        SomaAttrEffect fx;
        apply_soma_attribute(SANAFE_SOMA_LIF, to_soma_attr(key, param), n.p, fx);
        if (fx.bias_set) n.bias = fx.bias;
        if (fx.potential_set) n.v = fx.potential;
    }
    void reset() override
    {
        for (SomaNeuron &n : cx) n.input_current = 0.0, n.v = 0.0;
    }
    double get_potential(size_t address) override { return cx.at(address).v; }
    std::map<std::string, double> get_neuron_traces(size_t address) override { return {{"u", cx.at(address).input_current}}; }

private:
    std::vector<SomaNeuron> cx;
};

// TrueNorthModel, src/models.cpp:653-830 (4096 neurons per unit, src/models.hpp:284)
class HostTrueNorth : public sanafe::SomaUnit
{
public:
    HostTrueNorth() : nr(4096)
    {
        for (SomaNeuron &n : nr)
        {
            n.p.reset_mode = SANAFE_RESET_HARD;
            n.p.reverse_reset_mode = SANAFE_RESET_NONE;
            n.p.leak_towards_zero = 1;
        }
    }
    PipelineResult update(size_t address, std::optional<double> current, long int /*t*/) override
    {
        SomaNeuron &n = nr[address];
        sanafe::NeuronStatus state = sanafe::idle;
        if (std::fabs(n.v) > 0.0 || current.has_value() || std::fabs(n.bias) > 0.0 || n.p.force_update) state = sanafe::updated;
        if (n.p.leak_towards_zero)
        {
            if (n.v > 0.0) n.v -= n.p.leak_decay;
            else if (n.v < 0.0) n.v += n.p.leak_decay;
        }
        else
        {
            n.v += n.p.leak_decay;
        }
        n.v += n.bias;
        if (current.has_value()) n.v += *current;
        const double vt = n.v; // (random_mask is refused at load() for host cores: no rand() term)
        if (vt >= n.p.threshold)
        {
            if (n.p.reset_mode == SANAFE_RESET_HARD) n.v = n.p.reset;
            else if (n.p.reset_mode == SANAFE_RESET_SOFT) n.v -= n.p.threshold;
            else if (n.p.reset_mode == SANAFE_RESET_SATURATE) n.v = n.p.threshold;
            state = sanafe::fired;
        }
        else if (vt <= n.p.reverse_threshold)
        {
            if (n.p.reverse_reset_mode == SANAFE_RESET_HARD) n.v = n.p.reverse_reset;
            else if (n.p.reverse_reset_mode == SANAFE_RESET_SOFT) n.v += n.p.reverse_threshold;
            else if (n.p.reverse_reset_mode == SANAFE_RESET_SATURATE) n.v = n.p.reverse_threshold;
        }
        PipelineResult out;
        out.status = state;
        return out;
    }
    void set_attribute_hw(const std::string &, const ModelAttribute &) override {}
    void set_attribute_neuron(size_t address, const std::string &key, const ModelAttribute &param) override
    {
        SomaNeuron &n = nr.at(address);
        SomaAttrEffect fx;
        apply_soma_attribute(SANAFE_SOMA_TRUENORTH, to_soma_attr(key, param), n.p, fx);
        if (fx.bias_set) n.bias = fx.bias;
        if (fx.random_mask_set && fx.random_mask != 0) throw std::runtime_error("TrueNorth random_mask is not available on a core that runs on the host");
    }
    void reset() override
    {
        for (SomaNeuron &n : nr) n.v = 0.0;
    }
    double get_potential(size_t address) override { return nr.at(address).v; }

private:
    std::vector<SomaNeuron> nr;
};

// InputModel, src/models.cpp:832-903: ONE train, cursor and rate per unit instance
class HostInput : public sanafe::SomaUnit
{
public:
    PipelineResult update(size_t, std::optional<double> current, long int t) override
    {
        if (current.has_value() && *current != 0.0)
            throw std::runtime_error("Current sent to input neuron which cannot be processed (" + std::to_string(*current) + ")");
        bool send = false;
        if (cursor < train.size()) send = train[cursor++];
        if (rate > 0.0 && (t % static_cast<long int>(1.0 / rate)) == 0) send = true;
        PipelineResult out;
        out.status = send ? sanafe::fired : sanafe::idle;
        return out;
    }
    void set_attribute_hw(const std::string &, const ModelAttribute &) override {}
    void set_attribute_neuron(size_t, const std::string &key, const ModelAttribute &param) override
    {
        if (key == "spikes")
        {
            train = static_cast<std::vector<bool>>(param);
            cursor = 0;
        }
        else if (key == "poisson")
        {
            if (static_cast<double>(param) > 0.0) throw std::runtime_error("Poisson inputs are not available on a core that runs on the host");
        }
        else if (key == "rate")
        {
            rate = static_cast<double>(param);
            if (rate > 1.0) throw std::invalid_argument("input rate > 1 makes the reference divide by zero (SURVEY quirk 14)");
        }
    }
    void reset() override {}

private:
    std::vector<bool> train;
    size_t cursor{0};
    double rate{0.0};
};
} // namespace

std::unique_ptr<sanafe::PipelineUnit> make_builtin_host_unit(const std::string &model)
{
    if (model == "current_based") return std::make_unique<HostCurrentBasedSynapse>();
    if (model == "accumulator") return std::make_unique<HostAccumulator>();
    if (model == "accumulator_with_delay") return std::make_unique<HostAccumulatorWithDelay>();
    if (model == "taps") return std::make_unique<HostMultiTap>();
    if (model == "leaky_integrate_fire") return std::make_unique<HostLoihiLif>();
    if (model == "truenorth") return std::make_unique<HostTrueNorth>();
    if (model == "input") return std::make_unique<HostInput>();
    throw std::invalid_argument("Pipeline model not supported on a core that runs on the host (" + model + ")");
}
} // namespace sanafe_amd
