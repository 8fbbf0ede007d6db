// relay_units.cpp -- test fixture: a SYNAPSE plugin and a DENDRITE plugin written against the product's plugin ABI headers
// (sana-fe_amd/host/plugin_abi), loaded through `extern "C" PipelineUnit *create_<model>()` like any SANA-FE plugin
// (src/plugins.cpp:45-98).  Their arithmetic is that of the built-in `current_based` synapse and `accumulator` dendrite
// (src/models.cpp:29-94), so a network on them must produce, bit for bit, what the same network produces on the built-in
// units -- which the device runs and the oracle restates: the tests compare the host replay of plugin units against both.
//   test_synapse   hw attribute `gain` (default 1) scales the weight; with hw attribute `own_costs` the unit returns its
//                  own energy / latency per event (then the architecture must NOT give it default costs)
//   test_dendrite  an accumulator; counts its calls (`calls` neuron trace) so a test can see it really ran
#include <map>
#include <optional>
#include <string>
#include <vector>

#include "attribute.hpp"
#include "mapped.hpp"
#include "pipeline.hpp"
#include "print.hpp"

namespace
{
class TestSynapse : public sanafe::SynapseUnit
{
public:
    TestSynapse() { register_attributes({"w", "weight", "gain", "own_costs"}); }
    sanafe::PipelineResult update(size_t address, bool read, long int) override
    {
        sanafe::PipelineResult out;
        out.current = read ? gain_ * weights_.at(address) : 0.0;
        if (own_costs_)
        {
            out.energy = 20.0e-12;
            out.latency = 3.0e-9;
        }
        return out;
    }
    void set_attribute_hw(const std::string &key, const sanafe::ModelAttribute &value) override
    {
        if (key == "gain") gain_ = static_cast<double>(value);
        else if (key == "own_costs") own_costs_ = static_cast<bool>(value);
    }
    void set_attribute_edge(size_t address, const std::string &key, const sanafe::ModelAttribute &value) override
    {
        if (weights_.size() <= address) weights_.resize(address + 1, 0.0);
        if (key == "w" || key == "weight") weights_[address] = static_cast<double>(value);
    }
    void reset() override {}

private:
    std::vector<double> weights_;
    double gain_{1.0};
    bool own_costs_{false};
};

class TestDendrite : public sanafe::DendriteUnit
{
public:
    TestDendrite() { register_attributes({"anything"}); }
    sanafe::PipelineResult update(size_t n, std::optional<double> current, std::optional<size_t>, long int t) override
    {
        if (charge_.size() <= n)
        {
            charge_.resize(n + 1, 0.0);
            step_.resize(n + 1, 0);
        }
        if (step_[n] < t)
        {
            charge_[n] = 0.0;
            step_[n] = t;
        }
        if (current.has_value()) charge_[n] = charge_[n] + *current;
        calls_++;
        sanafe::PipelineResult out;
        out.current = charge_[n];
        return out;
    }
    void set_attribute_hw(const std::string &, const sanafe::ModelAttribute &) override {}
    void set_attribute_neuron(size_t, const std::string &, const sanafe::ModelAttribute &) override {}
    void set_attribute_edge(size_t, const std::string &, const sanafe::ModelAttribute &) override {}
    void reset() override { std::fill(charge_.begin(), charge_.end(), 0.0); }
    std::map<std::string, double> get_neuron_traces(size_t) override { return {{"calls", static_cast<double>(calls_)}}; }

private:
    std::vector<double> charge_;
    std::vector<long> step_;
    long calls_{0};
};
} // namespace

extern "C" sanafe::PipelineUnit *create_test_synapse() { return new TestSynapse(); }
extern "C" sanafe::PipelineUnit *create_test_dendrite() { return new TestDendrite(); }
