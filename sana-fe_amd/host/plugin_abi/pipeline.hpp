// pipeline.hpp -- the model base classes a SANA-FE plugin derives from, for the MI355X host.
//
// Source-compatible with the reference's plugin surface (src/pipeline.hpp:25-301): a plugin
// written against the reference (e.g. plugins/hodgkin_huxley.cpp) recompiles unchanged with
// `-I sana-fe_amd/host/plugin_abi` and is loaded through the same
// `extern "C" sanafe::PipelineUnit *create_<model>()` factory (src/plugins.cpp:45-98).
// Same names, same virtual signatures, same default behaviour (the wrong `update` overload
// throws std::logic_error), same public data members.  The host calls `update()` directly for
// every neuron mapped to the unit, once per timestep, and applies the architecture's default
// energy/latency costs itself (src/pipeline.hpp:631-714).
#ifndef SANAFE_AMD_PLUGIN_PIPELINE_HPP
#define SANAFE_AMD_PLUGIN_PIPELINE_HPP
#include <cstddef>
#include <cstdint>
#include <filesystem>
#include <map>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "attribute.hpp"
#include "fwd.hpp"
#include "mapped.hpp"

namespace sanafe
{
enum HardwareBitfield : uint8_t
{
    implements_invalid = 0U,
    implements_synapse = 1U << 0,
    implements_dendrite = 1U << 1,
    implements_soma = 1U << 2
};
inline HardwareBitfield operator|(HardwareBitfield a, HardwareBitfield b)
{
    return static_cast<HardwareBitfield>(static_cast<uint8_t>(a) | static_cast<uint8_t>(b));
}
inline HardwareBitfield operator&(HardwareBitfield a, HardwareBitfield b)
{
    return static_cast<HardwareBitfield>(static_cast<uint8_t>(a) & static_cast<uint8_t>(b));
}

struct SomaEnergyMetrics
{
    double energy_update_neuron{0.0};
    double energy_access_neuron{0.0};
    double energy_spike_out{0.0};
};
struct SomaLatencyMetrics
{
    double latency_update_neuron{0.0};
    double latency_access_neuron{0.0};
    double latency_spike_out{0.0};
};

// Aggregate, members in this order: plugins return `{current, status, energy, latency}`.
struct PipelineResult
{
    std::optional<double> current{std::nullopt};
    NeuronStatus status{neuron_state_unset};
    std::optional<double> energy{std::nullopt};
    std::optional<double> latency{std::nullopt};
};

class PipelineUnit
{
public:
    explicit PipelineUnit(const HardwareBitfield implemented)
            : implements_synapse((implemented & HardwareBitfield::implements_synapse) != 0)
            , implements_dendrite((implemented & HardwareBitfield::implements_dendrite) != 0)
            , implements_soma((implemented & HardwareBitfield::implements_soma) != 0)
    {
        if (implements_synapse && implements_soma && !implements_dendrite)
            throw std::logic_error("Invalid pipeline configuration: h/w supports synapse and soma but not dendrite functionality.");
        if (!implements_synapse && !implements_dendrite && !implements_soma)
            throw std::logic_error("H/w must implement at least one functional unit out of synapse/dendrite/soma");
    }
    PipelineUnit(const PipelineUnit &) = default;
    PipelineUnit(PipelineUnit &&) = default;
    virtual ~PipelineUnit() = default;
    PipelineUnit &operator=(const PipelineUnit &) = delete;
    PipelineUnit &operator=(PipelineUnit &&) = delete;

    virtual void set_attribute_hw(const std::string &attribute_name, const ModelAttribute &param) = 0;
    virtual void set_attribute_neuron(size_t neuron_address, const std::string &attribute_name, const ModelAttribute &param) = 0;
    virtual void set_attribute_edge(size_t synapse_address, const std::string &attribute_name, const ModelAttribute &param) = 0;
    virtual void reset() = 0;

    // synapse: (synapse address, read, timestep)
    virtual PipelineResult update(size_t /*synapse_address*/, bool /*read*/, long int /*timestep*/)
    {
        throw std::logic_error("Error: Synapse input not implemented");
    }
    // dendrite: (neuron address, current, synapse address, timestep)
    virtual PipelineResult update(size_t /*neuron_address*/, std::optional<double> /*current_in*/,
            std::optional<size_t> /*synaptic_address*/, long int /*timestep*/)
    {
        throw std::logic_error("Error: Dendrite input not implemented");
    }
    // soma: (neuron address, current, timestep)
    virtual PipelineResult update(size_t /*neuron_address*/, std::optional<double> /*current_in*/, long int /*timestep*/)
    {
        throw std::logic_error("Error: Soma input not implemented");
    }
    virtual void track_connection(size_t /*synapse_address*/, size_t /*src_neuron_id*/, size_t /*dest_neuron_id*/) {}
    virtual double get_potential(size_t /*neuron_address*/) { return 0.0; }
    virtual std::map<std::string, double> get_neuron_traces(size_t /*neuron_address*/) { return {}; }

    void register_attributes(const std::set<std::string> &attribute_names)
    {
        for (const std::string &a : attribute_names) supported_attributes[a] = "";
    }
    void register_attributes(const std::unordered_map<std::string, std::string> &attributes_with_descriptions)
    {
        for (const auto &kv : attributes_with_descriptions) supported_attributes[kv.first] = kv.second;
    }
    bool check_attribute(const std::string & /*attribute_name*/) { return true; } // warnings are disabled (max_attribute_warnings == 0)
    std::vector<std::string> get_attributes() const
    {
        std::vector<std::string> keys;
        for (const auto &kv : supported_attributes) keys.push_back(kv.first);
        return keys;
    }
    size_t add_neuron()
    {
        is_used = true;
        return static_cast<size_t>(neuron_count++);
    }

    // ---- public data members, as in the reference (src/pipeline.hpp:134-180) ----
    std::map<std::string, ModelAttribute> model_attributes;
    std::optional<std::filesystem::path> plugin_lib{std::nullopt};
    std::string name;
    std::string model;
    std::optional<double> default_energy_process_spike{std::nullopt};
    std::optional<double> default_latency_process_spike{std::nullopt};
    std::optional<double> default_energy_update{std::nullopt};
    std::optional<double> default_latency_update{std::nullopt};
    std::optional<SomaEnergyMetrics> default_soma_energy_metrics;
    std::optional<SomaLatencyMetrics> default_soma_latency_metrics;
    double energy{0.0};
    double latency{0.0};
    long int spikes_processed{0L};
    long int neurons_updated{0L};
    long int neurons_fired{0L};
    long int neuron_count{0L};
    long int connection_count{0L};
    long int attribute_warnings{0L};
    static constexpr long int max_attribute_warnings{0L};
    bool implements_synapse;
    bool implements_dendrite;
    bool implements_soma;
    bool log_energy{false};
    bool log_latency{false};
    bool is_used{false};
    bool update_every_timestep{false};

protected:
    std::unordered_map<std::string, std::string> supported_attributes;
};

class SynapseUnit : public PipelineUnit
{
public:
    SynapseUnit() : PipelineUnit(HardwareBitfield::implements_synapse) {}
    PipelineResult update(size_t synapse_address, bool read, long int timestep) override = 0;
    PipelineResult update(size_t, std::optional<double>, std::optional<size_t>, long int) final
    {
        throw std::logic_error("Error: Synapse H/W called with dendrite inputs");
    }
    PipelineResult update(size_t, std::optional<double>, long int) final
    {
        throw std::logic_error("Error: Synapse H/W called with soma inputs");
    }
    void set_attribute_neuron(size_t, const std::string &, const ModelAttribute &) final {}
};

class DendriteUnit : public PipelineUnit
{
public:
    DendriteUnit() : PipelineUnit(HardwareBitfield::implements_dendrite) {}
    PipelineResult update(size_t neuron_address, std::optional<double> current_in, std::optional<size_t> synaptic_address,
            long int timestep) override = 0;
    PipelineResult update(size_t, bool, long int) final { throw std::logic_error("Error: Dendrite H/W called with synapse inputs"); }
    PipelineResult update(size_t, std::optional<double>, long int) final
    {
        throw std::logic_error("Error: Dendrite H/W called with soma inputs");
    }
};

class SomaUnit : public PipelineUnit
{
public:
    SomaUnit() : PipelineUnit(HardwareBitfield::implements_soma) {}
    PipelineResult update(size_t neuron_address, std::optional<double> current_in, long int timestep) override = 0;
    PipelineResult update(size_t, bool, long int) final { throw std::logic_error("Error: Soma H/W called with synapse inputs"); }
    PipelineResult update(size_t, std::optional<double>, std::optional<size_t>, long int) final
    {
        throw std::logic_error("Error: Soma H/W called with dendrite inputs");
    }
    void set_attribute_edge(size_t, const std::string &, const ModelAttribute &) final {}
    void track_connection(size_t, size_t, size_t) final {}
};
}
#endif
