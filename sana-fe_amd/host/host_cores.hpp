// host_cores.hpp -- replay of the cores that cannot run on the device (mapper.hpp: MappedChip::HostCore).
//
// The reference runs every pipeline unit as a host C++ object (src/pipeline.hpp:69-301).  The MI355X path moves the
// built-in models of a core onto the device -- except where the pipeline's SHAPE forbids it:
//   * `buffer_position: soma` with the buffer inside the unit, or `axon_out`: the soma is part of the message
//     pipeline and is called once per synaptic event (src/mapped.cpp:27-58), in delivery order;
//   * a synapse or dendrite unit that is a plugin (`extern "C" PipelineUnit *create_<model>()`, src/plugins.cpp:45-98):
//     its `update()` runs per synaptic event on the host.
// For such cores the device keeps only the neurons' status and spike bits; this engine replays the core's neuron
// pipeline and -- from the chip's spike bitmap -- its message pipeline per timestep, in the reference's order
// (process_neuron / process_message / execute_pipeline / PipelineUnit::process, src/chip.cpp:710-789,
// src/pipeline.cpp:87-105), with the default costing of src/pipeline.hpp:511-731.
#ifndef SANAFE_HOST_HOST_CORES_HPP
#define SANAFE_HOST_HOST_CORES_HPP

#include <memory>
#include <string>
#include <vector>

#include "../../include/sanafe_desc.h"
#include "mapper.hpp"
#include "plugin_abi/pipeline.hpp"

namespace sanafe_amd
{
std::unique_ptr<sanafe::PipelineUnit> make_builtin_host_unit(const std::string &model); // builtin_units.cpp

class HostCores
{
public:
    // What one timestep of one host core adds to the chip's totals.
    struct Partial
    {
        double e_syn{0.0}, e_dend{0.0}, e_soma{0.0};
        double neuron_latency{0.0}; // sum of the neuron pipelines' latencies: the core's message generation delay (+ axon-out accesses)
        double processing{0.0};     // sum of the messages' processing delays (src/chip.cpp:738-764)
        long long updated{0}, fired{0};
    };

    HostCores(const MappedChip &mc, const sanafe_desc &d);
    ~HostCores();
    HostCores(const HostCores &) = delete;
    HostCores &operator=(const HostCores &) = delete;

    bool empty() const { return cores_.empty(); }
    // sim_reset_measurements, src/chip.cpp:1393-1445
    void begin_step();
    // process_neurons for the host cores: fills status[] (parallel to slots())
    void process_neurons(long timestep);
    // process_messages for the host cores; bits = the whole chip's spike bitmap of this step (global slots)
    void process_messages(long timestep, const uint32_t *bits);
    // forced_updates, src/chip.cpp:975-1026
    void forced_updates(long timestep);
    // sim_calculate_ts_energy for the host cores' units: fills partials()
    void end_step();
    void reset(); // SpikingChip::reset, src/chip.cpp:576-600

    const std::vector<uint32_t> &slots() const { return slots_; }           // local slots of all host-core neurons
    const std::vector<uint32_t> &slot_cores() const { return slot_cores_; } // local core of each
    const std::vector<uint8_t> &status() const { return status_; }            // after the neuron loop: who sends messages this step
    // MappedNeuron::status at the END of the step: the message pipeline's soma calls latch it too (execute_pipeline,
    // src/chip.cpp:780-783) -- what get_spikes / the Python spike trace read after step() (src/pytrace.cpp:190-222)
    const std::vector<uint8_t> &final_status() const { return final_status_; }
    const std::vector<uint32_t> &core_ids() const { return core_ids_; }     // local core id of every host core
    const std::vector<Partial> &partials() const { return partials_; }
    double potential(size_t i) const; // soma get_potential of host neuron i (order of slots())
    // MappedNeuron::set_attributes (src/mapped.cpp:113-166) for host neuron at local slot `slot`; false: not a host neuron
    bool set_attribute(uint32_t slot, const sanafe::ModelAttribute &attr);

private:
    struct UnitRt
    {
        std::unique_ptr<sanafe::PipelineUnit> obj;
        const MappedChip::HostCore::Unit *info{nullptr};
        bool used{false};
    };
    struct NeuronRt
    {
        uint32_t soma_addr{0}, dend_addr{0};
        int32_t soma_unit{0}, dend_unit{0};
        sanafe::NeuronStatus status{sanafe::neuron_state_unset};
        bool check_synapse_updates{false};
        std::vector<int32_t> pipeline; // neuron processing pipeline (unit indices)
    };
    struct CoreRt
    {
        const MappedChip::HostCore *hc{nullptr};
        std::vector<UnitRt> units;
        std::vector<NeuronRt> neurons;
        std::vector<sanafe::PipelineResult> buffer; // timestep_buffer, src/core.hpp
        size_t first{0};                            // index of the core's first neuron in slots_/status_
    };
    sanafe::PipelineResult process(CoreRt &c, int32_t unit, long t, NeuronRt &n, const MappedChip::HostCore::Synapse *con,
            const sanafe::PipelineResult &in);
    sanafe::PipelineResult execute(CoreRt &c, const int32_t *pipeline, size_t len, long t, NeuronRt &n,
            const MappedChip::HostCore::Synapse *con, const sanafe::PipelineResult &in);

    std::vector<CoreRt> cores_;
    std::vector<void *> plugin_handles_;
    std::vector<uint32_t> slots_, slot_cores_, core_ids_;
    std::vector<uint8_t> status_, final_status_;
    std::vector<Partial> partials_;
};
} // namespace sanafe_amd
#endif
