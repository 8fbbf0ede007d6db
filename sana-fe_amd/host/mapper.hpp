// mapper.hpp -- lowers a sanafe_desc (architecture + mapped SNN) to the flat device
// image of include/sanafe_hip.h, reproducing the reference's mapping ORDER rules:
//   SpikingChip::map_neurons / map_connections / map_axons, src/chip.cpp:186-408,
//   1263-1391; Core::map_neuron / map_connection, src/core.cpp:116-184.
// Everything is columnar; no per-neuron or per-synapse heap objects exist at any point.
#ifndef SANAFE_HOST_MAPPER_HPP
#define SANAFE_HOST_MAPPER_HPP

#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <array>
#include <vector>

#include "../../include/sanafe_desc.h"
#include "../../include/sanafe_hip.h"

namespace sanafe_amd
{
// Same exception vocabulary as the reference (src/mapped.hpp:30-38).
class HardwareMappingError : public std::runtime_error
{
public:
    explicit HardwareMappingError(const std::string &m) : std::runtime_error(m) {}
};
// Raised for configurations the MI355X backend does not implement yet.  The product never
// falls back to a CPU path: load() fails loudly instead.
class UnsupportedError : public std::runtime_error
{
public:
    explicit UnsupportedError(const std::string &m) : std::runtime_error(m) {}
};

// One attribute value as the soma models see it (ModelAttribute conversions, src/attribute.hpp:43-93).
struct SomaAttr
{
    std::string key;
    int type{SANAFE_ATTR_DOUBLE};
    double num{0.0};
    std::string str;
    double as_double() const;
    int as_int() const;
    bool as_bool() const;
    const std::string &as_string() const;
};
// What <Model>::set_attribute_neuron does with one attribute (src/models.cpp:375-439 LIF, 664-722 TrueNorth),
// applied to the parameter class / per-slot values of a neuron.  Used by the mapper at load() and by
// MappedNeuron::set_attributes between sim() calls.  Unknown keys are ignored, as in the reference.
struct SomaAttrEffect
{
    bool bias_set{false}, potential_set{false}, random_mask_set{false};
    double bias{0.0}, potential{0.0};
    uint32_t random_mask{0};
};
void apply_soma_attribute(uint32_t soma_model, const SomaAttr &a, sanafe_hip_soma_class &p, SomaAttrEffect &fx);
// Bytes of a class that define it (padding zeroed), for deduplication.
sanafe_hip_soma_class canonical_soma_class(const sanafe_hip_soma_class &p);

struct MappedChip
{
    // ---- geometry ----
    uint32_t n_tiles{0}, n_cores{0};
    std::vector<uint32_t> tile_x, tile_y;
    std::vector<uint32_t> core_tile, core_offset;
    uint32_t noc_width{1}, noc_height{1}, noc_buffer{0}, max_cores_per_tile{0};
    uint64_t mapped_tiles{0}, mapped_cores{0};
    double sync_delay{0.0};

    // ---- slot layout (global, all ranks) ----
    uint32_t n_global_slots{0};
    std::vector<uint32_t> core_nbase, core_ncount;     // global slot base per core
    std::vector<uint32_t> slot_of_gid;                 // neuron (desc order) -> global slot
    std::vector<int64_t> gid_of_slot;                  // global slot -> neuron, -1 for padding
    std::vector<uint32_t> core_of_slot;

    // ---- rank window ----
    uint32_t first_core{0}, last_core{0};              // [first, last) cores held by this rank
    uint32_t slot_offset{0}, n_slots{0};
    std::vector<uint32_t> rank_slot_begin;             // [n_ranks + 1] first global slot of every rank's window

    // ---- image arrays (local to the rank) ----
    std::vector<uint32_t> l_core_nbase, l_core_ncount;
    std::vector<double> core_axon_out_latency, core_axon_in_latency;
    std::vector<sanafe_hip_soma_class> soma_classes;
    std::vector<sanafe_hip_cost_class> cost_classes;
    std::vector<uint32_t> slot_cls, slot_aux, slot_packets, slot_hops, slot_events;
    std::vector<double> slot_bias, slot_v0, slot_e_net, slot_e_syn, slot_e_dend;
    std::vector<uint32_t> in_train_beg, in_train_len, in_train_bits;
    std::vector<int64_t> in_rate_period;
    std::vector<uint8_t> in_shared;    // per input neuron: its `input` unit instance holds other neurons too
    std::vector<uint32_t> in_seed;     // per input neuron: the std::mt19937 seed of its unit instance (src/models.hpp:347) ...
    std::vector<uint64_t> in_unit_key; // ... and the unit's key (ExtColumn::unit_key): a Poisson rate set after load() needs them
    // ---- external per-step value streams (include/sanafe_hip.h: slot_ext, sanafe_hip_write_ext) ----
    // One column per neuron that consumes a sequential host-side source at every update.
    struct ExtColumn
    {
        enum Kind : uint8_t { Poisson = 1, TrueNorthRand = 2, LifNoise = 3 };
        uint32_t slot{0};          // local slot
        uint8_t kind{0};
        double poisson{0.0};       // InputModel::poisson_probability
        uint32_t seed{0};          // std::mt19937 seed of the unit instance (src/models.hpp:347)
        uint32_t gen{0};           // which unit instance (neurons sharing an input unit share its generator)
        uint32_t mask{0};          // TrueNorth random_range_mask
        uint64_t rand_index{0};    // position of the neuron among ALL rand()-consuming neurons of the chip
        uint32_t stream{0};        // LIF noise: index into noise_streams
        uint64_t unit_key{0};      // Poisson: (core << 16 | soma unit) of the unit instance -- the same across lowerings of one chip
        int64_t skip_updates{0};   // Poisson column added after load(): updates the unit had made by then -- its generator draws at
                                   // every update whatever the rate (src/models.cpp:876), so a fresh one skips as many draws
    };
    struct NoiseStream // one per (core, LIF unit with a `noise` file): every instance opens its own stream
    {
        std::string path;
        long random_mask{0x7f}, sign_mask{0x100}; // src/models.hpp:271-272, src/models.cpp:367-371
        uint64_t unit_key{0};                     // (core << 16 | soma unit)
    };
    // ---- `taps` dendrites (MultiTapModel1D, src/models.cpp:167-348): one entry per neuron behind such a unit ----
    std::vector<uint32_t> tap_slot, tap_count; // local slot, number of taps (<= 8)
    std::vector<double> tap_tc, tap_sc;        // [n][8] time constants, [n][8] space constants (first taps-1 used)
    std::vector<ExtColumn> ext;        // in column order == ascending slot
    std::vector<uint32_t> slot_ext;    // [n_slots] column or 0xffffffff; empty when there is no column
    std::vector<NoiseStream> noise_streams;
    uint64_t n_rand_global{0};         // rand() calls per timestep on the whole chip
    uint32_t ring_slots{1};
    std::vector<uint32_t> slice_core;
    std::vector<uint64_t> slice_axon_beg, slice_axon_end, core_syn_base;
    std::vector<uint32_t> ax_pre, ax_syn_beg, ax_nsyn;
    std::vector<double> ax_proc_delay;
    std::vector<uint8_t> ax_lat_class;
    std::vector<double> lat_class_per_event;
    std::vector<uint32_t> syn_meta;
    std::vector<double> syn_weight;

    // ---- host-side tables for message reconstruction (detailed timing, message trace) ----
    // inbound axons (global over all destination cores of THIS rank) carry their destination;
    // out_ptr/out_axon list, per GLOBAL slot, the axons a spike of that neuron activates,
    // in the reference's message order (ascending destination core).
    std::vector<uint32_t> ax_dest_core;     // global core id
    std::vector<uint32_t> ax_dest_axon_id;  // index inside the destination core's axons_in
    std::vector<uint32_t> ax_hops;
    std::vector<double> ax_min_hop_delay;
    std::vector<uint64_t> out_ptr;          // [n_global_slots + 1]
    std::vector<uint64_t> out_axon;

    // ---- per-slot host info ----
    std::vector<uint8_t> slot_log_spikes, slot_log_potential;
    std::vector<uint8_t> slot_model;        // SANAFE_SOMA_* (local slots)

    // ---- host-evaluated (plugin) soma units: one instance per (core, unit), src/core.cpp:196-231 ----
    struct HostUnit
    {
        uint32_t core{0};
        int desc_unit{0};          // index into sanafe_desc::unit_* (attributes, model, plugin path)
        std::string name, model, plugin_path;
        bool has_energy{false}, has_latency{false};
        double energy[3]{}, latency[3]{}; // idle / updated / fired sums of the architecture defaults
    };
    struct HostNeuron
    {
        uint32_t slot{0}, core_local{0}, unit{0}, addr{0};
        int64_t gid{0};
    };
    std::vector<HostUnit> host_units;
    std::vector<HostNeuron> host_neurons;

    // ---- host cores: cores whose pipeline cannot run on the device by construction -- the soma unit is called once per
    //      synaptic EVENT (`buffer_position: soma` inside the unit, or `axon_out`; src/mapped.cpp:27-58, 168-188), or a
    //      synapse / dendrite unit is a plugin (a host C++ object, src/plugins.cpp:45-98).  Their neurons are
    //      SANAFE_SOMA_HOST slots on the device and their inbound axons are NOT part of the device image: the host library
    //      replays such a core's neuron and message pipelines per timestep from the chip's spike bitmap (host/host_cores.cpp).
    struct HostCore
    {
        struct Unit
        {
            int desc_unit{0};          // index into sanafe_desc::unit_*
            std::string name, model, plugin_path; // plugin_path empty: a built-in model
            bool syn{false}, dend{false}, soma{false}, update_every_timestep{false};
            std::optional<double> e_spike, l_spike, e_update, l_update; // architecture defaults (src/pipeline.cpp:177-266)
            bool has_soma_e{false}, has_soma_l{false};
            double se[3]{}, sl[3]{};   // access, update, spike_out
        };
        struct Neuron
        {
            uint32_t slot{0};          // local slot
            int64_t gid{0};
            int32_t soma_unit{0}, dend_unit{0};
            uint32_t soma_addr{0}, dend_addr{0}; // per-unit addresses: arrival (mapping) order on the unit
        };
        struct Synapse
        {
            int32_t unit{0};           // synapse unit
            uint32_t addr{0};          // address on that unit: arrival order in map_connections order
            uint32_t post{0};          // post neuron, offset within the core
            int64_t edge{0};           // index into sanafe_desc::edge_*
            double weight{0.0};        // the edge's weight (cores that run on the device: sanafe_hip_image::msg_syn_weight)
            bool pre_checks_synapses{false}; // the SOURCE neuron itself receives through a synapse unit flagged
                                             // update_every_timestep: forced_updates walks its connections (src/chip.cpp:989-1005)
        };
        struct Axon
        {
            uint32_t pre{0};           // GLOBAL slot of the source neuron
            uint32_t syn_beg{0}, n_syn{0};
        };
        uint32_t core{0};              // global core id (== local: host cores need a single-rank chip)
        int bp{0};                     // SANAFE_BUF_*
        double ain_latency{0.0};
        std::vector<Unit> units;       // template order
        std::vector<Neuron> neurons;   // mapped order (offset within the core)
        std::vector<Synapse> synapses; // delivery order: axon by axon
        std::vector<Axon> axons;       // delivery order (source core, source neuron): src/chip.cpp:661-690
    };
    std::vector<HostCore> host_cores;
    // Buffer inside the soma unit / before axon_out with built-in units only (`current_based`, `accumulator`, `truenorth`, one
    // unit of each role): these cores run on the DEVICE (sanafe_hip_image::msg_*, msgsoma_kernel); host_cores then only carries
    // their tables to the image and no host replay object is created.  All such cores of a chip or none (SANAFE_HOST_CORES=1
    // keeps them on the host).
    bool msg_on_device{false};
    // ... their inbound axons for the HOST's message reconstruction (detailed timing, message trace): axon ids from
    // ax_pre.size() on in ax_dest_core / ax_dest_axon_id / ax_hops / ax_min_hop_delay and in out_axon (a source neuron's
    // axons stay in ascending destination-core order); id - ax_pre.size() indexes msg_ax_* and the per-step fired counts
    uint64_t n_device_axons{0};
    std::vector<uint32_t> msg_core, msg_ax_beg, msg_ax_pre, msg_ax_nsyn, msg_syn_beg, msg_syn_post;
    std::vector<double> msg_syn_weight;
    std::vector<sanafe_hip_msg_core_costs> msg_costs;
    std::vector<std::array<uint32_t, 3>> msg_units; // per such core: index of its synapse, dendrite and soma unit (optional perf columns)

    // ---- optional perf-trace columns: tiles / cores with log_energy, units with log_energy / log_latency
    //      (sim_trace_get_optional_traces, src/chip.cpp:1541-1579).  Filled only when some flag is set. ----
    struct LogPlan
    {
        bool any{false};
        struct Column
        {
            std::string name;
            uint8_t kind{0}; // 0 tile energy, 1 core energy, 2 unit energy, 3 unit latency
            uint32_t tile{0}, core{0}, unit{0};
        };
        std::vector<Column> columns;               // std::map order (lexicographic by name)
        std::vector<uint32_t> core_unit_beg;       // [n_cores + 1] first entry of each core in the unit_* arrays
        std::vector<double> unit_e_spike, unit_e_update; // per (core, unit): energy per synaptic event / per dendrite update
        std::vector<uint8_t> unit_used;            // PipelineUnit::is_used
        std::vector<uint8_t> slot_soma_unit, slot_dend_unit; // per local slot: unit index inside its core
        std::vector<uint16_t> syn_units;           // per local synapse: synapse unit | dendrite unit << 8
        std::vector<double> ax_e_hop;              // per local axon: hop energy charged to the destination tile
        std::vector<double> core_e_ain, core_e_aout; // per core: energy per message in / out (0 unless exactly one axon unit)
        std::vector<int> core_bp;                  // buffer position of each core
    };
    LogPlan log;

    // neuron groups (for trace ordering: lexicographic group name, offset)
    std::vector<std::string> group_names;
    std::vector<int64_t> group_ptr;
    std::vector<int> group_lex_order;

    sanafe_hip_image image() const;
};

// Maps `desc` and lowers the part owned by `rank` of `n_ranks` (tiles are split in
// contiguous blocks).  Throws std::exception subclasses on any error.
void map_and_lower(const sanafe_desc &desc, int n_ranks, int rank, uint32_t target_slices, uint32_t min_slice_axons, MappedChip &out);
} // namespace sanafe_amd

#endif
