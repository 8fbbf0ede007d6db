// description.hpp -- C++17 description objects of the MI355X host: the `Architecture` and
// `SpikingNetwork` a user builds through the API or loads from YAML, and their lowering to the
// flat `sanafe_desc` (include/sanafe_desc.h) that SpikingChip::load() consumes.
//
// Public surface mirrors the reference: Architecture::create_tile / create_core and the
// per-core unit builders (src/arch.cpp:41-180, src/yaml_arch.cpp:149-186),
// SpikingNetwork::create_neuron_group, NeuronGroup::connect_neurons_dense / _sparse / _conv2d,
// Neuron::map_to_core / set_attributes / connect_to_neuron (src/network.cpp:62-605), load_arch /
// load_net (src/arch.cpp:106-117, src/network.cpp:194-222).  Storage is columnar: no per-neuron
// or per-edge objects, so 10^6 neurons / 10^9 edges stay a handful of vectors.
#ifndef SANAFE_AMD_DESCRIPTION_HPP
#define SANAFE_AMD_DESCRIPTION_HPP

#include <array>
#include <deque>
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sanafe_desc.h"

namespace sanafe_amd
{
// One typed attribute value: the columnar twin of sanafe::ModelAttribute (src/attribute.hpp:41-176).
struct AttrValue
{
    int type{SANAFE_ATTR_DOUBLE};
    double num{0.0};
    std::string str;
    std::vector<double> list;
    static AttrValue Int(long v)
    {
        AttrValue a;
        a.type = SANAFE_ATTR_INT;
        a.num = static_cast<double>(v);
        return a;
    }
    static AttrValue Double(double v)
    {
        AttrValue a;
        a.type = SANAFE_ATTR_DOUBLE;
        a.num = v;
        return a;
    }
    static AttrValue Bool(bool v)
    {
        AttrValue a;
        a.type = SANAFE_ATTR_BOOL;
        a.num = v ? 1.0 : 0.0;
        return a;
    }
    static AttrValue String(std::string s)
    {
        AttrValue a;
        a.type = SANAFE_ATTR_STRING;
        a.str = std::move(s);
        return a;
    }
    static AttrValue List(std::vector<double> l)
    {
        AttrValue a;
        a.type = SANAFE_ATTR_LIST;
        a.list = std::move(l);
        return a;
    }
    bool operator==(const AttrValue &o) const { return type == o.type && num == o.num && str == o.str && list == o.list; }
};
// YAML scalar typing order int -> double -> bool -> string (src/yaml_common.cpp:205-262)
AttrValue scalar_attr(const std::string &text);

struct UnitConfig // PipelineUnitConfiguration + ModelInfo (src/arch.hpp:51-59, 178-192)
{
    std::string name, model, plugin;
    int implements{0}; // SANAFE_IMPL_*
    bool log_energy{false}, log_latency{false}, update_every_timestep{false};
    std::map<std::string, AttrValue> attributes;
};

// The axon units and pipeline units of a core; cores replicated from one description entry
// (`name[a..b]`) share a template, every core still gets its own unit instances on the chip.
struct CoreTemplate
{
    std::vector<std::array<double, 2>> axon_in, axon_out; // {energy, latency}
    std::vector<UnitConfig> units;
};

class CoreConfig // CoreConfiguration (src/arch.hpp:154-169)
{
public:
    std::string name;
    int parent_tile_id{0}, offset_within_tile{0}, id{0};
    int buffer_position{SANAFE_BUF_BEFORE_SOMA};
    int64_t max_neurons_supported{1024};
    bool log_energy{false};
    std::shared_ptr<CoreTemplate> tmpl;

    void create_axon_in(const std::string &unit_name, double energy_message_in, double latency_message_in);
    void create_axon_out(const std::string &unit_name, double energy_message_out, double latency_message_out);
    // yaml_merge_or_create_hardware_unit (src/yaml_arch.cpp:149-186); section: "synapse" | "dendrite" | "soma"
    UnitConfig &create_unit(const std::string &section, const std::string &unit_name, const std::string &model,
            const std::map<std::string, AttrValue> &attributes, const std::string &plugin = "", bool log_energy = false,
            bool log_latency = false, bool update_every_timestep = false);
};

struct TileConfig // TileConfiguration + TilePowerMetrics (src/arch.hpp:131-152)
{
    std::string name;
    int id{0};
    std::array<double, 4> hop_energy{}, hop_latency{}; // N, E, S, W
    bool log_energy{false};
    std::vector<int> cores; // global core ids
};

int parse_buffer_position(const std::string &s, bool inside); // src/pipeline.cpp:268-310

class Architecture // src/arch.hpp:70-101
{
public:
    std::string name;
    int noc_width{1}, noc_height{1}, noc_buffer_size{0};
    std::map<int64_t, double> sync_table{{0, 0.0}};
    std::deque<TileConfig> tiles; // deques: references handed to Python stay valid while the chip grows
    std::deque<CoreConfig> cores;

    Architecture() = default;
    Architecture(std::string name, int width, int height, int link_buffer_size, std::map<int64_t, double> sync = {{0, 0.0}});
    TileConfig &create_tile(const std::string &tile_name, const std::array<double, 4> &hop_energy = {},
            const std::array<double, 4> &hop_latency = {}, bool log_energy = false);
    CoreConfig &create_core(const std::string &core_name, int parent_tile_id, int buffer_position = SANAFE_BUF_BEFORE_SOMA,
            int64_t max_neurons_supported = 1024, bool log_energy = false, std::shared_ptr<CoreTemplate> share = nullptr);
    size_t core_count() const { return cores.size(); }
};

class SpikingNetwork;

class NeuronGroup // src/network.hpp:111-146, columnar
{
public:
    struct Column
    {
        std::vector<uint8_t> mask, type, fwd;
        std::vector<double> num;
        std::vector<int32_t> str;
    };
    SpikingNetwork *net{nullptr};
    std::string name;
    int64_t count{0}, base{0};
    std::vector<int32_t> core, soma_hw, dendrite_hw, synapse_hw;
    std::vector<int64_t> map_order;
    std::vector<uint8_t> log_spikes, log_potential;
    std::map<std::string, Column> columns;
    std::map<std::pair<std::string, int64_t>, std::vector<double>> list_values;

    NeuronGroup(SpikingNetwork *net, std::string name, int64_t count, int64_t base);
    void set_attribute(const std::string &key, const AttrValue &value, int fwd, int64_t lo, int64_t hi);
    void set_attribute_column(const std::string &key, const double *values, int attr_type, int fwd = 7);
    // Neuron::set_attributes on a range (src/network.cpp:94-128); empty optional == leave unchanged
    void apply_config(int64_t lo, int64_t hi, const std::optional<std::string> &soma_hw_name,
            const std::optional<std::string> &default_synapse_hw_name, const std::optional<std::string> &dendrite_hw_name,
            const std::optional<bool> &set_log_spikes, const std::optional<bool> &set_log_potential,
            const std::map<std::string, std::pair<AttrValue, int>> &attributes);
    void map_to_core(const CoreConfig &core_config, int64_t lo, int64_t hi); // src/network.cpp:85-92
    // hyper-edges (src/network.cpp:229-605); weights/delays are per-edge lists, delay may be empty
    void connect_neurons_sparse(NeuronGroup &dest, const std::vector<std::pair<int64_t, int64_t>> &pairs,
            const std::vector<double> &weight, const std::vector<int> &delay);
    void connect_neurons_dense(NeuronGroup &dest, const std::vector<double> &weight, const std::vector<int> &delay);
    void connect_neurons_conv2d(NeuronGroup &dest, const std::vector<double> &weight, const std::vector<int> &delay,
            int input_width, int input_height, int input_channels, int kernel_width, int kernel_height, int kernel_count = 1,
            int stride_width = 1, int stride_height = 1);
};

class SpikingNetwork // src/network.hpp:148-176
{
public:
    std::string name;
    std::vector<std::unique_ptr<NeuronGroup>> order;         // creation order
    std::map<std::string, NeuronGroup *> groups;             // by name (lexicographic, like std::map in the reference)
    int64_t neuron_count{0}, mapping_count{0};
    // edges in creation order
    std::vector<int64_t> edge_src, edge_dst;
    std::vector<double> edge_weight;
    std::vector<int8_t> edge_delay; // -1 = none; stays empty until a delay is seen
    std::vector<int32_t> edge_synapse_hw;
    // string table shared with the architecture at lowering time
    std::vector<std::string> strings;
    std::map<std::string, int32_t> string_ids;

    explicit SpikingNetwork(std::string net_name = "") : name(std::move(net_name)) {}
    SpikingNetwork(const SpikingNetwork &) = delete;
    SpikingNetwork &operator=(const SpikingNetwork &) = delete;
    int32_t intern(const std::string &s);
    NeuronGroup &create_neuron_group(const std::string &group_name, int64_t neuron_count_,
            const std::map<std::string, std::pair<AttrValue, int>> &attributes = {}, const std::string &default_synapse_hw_name = "",
            const std::string &default_dendrite_hw_name = "", bool log_potential = false, bool log_spikes = false,
            const std::string &soma_hw_name = "");
    void add_edge(int64_t src_gid, int64_t dst_gid, double weight, int delay, int32_t synapse_hw);
    // bulk, zero-copy-ish append for synthetic networks (arrays are copied once into the columns)
    void add_edges(const int64_t *src, const int64_t *dst, const double *weight, const int8_t *delay, int32_t synapse_hw,
            int64_t n);
    int64_t edge_count() const { return static_cast<int64_t>(edge_src.size()); }
    // Appends a copy of every group, mapping and edge of `other` (neuron ids and mapping order continue after this
    // network's): what a chip holds after SpikingChip::load(other, overwrite = false), src/chip.cpp:129-138.
    void absorb(const SpikingNetwork &other);
};

// Owns every buffer behind a sanafe_desc.
struct BuiltDesc
{
    sanafe_desc desc{};
    std::vector<std::vector<int32_t>> i32;
    std::vector<std::vector<int64_t>> i64;
    std::vector<std::vector<uint8_t>> u8;
    std::vector<std::vector<int8_t>> i8;
    std::vector<std::vector<double>> f64;
    std::vector<std::string> strings;
    std::vector<const char *> string_ptrs;
};
std::unique_ptr<BuiltDesc> to_desc(const Architecture &arch, SpikingNetwork &net);

Architecture load_arch(const std::string &path);                               // src/arch.cpp:106-117
std::unique_ptr<SpikingNetwork> load_net(const std::string &path, Architecture &arch); // src/network.cpp:194-222
} // namespace sanafe_amd
#endif
