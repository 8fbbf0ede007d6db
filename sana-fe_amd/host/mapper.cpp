// mapper.cpp -- see mapper.hpp.
#include "mapper.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <numeric>
#include <exception>
#include <optional>
#include <thread>
#include <tuple>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace sanafe_amd
{
namespace
{
struct Attr
{
    const sanafe_attr_table *t;
    int64_t i;
    const sanafe_desc *d;
    std::string key() const { return str(t->key[i]); }
    std::string str(int32_t id) const { return id < 0 ? std::string() : std::string(d->strings[id]); }
    int type() const { return t->type[i]; }
    int fwd() const { return t->fwd ? t->fwd[i] : 7; }
    // ModelAttribute conversion operators, src/attribute.hpp:43-93
    double as_double() const
    {
        if (type() == SANAFE_ATTR_DOUBLE || type() == SANAFE_ATTR_INT) return t->num[i];
        throw std::runtime_error("Error: Attribute " + key() + " cannot be cast to a double");
    }
    int as_int() const
    {
        if (type() != SANAFE_ATTR_INT) throw std::runtime_error("Error: Attribute " + key() + " is not an integer");
        return static_cast<int>(t->num[i]);
    }
    bool as_bool() const
    {
        if (type() == SANAFE_ATTR_BOOL || type() == SANAFE_ATTR_INT) return t->num[i] != 0.0;
        throw std::runtime_error("Error: Attribute " + key() + " cannot be cast to a bool ()");
    }
    std::string as_string() const
    {
        if (type() != SANAFE_ATTR_STRING) throw std::runtime_error("Error: Attribute " + key() + " is not a string");
        return str(t->str[i]);
    }
};

uint8_t parse_reset_mode(const std::string &s) // src/models.cpp:905-931
{
    if (s == "none") return SANAFE_RESET_NONE;
    if (s == "soft") return SANAFE_RESET_SOFT;
    if (s == "hard") return SANAFE_RESET_HARD;
    if (s == "saturate") return SANAFE_RESET_SATURATE;
    throw std::invalid_argument("Reset mode not recognized");
}

} // namespace

double SomaAttr::as_double() const
{
    if (type == SANAFE_ATTR_DOUBLE || type == SANAFE_ATTR_INT) return num;
    throw std::runtime_error("Error: Attribute " + key + " cannot be cast to a double");
}
int SomaAttr::as_int() const
{
    if (type != SANAFE_ATTR_INT) throw std::runtime_error("Error: Attribute " + key + " is not an integer");
    return static_cast<int>(num);
}
bool SomaAttr::as_bool() const
{
    if (type == SANAFE_ATTR_BOOL || type == SANAFE_ATTR_INT) return num != 0.0;
    throw std::runtime_error("Error: Attribute " + key + " cannot be cast to a bool ()");
}
const std::string &SomaAttr::as_string() const
{
    if (type != SANAFE_ATTR_STRING) throw std::runtime_error("Error: Attribute " + key + " is not a string");
    return str;
}

void apply_soma_attribute(uint32_t model, const SomaAttr &a, sanafe_hip_soma_class &p, SomaAttrEffect &fx)
{
    const std::string &k = a.key;
    if (model == SANAFE_SOMA_LIF)
    {
        if (k == "threshold") p.threshold = a.as_double();
        else if (k == "reverse_threshold") p.reverse_threshold = a.as_double();
        else if (k == "reset") p.reset = a.as_double();
        else if (k == "reverse_reset") p.reverse_reset = a.as_double();
        else if (k == "reset_mode") p.reset_mode = parse_reset_mode(a.as_string());
        else if (k == "reverse_reset_mode") p.reverse_reset_mode = parse_reset_mode(a.as_string());
        else if (k == "leak_decay") p.leak_decay = a.as_double();
        else if (k == "log_u") (void) a.as_bool();
        else if (k == "input_decay") p.input_decay = a.as_double();
        else if (k == "bias") fx.bias = a.as_double(), fx.bias_set = true;
        else if (k == "force_update" || k == "force_update_every_timestep") p.force_update = a.as_bool();
        else if (k == "refractory_delay") p.refractory_delay = a.as_int();
        else if (k == "potential") fx.potential = a.as_double(), fx.potential_set = true;
    }
    else if (model == SANAFE_SOMA_TRUENORTH)
    {
        if (k == "threshold") p.threshold = a.as_double();
        else if (k == "reverse_threshold") p.reverse_threshold = a.as_double();
        else if (k == "reset") p.reset = a.as_double();
        else if (k == "reverse_reset") p.reverse_reset = a.as_double();
        else if (k == "reset_mode") p.reset_mode = parse_reset_mode(a.as_string());
        else if (k == "reverse_reset_mode") p.reverse_reset_mode = parse_reset_mode(a.as_string());
        else if (k == "leak") p.leak_decay = a.as_double();
        else if (k == "bias") fx.bias = a.as_double(), fx.bias_set = true;
        else if (k == "force_update_every_timestep" || k == "force_update") p.force_update = a.as_bool();
        else if (k == "leak_towards_zero") p.leak_towards_zero = a.as_bool();
        else if (k == "random_mask")
        {
            const int m = a.as_int();
            if (m < 0) throw std::invalid_argument("random_mask < 0; must be unsigned.");
            fx.random_mask = static_cast<uint32_t>(m);
            fx.random_mask_set = true;
        }
    }
}

sanafe_hip_soma_class canonical_soma_class(const sanafe_hip_soma_class &p)
{
    sanafe_hip_soma_class canon;
    std::memset(&canon, 0, sizeof(canon));
    canon.threshold = p.threshold;
    canon.reverse_threshold = p.reverse_threshold;
    canon.reset = p.reset;
    canon.reverse_reset = p.reverse_reset;
    canon.leak_decay = p.leak_decay;
    canon.input_decay = p.input_decay;
    canon.refractory_delay = p.refractory_delay;
    canon.reset_mode = p.reset_mode;
    canon.reverse_reset_mode = p.reverse_reset_mode;
    canon.force_update = p.force_update;
    canon.leak_towards_zero = p.leak_towards_zero;
    return canon;
}

namespace
{
enum Model { M_CURRENT_BASED, M_ACCUMULATOR, M_ACC_DELAY, M_TAPS, M_INPUT, M_LIF, M_TRUENORTH, M_PLUGIN };

struct UnitInfo // one pipeline unit of a core template
{
    std::string name;
    Model model{M_PLUGIN};
    bool syn{false}, dend{false}, soma{false};
    bool update_every_timestep{false}, log{false}, log_energy{false}, log_latency{false};
    std::optional<double> e_spike, l_spike, e_update, l_update;
    bool has_soma_e{false}, has_soma_l{false};
    double se[3]{}, sl[3]{}; // access, update, spike_out
    size_t capacity{SIZE_MAX};
    std::string noise_path; // LIF `noise` file (src/models.cpp:354-366)
    long noise_random_mask{0x7f};
    int input_rank{0};      // `input` units before this one in the template (seed order, src/models.hpp:347)
};

struct Template
{
    std::vector<UnitInfo> units;
    std::vector<double> ain_e, ain_l, aout_e, aout_l;
};

std::string S(const sanafe_desc &d, int32_t id) { return id < 0 ? std::string() : std::string(d.strings[id]); }

Template read_template(const sanafe_desc &d, int tm)
{
    Template t;
    for (int i = d.tmpl_axon_in_ptr[tm]; i < d.tmpl_axon_in_ptr[tm + 1]; i++)
    {
        t.ain_e.push_back(d.axon_in_energy[i]);
        t.ain_l.push_back(d.axon_in_latency[i]);
    }
    for (int i = d.tmpl_axon_out_ptr[tm]; i < d.tmpl_axon_out_ptr[tm + 1]; i++)
    {
        t.aout_e.push_back(d.axon_out_energy[i]);
        t.aout_l.push_back(d.axon_out_latency[i]);
    }
    for (int u = d.tmpl_unit_ptr[tm]; u < d.tmpl_unit_ptr[tm + 1]; u++)
    {
        UnitInfo ui;
        ui.name = S(d, d.unit_name[u]);
        const std::string model = S(d, d.unit_model[u]);
        ui.syn = d.unit_implements[u] & SANAFE_IMPL_SYNAPSE;
        ui.dend = d.unit_implements[u] & SANAFE_IMPL_DENDRITE;
        ui.soma = d.unit_implements[u] & SANAFE_IMPL_SOMA;
        ui.update_every_timestep = d.unit_flags[u] & SANAFE_UNIT_UPDATE_EVERY_TIMESTEP;
        ui.log = d.unit_flags[u] & (SANAFE_UNIT_LOG_ENERGY | SANAFE_UNIT_LOG_LATENCY);
        ui.log_energy = d.unit_flags[u] & SANAFE_UNIT_LOG_ENERGY;
        ui.log_latency = d.unit_flags[u] & SANAFE_UNIT_LOG_LATENCY;
        bool want_syn = false, want_dend = false, want_soma = false;
        if (d.unit_plugin[u] >= 0) ui.model = M_PLUGIN;
        else if (model == "current_based") ui.model = M_CURRENT_BASED, want_syn = true;
        else if (model == "accumulator") ui.model = M_ACCUMULATOR, want_dend = true, ui.capacity = 1024;
        else if (model == "accumulator_with_delay") ui.model = M_ACC_DELAY, want_dend = true, ui.capacity = 1024;
        else if (model == "taps") ui.model = M_TAPS, want_dend = true; // one RC line per unit; see shared_taps in map_and_lower
        else if (model == "input") ui.model = M_INPUT, want_soma = true; // no bound: neurons on one unit SHARE its state
        else if (model == "leaky_integrate_fire") ui.model = M_LIF, want_soma = true, ui.capacity = 1024;
        else if (model == "truenorth") ui.model = M_TRUENORTH, want_soma = true, ui.capacity = 4096;
        else throw std::invalid_argument("Pipeline model not supported (" + model + ")\n"); // src/models.cpp:964-966
        if (ui.model != M_PLUGIN && (want_syn != ui.syn || want_dend != ui.dend || want_soma != ui.soma))
            throw std::runtime_error("Unit '" + ui.name + "' (" + model +
                    ") is listed in a hardware section it does not implement"); // check_implemented, src/pipeline.cpp:20-57
        // PipelineUnit::set_attributes_hw, src/pipeline.cpp:151-266
        std::map<std::string, Attr> m;
        for (int64_t i = d.unit_attr_ptr[u]; i < d.unit_attr_ptr[u + 1]; i++)
        {
            Attr a{&d.unit_attrs, i, &d};
            m.emplace(a.key(), a);
        }
        auto has = [&](const char *k) { return m.count(k) > 0; };
        auto num = [&](const char *k) { return m.at(k).as_double(); };
        if (has("energy_process_spike")) ui.e_spike = num("energy_process_spike");
        if (has("latency_process_spike")) ui.l_spike = num("latency_process_spike");
        if (has("energy_update")) ui.e_update = num("energy_update");
        if (has("latency_update")) ui.l_update = num("latency_update");
        const char *en[3] = {"energy_access_neuron", "energy_update_neuron", "energy_spike_out"};
        if (has(en[0]) || has(en[1]) || has(en[2]))
        {
            for (int k = 0; k < 3; k++)
            {
                if (!has(en[k])) throw std::invalid_argument(std::string("Metric not defined: ") + en[k]);
                ui.se[k] = num(en[k]);
            }
            ui.has_soma_e = true;
        }
        const char *ln[3] = {"latency_access_neuron", "latency_update_neuron", "latency_spike_out"};
        if (has(ln[0]) || has(ln[1]) || has(ln[2]))
        {
            for (int k = 0; k < 3; k++)
            {
                if (!has(ln[k])) throw std::invalid_argument(std::string("Missing metric: ") + ln[k]);
                ui.sl[k] = num(ln[k]);
            }
            ui.has_soma_l = true;
        }
        if (ui.model == M_LIF)
        {
            // set_attribute_hw runs in attribute (key) order, src/models.cpp:351-373
            if (has("noise")) ui.noise_path = m.at("noise").as_string();
            if (has("noise_bits")) ui.noise_random_mask = (1L << m.at("noise_bits").as_int()) - 1L;
        }
        if (ui.model == M_INPUT)
        {
            for (const UnitInfo &prev : t.units) ui.input_rank += (prev.model == M_INPUT);
        }
        t.units.push_back(std::move(ui));
    }
    return t;
}

// Core::get_hw, src/core.cpp:61-97
int find_unit(const Template &t, const std::string &name, bool syn, bool dend, bool soma)
{
    for (size_t i = 0; i < t.units.size(); i++)
    {
        const UnitInfo &u = t.units[i];
        if ((syn && !u.syn) || (dend && !u.dend) || (soma && !u.soma)) continue;
        if (name.empty() || name == u.name) return static_cast<int>(i);
    }
    throw HardwareMappingError("Could not find h/w (with name:" + name + ") that implements synapse:" +
            std::to_string(int(syn)) + ", dendrite:" + std::to_string(int(dend)) + ", soma:" + std::to_string(int(soma)));
}

// Runs fn(t) for t in [0, T) on T threads; the exception of the lowest task wins, which for
// contiguous blocks is the error a serial scan would have hit first.
template <typename F> void parallel_tasks(int T, const F &fn)
{
    if (T <= 1)
    {
        fn(0);
        return;
    }
    std::vector<std::exception_ptr> errors(T);
    std::vector<std::thread> pool;
    for (int t = 0; t < T; t++)
        pool.emplace_back([&, t] {
            try
            {
                fn(t);
            }
            catch (...)
            {
                errors[t] = std::current_exception();
            }
        });
    for (std::thread &th : pool) th.join();
    for (const std::exception_ptr &e : errors)
        if (e) std::rethrow_exception(e);
}

int block_count(int n_threads, uint64_t n) { return static_cast<int>(std::max<uint64_t>(1, std::min<uint64_t>(n_threads, n / 65536 + 1))); }

// Stable counting sort of `order` by key_of(order[i]); blocks of the input are counted and scattered
// by separate threads, block b's elements of one key landing before block b+1's.
template <typename I, typename K>
void counting_sort(std::vector<I> &order, const K &key_of, size_t n_keys, std::vector<I> &tmp, int n_threads)
{
    const uint64_t n = order.size();
    int T = block_count(n_threads, n);
    while (T > 1 && static_cast<uint64_t>(T) * n_keys > n / 2 + (uint64_t{1} << 26)) T--; // histogram memory stays modest
    std::vector<std::vector<uint64_t>> count(T);
    tmp.resize(n);
    auto range = [&](int t) { return std::pair<uint64_t, uint64_t>(n * t / T, n * (t + 1) / T); };
    parallel_tasks(T, [&](int t) {
        std::vector<uint64_t> &c = count[t];
        c.assign(n_keys, 0);
        const auto r = range(t);
        for (uint64_t i = r.first; i < r.second; i++) c[key_of(order[i])]++;
    });
    uint64_t run = 0;
    for (size_t k = 0; k < n_keys; k++)
        for (int t = 0; t < T; t++)
        {
            const uint64_t c = count[t][k];
            count[t][k] = run;
            run += c;
        }
    parallel_tasks(T, [&](int t) {
        std::vector<uint64_t> &c = count[t];
        const auto r = range(t);
        for (uint64_t i = r.first; i < r.second; i++)
        {
            const I e = order[i];
            tmp[c[key_of(e)]++] = e;
        }
    });
    order.swap(tmp);
}

// A permutation of the edges: 4-byte indices unless there are 2^32 edges or more (a 687 M-edge network then
// sorts 5.5 GB less than with 8-byte indices).
struct EdgeOrder
{
    bool wide{false};
    std::vector<uint32_t> a32, t32;
    std::vector<uint64_t> a64, t64;
    explicit EdgeOrder(uint64_t n) : wide(n > 0xffffffffull)
    {
        if (wide)
        {
            a64.resize(n);
            std::iota(a64.begin(), a64.end(), uint64_t{0});
        }
        else
        {
            a32.resize(n);
            std::iota(a32.begin(), a32.end(), uint32_t{0});
        }
    }
    uint64_t operator[](int64_t i) const { return wide ? a64[i] : a32[i]; }
    uint64_t size() const { return wide ? a64.size() : a32.size(); }
    template <typename K> void sort(const K &key_of, size_t n_keys, int n_threads)
    {
        if (wide) counting_sort(a64, key_of, n_keys, t64, n_threads);
        else counting_sort(a32, key_of, n_keys, t32, n_threads);
    }
    void release_scratch()
    {
        std::vector<uint32_t>().swap(t32);
        std::vector<uint64_t>().swap(t64);
    }
};

struct ClassKey
{
    unsigned char b[sizeof(sanafe_hip_soma_class)];
    bool operator<(const ClassKey &o) const { return std::memcmp(b, o.b, sizeof(b)) < 0; }
};
struct CostKey
{
    unsigned char b[sizeof(sanafe_hip_cost_class)];
    bool operator<(const CostKey &o) const { return std::memcmp(b, o.b, sizeof(b)) < 0; }
};
} // namespace

sanafe_hip_image MappedChip::image() const
{
    sanafe_hip_image im{};
    im.n_cores = static_cast<uint32_t>(l_core_nbase.size());
    im.n_slots = n_slots;
    im.n_soma_classes = static_cast<uint32_t>(soma_classes.size());
    im.n_cost_classes = static_cast<uint32_t>(cost_classes.size());
    im.ring_slots = ring_slots;
    im.n_slices = static_cast<uint32_t>(slice_core.size());
    im.n_axons = ax_pre.size();
    im.n_synapses = syn_meta.size();
    im.n_input = static_cast<uint32_t>(in_train_beg.size());
    im.n_train_words = in_train_bits.size();
    im.slot_offset = slot_offset;
    im.n_global_slots = n_global_slots;
    im.sync_delay = sync_delay;
    im.core_nbase = l_core_nbase.data();
    im.core_ncount = l_core_ncount.data();
    im.core_axon_out_latency = core_axon_out_latency.data();
    im.soma_classes = soma_classes.data();
    im.cost_classes = cost_classes.data();
    im.slot_cls = slot_cls.data();
    im.slot_bias = slot_bias.data();
    im.slot_v0 = slot_v0.data();
    im.slot_aux = slot_aux.data();
    im.slot_packets = slot_packets.data();
    im.slot_hops = slot_hops.data();
    im.slot_events = slot_events.data();
    im.slot_e_net = slot_e_net.data();
    im.slot_e_syn = slot_e_syn.data();
    im.slot_e_dend = slot_e_dend.data();
    im.in_train_beg = in_train_beg.data();
    im.in_train_len = in_train_len.data();
    im.in_rate_period = in_rate_period.data();
    im.in_train_bits = in_train_bits.data();
    im.n_taps = static_cast<uint32_t>(tap_slot.size());
    im.tap_slot = tap_slot.data();
    im.tap_count = tap_count.data();
    im.tap_tc = tap_tc.data();
    im.tap_sc = tap_sc.data();
    im.n_ext = static_cast<uint32_t>(ext.size());
    im.slot_ext = ext.empty() ? nullptr : slot_ext.data();
    im.slice_core = slice_core.data();
    im.slice_axon_beg = slice_axon_beg.data();
    im.slice_axon_end = slice_axon_end.data();
    im.core_syn_base = core_syn_base.data();
    im.core_axon_in_latency = core_axon_in_latency.data();
    im.ax_pre = ax_pre.data();
    im.ax_syn_beg = ax_syn_beg.data();
    im.ax_nsyn = ax_nsyn.data();
    im.ax_proc_delay = ax_proc_delay.data();
    im.ax_lat_class = ax_lat_class.data();
    im.lat_class_per_event = lat_class_per_event.data();
    im.syn_meta = syn_meta.data();
    im.syn_weight = syn_weight.data();
    im.n_msg_cores = static_cast<uint32_t>(msg_core.size());
    im.msg_core = msg_core.data();
    im.msg_ax_beg = msg_ax_beg.data();
    im.msg_ax_pre = msg_ax_pre.data();
    im.msg_ax_nsyn = msg_ax_nsyn.data();
    im.msg_syn_beg = msg_syn_beg.data();
    im.msg_syn_post = msg_syn_post.data();
    im.msg_syn_weight = msg_syn_weight.data();
    im.msg_costs = msg_costs.data();
    return im;
}

void map_and_lower(const sanafe_desc &d, int n_ranks, int rank, uint32_t target_slices, uint32_t min_slice_axons, MappedChip &mc)
{
    const bool timing = std::getenv("SANAFE_MAP_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[map] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw std::invalid_argument("bad rank / n_ranks");
    // host threads of the edge passes (results do not depend on the count)
    int n_threads = static_cast<int>(std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency())));
    if (const char *env = std::getenv("SANAFE_MAP_THREADS")) n_threads = std::max(1, std::min(256, std::atoi(env)));
    // ------------------------------------------------------------------ architecture
    mc.n_tiles = d.n_tiles;
    mc.n_cores = d.n_cores;
    mc.noc_width = d.noc_width;
    mc.noc_height = d.noc_height;
    mc.noc_buffer = d.noc_buffer_size;
    mc.tile_x.resize(d.n_tiles);
    mc.tile_y.resize(d.n_tiles);
    for (int t = 0; t < d.n_tiles; t++)
    {
        mc.tile_x[t] = t / d.noc_height; // src/arch.cpp:78-88
        mc.tile_y[t] = t % d.noc_height;
    }
    mc.core_tile.assign(d.core_tile, d.core_tile + d.n_cores);
    mc.core_offset.resize(d.n_cores);
    {
        std::vector<uint32_t> per_tile(d.n_tiles, 0);
        for (int c = 0; c < d.n_cores; c++)
        {
            if (c > 0 && d.core_tile[c] < d.core_tile[c - 1]) throw std::invalid_argument("cores must be listed in tile order");
            mc.core_offset[c] = per_tile[d.core_tile[c]]++;
        }
        for (uint32_t n : per_tile) mc.max_cores_per_tile = std::max(mc.max_cores_per_tile, n);
    }
    std::vector<Template> templates;
    for (int t = 0; t < d.n_templates; t++) templates.push_back(read_template(d, t));
    auto tmpl_of = [&](uint32_t core) -> const Template & { return templates[d.core_template[core]]; };

    // ------------------------------------------------------------------ groups
    const int64_t N = d.n_neurons;
    mc.group_ptr.assign(d.group_ptr, d.group_ptr + d.n_groups + 1);
    mc.group_names.clear();
    for (int g = 0; g < d.n_groups; g++) mc.group_names.push_back(S(d, d.group_name[g]));
    mc.group_lex_order.resize(d.n_groups);
    std::iota(mc.group_lex_order.begin(), mc.group_lex_order.end(), 0);
    std::sort(mc.group_lex_order.begin(), mc.group_lex_order.end(),
            [&](int a, int b) { return mc.group_names[a] < mc.group_names[b]; }); // std::map<std::string,...> order
    std::vector<int> lex_rank(d.n_groups);
    for (int i = 0; i < d.n_groups; i++) lex_rank[mc.group_lex_order[i]] = i;
    std::vector<int32_t> group_of(N);
    for (int g = 0; g < d.n_groups; g++)
        for (int64_t n = d.group_ptr[g]; n < d.group_ptr[g + 1]; n++) group_of[n] = g;

    // ------------------------------------------------------------------ map_neurons, src/chip.cpp:186-234
    std::vector<int64_t> order;
    order.reserve(N);
    for (int g : mc.group_lex_order)
        for (int64_t n = d.group_ptr[g]; n < d.group_ptr[g + 1]; n++) order.push_back(n);
    std::stable_sort(order.begin(), order.end(),
            [&](int64_t a, int64_t b) { return d.neuron_map_order[a] < d.neuron_map_order[b]; });
    mc.core_ncount.assign(d.n_cores, 0);
    std::vector<uint32_t> offset_in_core(N);
    for (int64_t gid : order)
    {
        const int32_t c = d.neuron_core[gid];
        if (c < 0 || c >= d.n_cores)
            throw HardwareMappingError("Neuron: " + mc.group_names[group_of[gid]] + "." +
                    std::to_string(gid - d.group_ptr[group_of[gid]]) + " not mapped.");
        if (mc.core_ncount[c] >= static_cast<uint64_t>(d.core_max_neurons[c]))
            throw HardwareMappingError("Error: Exceeded maximum neurons per core.");
        offset_in_core[gid] = mc.core_ncount[c]++;
    }
    mc.core_nbase.resize(d.n_cores);
    uint64_t slots = 0;
    for (int c = 0; c < d.n_cores; c++)
    {
        mc.core_nbase[c] = static_cast<uint32_t>(slots);
        slots += (mc.core_ncount[c] + 63u) & ~63u;
        if (slots > 0xffffffc0ull) throw UnsupportedError("more than 2^32 neuron slots");
    }
    if (slots == 0) slots = 64;
    mc.n_global_slots = static_cast<uint32_t>(slots);
    mc.slot_of_gid.resize(N);
    mc.gid_of_slot.assign(mc.n_global_slots, -1);
    mc.core_of_slot.assign(mc.n_global_slots, 0);
    for (int c = 0; c < d.n_cores; c++)
        for (uint32_t k = 0; k < ((mc.core_ncount[c] + 63u) & ~63u); k++) mc.core_of_slot[mc.core_nbase[c] + k] = c;
    for (int64_t gid = 0; gid < N; gid++)
    {
        const uint32_t s = mc.core_nbase[d.neuron_core[gid]] + offset_in_core[gid];
        mc.slot_of_gid[gid] = s;
        mc.gid_of_slot[s] = gid;
    }
    // track_mapped_tiles_and_cores, src/chip.cpp:283-306
    {
        std::vector<uint8_t> used(d.n_tiles, 0);
        for (int c = 0; c < d.n_cores; c++)
            if (mc.core_ncount[c] > 0)
            {
                used[d.core_tile[c]] = 1;
                mc.mapped_cores++;
            }
        for (uint8_t u : used) mc.mapped_tiles += u;
    }
    // sim_timestep_sync, src/chip.cpp:562-574; LookupTable::get, src/utils.hpp:19-44
    if (d.n_sync <= 0) throw std::runtime_error("Table is empty");
    {
        int k = 0;
        while (k + 1 < d.n_sync && static_cast<uint64_t>(d.sync_key[k + 1]) <= mc.mapped_tiles) k++;
        if (static_cast<uint64_t>(d.sync_key[0]) > mc.mapped_tiles) k = 0;
        mc.sync_delay = d.sync_val[k];
    }

    // ------------------------------------------------------------------ rank window (tiles in contiguous blocks)
    {
        const uint32_t tpr = (d.n_tiles + n_ranks - 1) / n_ranks;
        const uint32_t t0 = std::min<uint32_t>(d.n_tiles, rank * tpr), t1 = std::min<uint32_t>(d.n_tiles, (rank + 1) * tpr);
        mc.first_core = d.n_cores;
        mc.last_core = 0;
        for (int c = 0; c < d.n_cores; c++)
            if (static_cast<uint32_t>(d.core_tile[c]) >= t0 && static_cast<uint32_t>(d.core_tile[c]) < t1)
            {
                mc.first_core = std::min<uint32_t>(mc.first_core, c);
                mc.last_core = std::max<uint32_t>(mc.last_core, c + 1);
            }
        if (mc.first_core >= mc.last_core) mc.first_core = mc.last_core = 0;
        mc.slot_offset = mc.first_core < static_cast<uint32_t>(d.n_cores) && mc.last_core > mc.first_core ? mc.core_nbase[mc.first_core] : 0;
        const uint32_t end_slot = mc.last_core > mc.first_core
                ? (mc.last_core < static_cast<uint32_t>(d.n_cores) ? mc.core_nbase[mc.last_core] : mc.n_global_slots)
                : mc.slot_offset;
        mc.n_slots = std::max<uint32_t>(64, end_slot - mc.slot_offset);
        if (mc.last_core == mc.first_core) throw UnsupportedError("rank " + std::to_string(rank) + " owns no tiles");
        // every rank's window of the global slot space (the spike exchange gathers these windows)
        mc.rank_slot_begin.assign(n_ranks + 1, mc.n_global_slots);
        for (int r = 0; r < n_ranks; r++)
        {
            const uint32_t r0 = std::min<uint32_t>(d.n_tiles, r * tpr);
            uint32_t fc = d.n_cores;
            for (int c = 0; c < d.n_cores; c++)
                if (static_cast<uint32_t>(d.core_tile[c]) >= r0)
                {
                    fc = c;
                    break;
                }
            mc.rank_slot_begin[r] = fc < static_cast<uint32_t>(d.n_cores) ? mc.core_nbase[fc] : mc.n_global_slots;
        }
    }
    const uint32_t LC = mc.last_core - mc.first_core;
    const uint32_t LS = mc.n_slots, SO = mc.slot_offset;

    // ------------------------------------------------------------------ Core::map_neuron, src/core.cpp:116-168
    std::vector<int32_t> dend_unit(N), soma_unit(N);
    {
        std::vector<std::vector<uint32_t>> used(d.n_cores); // neurons per (core, unit)
        std::map<std::tuple<int32_t, int32_t, int>, int> memo; // (template, name, role) -> unit
        auto lookup = [&](int32_t tm, int32_t name, bool dend) {
            const auto key = std::make_tuple(tm, name, dend ? 1 : 0);
            auto it = memo.find(key);
            if (it == memo.end()) it = memo.emplace(key, find_unit(templates[tm], S(d, name), false, dend, !dend)).first;
            return it->second;
        };
        for (int64_t gid : order)
        {
            const int32_t c = d.neuron_core[gid];
            const Template &t = tmpl_of(c);
            if (t.units.empty()) throw std::runtime_error("Error: No units defined");
            const int du = lookup(d.core_template[c], d.neuron_dendrite_hw[gid], true);
            const int su = lookup(d.core_template[c], d.neuron_soma_hw[gid], false);
            if (t.aout_e.empty()) throw std::runtime_error("Error: No axon out units defined");
            dend_unit[gid] = du;
            soma_unit[gid] = su;
            if (used[c].empty()) used[c].assign(t.units.size(), 0);
            used[c][du]++;
            if (su != du) used[c][su]++;
            if (used[c][su] > t.units[su].capacity || used[c][du] > t.units[du].capacity)
                throw HardwareMappingError("Too many neurons mapped to one h/w unit on core " + std::to_string(c) +
                        " (model capacity: src/models.hpp:29, 284, 347)");
        }
    }

    // ------------------------------------------------------------------ per-slot soma state (local slots)
    mc.l_core_nbase.resize(LC);
    mc.l_core_ncount.resize(LC);
    mc.core_axon_out_latency.resize(LC);
    mc.core_axon_in_latency.resize(LC);
    for (uint32_t k = 0; k < LC; k++)
    {
        const uint32_t c = mc.first_core + k;
        mc.l_core_nbase[k] = mc.core_nbase[c] - SO;
        mc.l_core_ncount[k] = mc.core_ncount[c];
        const Template &t = tmpl_of(c);
        mc.core_axon_out_latency[k] = t.aout_l.empty() ? 0.0 : t.aout_l[0];
        mc.core_axon_in_latency[k] = t.ain_l.empty() ? 0.0 : t.ain_l[0];
    }
    mc.slot_cls.assign(LS, 0);
    mc.slot_aux.assign(LS, 0);
    mc.slot_bias.assign(LS, 0.0);
    mc.slot_v0.assign(LS, 0.0);
    mc.slot_log_spikes.assign(LS, 0);
    mc.slot_log_potential.assign(LS, 0);
    mc.slot_model.assign(LS, SANAFE_SOMA_NONE);
    std::map<ClassKey, uint32_t> class_ids;
    std::map<CostKey, uint32_t> cost_ids;
    bool any_delay_dendrite = false, any_gated_delay = false, any_taps = false;
    std::vector<int32_t> taps_index(N, -1); // neuron -> entry of the tap tables (local neurons only)
    std::vector<uint8_t> neuron_dend_kind(N, 0); // 0 buffered accumulator, 1 zero, 2 delay line
    // InputModel seeds: every unit of every core is constructed up front, cores in id order, units in
    // template order (src/chip.cpp:83-87), and each `input` instance takes ++counter (src/models.hpp:347).
    // The counter is process-wide in the reference; this build numbers from a fresh process.
    std::vector<uint32_t> core_input_base(d.n_cores + 1, 0);
    for (int c = 0; c < d.n_cores; c++)
    {
        uint32_t n_in = 0;
        for (const UnitInfo &u : tmpl_of(c).units) n_in += (u.model == M_INPUT);
        core_input_base[c + 1] = core_input_base[c] + n_in;
    }
    int32_t random_mask_key = -1;
    for (int32_t k = 0; k < d.n_strings; k++)
        if (std::strcmp(d.strings[k], "random_mask") == 0) random_mask_key = k;
    // An `input` unit keeps ONE spike train, cursor, rate and generator (src/models.hpp:344-378), and nothing stops a
    // network from putting several neurons on it (snn/dendrite.yaml does): they then share that state -- every
    // neuron's attributes overwrite the unit's in mapping order, and at run time each update consumes the next
    // train element / Poisson draw.  With k neurons on the unit, the j-th in update order therefore sees elements
    // j, j+k, j+2k, ... of the final train.
    struct SharedInput
    {
        std::vector<double> train;
        double rate{0.0}, poisson{0.0};
        std::vector<int64_t> members; // in update (= mapping) order
        uint32_t gen{0};
    };
    std::map<std::pair<int32_t, int>, SharedInput> shared_inputs;
    for (int64_t gid : order)
    {
        const int32_t c = d.neuron_core[gid];
        const Template &t = tmpl_of(c);
        if (t.units[soma_unit[gid]].model != M_INPUT) continue;
        SharedInput &sh = shared_inputs[{c, soma_unit[gid]}];
        sh.members.push_back(gid);
        for (int64_t i = d.neuron_attr_ptr[gid]; i < d.neuron_attr_ptr[gid + 1]; i++)
        {
            const Attr a{&d.neuron_attrs, i, &d};
            if (!(a.fwd() & SANAFE_FWD_SOMA)) continue;
            const std::string k = a.key();
            if (k == "spikes")
            {
                if (a.type() != SANAFE_ATTR_LIST) throw std::runtime_error("Error: Attribute spikes is not a list");
                sh.train.assign(d.neuron_attrs.list_num + d.neuron_attrs.list_ptr[i], d.neuron_attrs.list_num + d.neuron_attrs.list_ptr[i + 1]);
            }
            else if (k == "poisson") sh.poisson = a.as_double();
            else if (k == "rate") sh.rate = a.as_double();
        }
    }
    {
        uint32_t next_gen = 0;
        for (auto &kv : shared_inputs) kv.second.gen = next_gen++;
    }
    // A `taps` unit is ONE RC line (src/models.hpp:104-156), whatever is mapped to it (snn/dendrite.yaml leaves its
    // input neurons on the default dendrite unit, the same line as its dendritic neuron).  Its configuration is what
    // all mapped neurons' attributes leave behind, applied in mapping order (MultiTapModel1D::set_attribute_neuron,
    // src/models.cpp:263-329).  At most one of those neurons may receive synapses (checked with the edges):
    // the others never touch the line, so giving each neuron a private copy is unobservable.
    struct SharedTaps
    {
        std::vector<double> v{0.0}, tc{0.0}, space;
    };
    std::map<std::pair<int32_t, int>, SharedTaps> shared_taps;
    for (int64_t gid : order)
    {
        const int32_t c = d.neuron_core[gid];
        const Template &t = tmpl_of(c);
        if (t.units[dend_unit[gid]].model != M_TAPS) continue;
        SharedTaps &sh = shared_taps[{c, dend_unit[gid]}];
        for (int64_t i = d.neuron_attr_ptr[gid]; i < d.neuron_attr_ptr[gid + 1]; i++)
        {
            const Attr a{&d.neuron_attrs, i, &d};
            if (!(a.fwd() & SANAFE_FWD_DENDRITE)) continue;
            const std::string k = a.key();
            auto list_of = [&]() {
                if (a.type() != SANAFE_ATTR_LIST) throw std::runtime_error("Error: Attribute " + k + " is not a list");
                return std::vector<double>(d.neuron_attrs.list_num + d.neuron_attrs.list_ptr[i], d.neuron_attrs.list_num + d.neuron_attrs.list_ptr[i + 1]);
            };
            if (k == "taps")
            {
                const size_t n_taps = static_cast<size_t>(a.as_int());
                if (n_taps == 0) throw std::invalid_argument("Number of taps must be > 0\n");
                sh.v.resize(n_taps);
                sh.tc.resize(n_taps);
                sh.space.resize(n_taps - 1);
            }
            else if (k == "time_constants")
            {
                sh.tc = list_of();
                if (sh.tc.size() < sh.v.size())
                    throw std::invalid_argument("Expected " + std::to_string(sh.v.size()) + " but received " + std::to_string(sh.tc.size()) + "time constants.");
            }
            else if (k == "space_constants")
            {
                sh.space = list_of();
                if (sh.space.size() < sh.v.size() - 1)
                    throw std::invalid_argument("Expected " + std::to_string(sh.v.size() - 1) + " but received " + std::to_string(sh.tc.size()) + "time constants.");
            }
        }
    }
    // `taps` dendrites the device does not cover run on the host instead: more than 8 taps, a buffer position other than
    // `soma` (outside), or several synapse-receiving neurons on ONE unit -- they then share its RC line, and every call
    // of any of them advances / charges the same state (MultiTapModel1D keeps no per-neuron state, src/models.hpp:165-198).
    std::vector<uint8_t> taps_need_host(d.n_cores, 0);
    {
        for (const auto &kv : shared_taps)
        {
            const int32_t c = kv.first.first;
            if (kv.second.v.size() > 8 || d.core_buffer_pos[c] != SANAFE_BUF_BEFORE_SOMA) taps_need_host[c] = 1;
        }
        if (!shared_taps.empty())
        {
            std::map<std::pair<int32_t, int>, int64_t> receiver; // (core, taps unit) -> the neuron that receives synapses
            for (int64_t e = 0; e < d.n_edges; e++)
            {
                const int64_t dst = d.edge_dst[e];
                if (dst < 0 || dst >= N) continue;
                const int32_t c = d.neuron_core[dst];
                if (tmpl_of(c).units[dend_unit[dst]].model != M_TAPS) continue;
                auto it = receiver.emplace(std::make_pair(c, dend_unit[dst]), dst).first;
                if (it->second != dst) taps_need_host[c] = 1;
            }
        }
    }
    // ------------------------------------------------------------------ host cores (MappedChip::HostCore)
    // A core runs on the host when its soma is part of the MESSAGE pipeline (buffer inside the soma unit or before
    // axon_out: the soma is called once per synaptic event) or when its template holds a plugin synapse / dendrite unit.
    std::vector<int32_t> host_core_index(d.n_cores, -1);
    for (int c = 0; c < d.n_cores; c++)
    {
        const Template &t = tmpl_of(c);
        bool host = d.core_buffer_pos[c] == SANAFE_BUF_INSIDE_SOMA || d.core_buffer_pos[c] == SANAFE_BUF_BEFORE_AXON_OUT;
        for (const UnitInfo &u : t.units) host = host || (u.model == M_PLUGIN && (u.syn || u.dend));
        host = host || taps_need_host[c]; // `taps` dendrites beyond what the device kernels cover (see above)
        if (!host || mc.core_ncount[c] == 0) continue;
        if (n_ranks != 1) throw UnsupportedError("cores that run on the host (buffer inside the soma unit / before axon_out, plugin synapse or dendrite units) need a single-rank chip");
        host_core_index[c] = static_cast<int32_t>(mc.host_cores.size());
        MappedChip::HostCore hc;
        hc.core = static_cast<uint32_t>(c);
        hc.bp = d.core_buffer_pos[c];
        if (t.ain_l.empty()) throw std::runtime_error("core receives spike messages but has no axon_in unit");
        hc.ain_latency = t.ain_l[0];
        for (size_t u = 0; u < t.units.size(); u++)
        {
            const UnitInfo &ui = t.units[u];
            MappedChip::HostCore::Unit hu;
            hu.desc_unit = d.tmpl_unit_ptr[d.core_template[c]] + static_cast<int>(u);
            hu.name = ui.name;
            hu.model = S(d, d.unit_model[hu.desc_unit]);
            hu.plugin_path = S(d, d.unit_plugin[hu.desc_unit]);
            hu.syn = ui.syn;
            hu.dend = ui.dend;
            hu.soma = ui.soma;
            hu.update_every_timestep = ui.update_every_timestep;
            hu.e_spike = ui.e_spike;
            hu.l_spike = ui.l_spike;
            hu.e_update = ui.e_update;
            hu.l_update = ui.l_update;
            hu.has_soma_e = ui.has_soma_e;
            hu.has_soma_l = ui.has_soma_l;
            for (int k = 0; k < 3; k++) hu.se[k] = ui.se[k], hu.sl[k] = ui.sl[k];
            if (ui.model == M_LIF && !ui.noise_path.empty()) throw UnsupportedError("LIF noise files on a core that runs on the host");
            hc.units.push_back(std::move(hu));
        }
        hc.neurons.resize(mc.core_ncount[c]);
        mc.host_cores.push_back(std::move(hc));
    }
    const bool any_host_core = !mc.host_cores.empty();
    {
        // Can these cores run on the device?  Buffer inside the soma unit / before axon_out (not a plugin or `taps` case), one
        // synapse unit `current_based`, one dendrite unit `accumulator`, one soma unit `truenorth`, each with its default
        // costs: then the message pipeline is synapse -> running sum -> TrueNorth update per event, which msgsoma_kernel
        // runs per post-synaptic neuron in delivery order.  All of the chip's host cores or none.
        bool ok = any_host_core && !(std::getenv("SANAFE_HOST_CORES") != nullptr && std::atoi(std::getenv("SANAFE_HOST_CORES")) != 0);
        for (const MappedChip::HostCore &hc : mc.host_cores)
        {
            ok = ok && (hc.bp == SANAFE_BUF_INSIDE_SOMA || hc.bp == SANAFE_BUF_BEFORE_AXON_OUT) && !taps_need_host[hc.core];
            int n_syn = 0, n_dend = 0, n_soma = 0;
            for (const UnitInfo &u : tmpl_of(static_cast<int>(hc.core)).units)
            {
                n_syn += u.syn, n_dend += u.dend, n_soma += u.soma;
                if (u.syn) ok = ok && u.model == M_CURRENT_BASED && !u.dend && !u.soma && u.e_spike && u.l_spike;
                if (u.dend) ok = ok && u.model == M_ACCUMULATOR && !u.syn && !u.soma && u.e_update && u.l_update;
                if (u.soma) ok = ok && u.model == M_TRUENORTH && !u.syn && !u.dend && u.has_soma_e && u.has_soma_l;
            }
            ok = ok && n_syn == 1 && n_dend == 1 && n_soma == 1;
        }
        mc.msg_on_device = ok;
    }
    if (any_host_core)
    {
        // per-unit neuron addresses in arrival (mapping) order: Core::map_neuron, src/core.cpp:116-168
        std::map<std::pair<int32_t, int>, uint32_t> next_addr;
        for (int64_t gid : order)
        {
            const int32_t c = d.neuron_core[gid];
            if (host_core_index[c] < 0) continue;
            MappedChip::HostCore::Neuron &hn = mc.host_cores[host_core_index[c]].neurons[offset_in_core[gid]];
            hn.gid = gid;
            hn.slot = mc.slot_of_gid[gid] - SO;
            hn.soma_unit = soma_unit[gid];
            hn.dend_unit = dend_unit[gid];
            hn.dend_addr = next_addr[{c, dend_unit[gid]}]++;
            hn.soma_addr = (soma_unit[gid] != dend_unit[gid]) ? next_addr[{c, soma_unit[gid]}]++ : hn.dend_addr; // combined unit: one address
        }
    }

    // ---- optional perf-trace columns (MappedChip::LogPlan) ----
    {
        MappedChip::LogPlan &lg = mc.log;
        for (int t = 0; t < d.n_tiles; t++) lg.any = lg.any || d.tile_log_energy[t];
        for (int c = 0; c < d.n_cores; c++)
        {
            lg.any = lg.any || d.core_log_energy[c];
            for (const UnitInfo &u : tmpl_of(c).units) lg.any = lg.any || u.log;
        }
        // (a rank of a tile-sharded chip only notes that columns are wanted: they are whole-chip sums, computed on the
        //  single-rank twin of sanafe_chip_attach_whole from the gathered statuses)
        if (lg.any && n_ranks == 1)
        {
            lg.core_unit_beg.assign(d.n_cores + 1, 0);
            for (int c = 0; c < d.n_cores; c++) lg.core_unit_beg[c + 1] = lg.core_unit_beg[c] + static_cast<uint32_t>(tmpl_of(c).units.size());
            const size_t nu = lg.core_unit_beg[d.n_cores];
            lg.unit_e_spike.assign(nu, 0.0);
            lg.unit_e_update.assign(nu, 0.0);
            lg.unit_used.assign(nu, 0);
            lg.slot_soma_unit.assign(LS, 0);
            lg.slot_dend_unit.assign(LS, 0);
            lg.core_e_ain.assign(d.n_cores, 0.0);
            lg.core_e_aout.assign(d.n_cores, 0.0);
            lg.core_bp.assign(d.n_cores, 0);
            std::map<std::string, MappedChip::LogPlan::Column> cols;
            for (int c = 0; c < d.n_cores; c++)
            {
                const Template &t = tmpl_of(c);
                const uint32_t tile = d.core_tile[c];
                const std::string tn = S(d, d.tile_name[tile]), cn = S(d, d.core_name[c]);
                lg.core_bp[c] = d.core_buffer_pos[c];
                lg.core_e_ain[c] = t.ain_e.size() == 1 ? t.ain_e[0] : 0.0;   // counters on unit 0, energy read from the LAST
                lg.core_e_aout[c] = t.aout_e.size() == 1 ? t.aout_e[0] : 0.0; // unit (src/chip.cpp:1215-1221, 1248-1253)
                if (d.tile_log_energy[tile]) cols[tn + ".energy"] = {tn + ".energy", 0, tile, 0, 0};
                if (d.core_log_energy[c]) cols[tn + "." + cn + ".energy"] = {tn + "." + cn + ".energy", 1, tile, static_cast<uint32_t>(c), 0};
                for (size_t u = 0; u < t.units.size(); u++)
                {
                    const UnitInfo &ui = t.units[u];
                    const size_t k = lg.core_unit_beg[c] + u;
                    if (ui.e_spike) lg.unit_e_spike[k] = *ui.e_spike;
                    if (ui.e_update) lg.unit_e_update[k] = *ui.e_update;
                    const std::string base = tn + "." + cn + "." + ui.name;
                    if (ui.log_energy)
                        cols[base + ".energy"] = {base + ".energy", 2, tile, static_cast<uint32_t>(c), static_cast<uint32_t>(u)};
                    if (ui.log_latency)
                        cols[base + ".latency"] = {base + ".latency", 3, tile, static_cast<uint32_t>(c), static_cast<uint32_t>(u)};
                }
            }
            for (auto &kv : cols) lg.columns.push_back(kv.second);
            for (int64_t gid = 0; gid < N; gid++)
            {
                const int32_t c = d.neuron_core[gid];
                lg.unit_used[lg.core_unit_beg[c] + soma_unit[gid]] = 1;
                lg.unit_used[lg.core_unit_beg[c] + dend_unit[gid]] = 1;
                const uint32_t s = mc.slot_of_gid[gid];
                if (s < SO || s >= SO + LS) continue;
                lg.slot_soma_unit[s - SO] = static_cast<uint8_t>(soma_unit[gid]);
                lg.slot_dend_unit[s - SO] = static_cast<uint8_t>(dend_unit[gid]);
            }
        }
    }
    std::vector<uint32_t> rand_slots; // global slots of all TrueNorth neurons with random_mask != 0
    std::map<std::pair<int32_t, int>, uint32_t> noise_stream_ids; // (core, unit) -> stream
    for (int64_t gid = 0; gid < N; gid++)
    {
        const int32_t c = d.neuron_core[gid];
        const Template &t = tmpl_of(c);
        const UnitInfo &du = t.units[dend_unit[gid]];
        const UnitInfo &su = t.units[soma_unit[gid]];
        const int bp = d.core_buffer_pos[c];
        const bool msg_core = host_core_index[c] >= 0 && mc.msg_on_device; // soma in the message pipeline, on the device
        if (host_core_index[c] >= 0 && !msg_core)
        {
            // evaluated by the host library every step (host/host_cores.cpp); the device only keeps its status and spike bit
            const uint32_t hs = mc.slot_of_gid[gid];
            if (su.model == M_TRUENORTH && random_mask_key >= 0)
                for (int64_t i = d.neuron_attr_ptr[gid]; i < d.neuron_attr_ptr[gid + 1]; i++)
                    if (d.neuron_attrs.key[i] == random_mask_key && d.neuron_attrs.num[i] != 0.0)
                        throw UnsupportedError("TrueNorth random_mask on a core that runs on the host");
            mc.slot_cls[hs - SO] = SANAFE_SOMA_HOST;
            mc.slot_model[hs - SO] = SANAFE_SOMA_HOST;
            mc.slot_log_spikes[hs - SO] = d.neuron_log_spikes[gid];
            mc.slot_log_potential[hs - SO] = d.neuron_log_potential[gid];
            continue;
        }
        if (msg_core && su.model == M_TRUENORTH && random_mask_key >= 0)
            for (int64_t i = d.neuron_attr_ptr[gid]; i < d.neuron_attr_ptr[gid + 1]; i++)
                if (d.neuron_attrs.key[i] == random_mask_key && d.neuron_attrs.num[i] != 0.0)
                    throw UnsupportedError("TrueNorth random_mask on a core whose soma is part of the message pipeline");
        if (!msg_core && bp != SANAFE_BUF_BEFORE_SOMA && bp != SANAFE_BUF_INSIDE_DENDRITE && bp != SANAFE_BUF_BEFORE_DENDRITE)
            throw UnsupportedError("buffer position " + std::to_string(bp) +
                    " is not implemented on the MI355X backend (supported: soma/outside, dendrite/inside, dendrite/outside)");
        if (du.model == M_PLUGIN)
            throw UnsupportedError("dendrite model of unit '" + du.name + "' is not implemented on the MI355X backend");
        if (du.model == M_TAPS && bp != SANAFE_BUF_BEFORE_SOMA)
            throw UnsupportedError("`taps` dendrites are implemented for `buffer_position: soma` (outside the unit) only");
        uint8_t kind = msg_core ? SANAFE_IN_NONE : SANAFE_IN_BUFFERED; // (msg cores: the soma's input arrives per event, msgsoma_kernel)
        if (bp == SANAFE_BUF_INSIDE_DENDRITE && du.model == M_ACCUMULATOR) kind = SANAFE_IN_ZERO, neuron_dend_kind[gid] = 1;
        if (bp == SANAFE_BUF_BEFORE_DENDRITE && du.model == M_ACCUMULATOR) kind = SANAFE_IN_LAST;
        if (du.model == M_TAPS) kind = SANAFE_IN_TAPS, neuron_dend_kind[gid] = 3, any_taps = true;
        if (du.model == M_ACC_DELAY)
        {
            if (bp == SANAFE_BUF_BEFORE_SOMA) kind = SANAFE_IN_GATED, any_gated_delay = true;
            any_delay_dendrite = true;
            neuron_dend_kind[gid] = 2;
            if (bp == SANAFE_BUF_BEFORE_DENDRITE)
            {
                // the message pipeline stops after the synapse: delays of the edges never reach the delivery; the unit's
                // neuron-side call uses the delay of ITS synapse address 0 (filled in below, with the edges)
                kind = SANAFE_IN_LAST_DELAY;
                neuron_dend_kind[gid] = 4;
                any_gated_delay = true; // a current added at step t with delay 5 matures at t + 6: seven ring slots
            }
        }
        const uint32_t s = mc.slot_of_gid[gid];
        if (su.model == M_TRUENORTH && random_mask_key >= 0)
        {
            // std::rand() is one process-wide sequence (src/models.cpp:757): every rank needs the position of its
            // neurons among all consumers of the chip
            long mask = 0;
            for (int64_t i = d.neuron_attr_ptr[gid]; i < d.neuron_attr_ptr[gid + 1]; i++)
                if (d.neuron_attrs.key[i] == random_mask_key && ((d.neuron_attrs.fwd ? d.neuron_attrs.fwd[i] : 7) & SANAFE_FWD_SOMA) &&
                        d.neuron_attrs.type[i] == SANAFE_ATTR_INT)
                    mask = static_cast<long>(d.neuron_attrs.num[i]);
            if (mask > 0) rand_slots.push_back(s);
        }
        if (s < SO || s >= SO + LS) continue; // not ours
        const uint32_t ls = s - SO;
        // ---- cost class (src/pipeline.hpp:574-714) ----
        sanafe_hip_cost_class cc{};
        const bool host_soma = (su.model == M_PLUGIN);
        if (host_soma && kind == SANAFE_IN_LAST_DELAY)
            throw UnsupportedError("plugin somas behind an accumulator_with_delay with the buffer before the dendrite unit");
        if (host_soma && (su.syn || su.dend))
            throw UnsupportedError("plugin unit '" + su.name + "': only soma plugins are implemented on the MI355X backend");
        // a plugin may simulate its own energy/latency instead of using architecture defaults
        if (!host_soma && !su.has_soma_e) throw std::runtime_error("Soma unit does not simulate energy or provide default energy costs in the architecture description.");
        if (!host_soma && !su.has_soma_l) throw std::runtime_error("Soma unit does not simulate latency or provide default latency costs in the architecture description.");
        cc.soma_energy[0] = su.se[0];
        cc.soma_energy[1] = su.se[0] + su.se[1];
        cc.soma_energy[2] = (su.se[0] + su.se[1]) + su.se[2];
        cc.soma_latency[0] = su.sl[0];
        cc.soma_latency[1] = su.sl[0] + su.sl[1];
        cc.soma_latency[2] = (su.sl[0] + su.sl[1]) + su.sl[2];
        if (bp <= SANAFE_BUF_INSIDE_DENDRITE)
        {
            if (!du.e_update) throw std::runtime_error("Dendrite unit does not simulate energy or provide a default energy cost in the architecture description.");
            if (!du.l_update) throw std::runtime_error("Dendrite unit does not simulate latency or provide a default latency cost in the architecture description.");
            cc.dendrite_energy = *du.e_update;
            cc.dendrite_latency = *du.l_update;
        }
        // buffer before axon_out: the neuron pipeline holds no unit at all -- nothing is costed in the neuron loop
        if (msg_core && bp == SANAFE_BUF_BEFORE_AXON_OUT) cc = sanafe_hip_cost_class{};
        CostKey ck;
        std::memcpy(ck.b, &cc, sizeof(cc));
        auto cit = cost_ids.find(ck);
        if (cit == cost_ids.end())
        {
            cit = cost_ids.emplace(ck, static_cast<uint32_t>(mc.cost_classes.size())).first;
            mc.cost_classes.push_back(cc);
            if (mc.cost_classes.size() > 1024) throw UnsupportedError("more than 1024 distinct cost classes");
        }
        // ---- soma parameters: set_attribute_neuron in key order, src/models.cpp:375-439, 664-722, 832-853 ----
        uint32_t model = SANAFE_SOMA_NONE, pcls = 0;
        sanafe_hip_soma_class p{};
        p.leak_decay = 1.0;
        p.reset_mode = SANAFE_RESET_HARD;
        p.reverse_reset_mode = SANAFE_RESET_NONE;
        p.leak_towards_zero = 1;
        if (su.model == M_LIF) model = SANAFE_SOMA_LIF;
        else if (su.model == M_TRUENORTH) model = SANAFE_SOMA_TRUENORTH, p.leak_decay = 0.0;
        else if (su.model == M_INPUT) model = SANAFE_SOMA_INPUT;
        else model = SANAFE_SOMA_HOST;
        const bool persist = msg_core && bp == SANAFE_BUF_BEFORE_AXON_OUT; // (the class word gets SANAFE_SOMA_PERSIST below)
        // (an `input` soma never asks its dendrite for anything: such a neuron merely sits on the unit)
        if (du.model == M_TAPS && model == SANAFE_SOMA_INPUT) kind = SANAFE_IN_BUFFERED;
        if (du.model == M_TAPS && model != SANAFE_SOMA_INPUT)
        {
            const SharedTaps &sh = shared_taps.at({c, dend_unit[gid]});
            const std::vector<double> &v = sh.v, &tc = sh.tc, &space = sh.space;
            if (v.size() > 8) throw UnsupportedError("`taps` dendrites with more than 8 taps are not implemented on the MI355X backend");
            taps_index[gid] = static_cast<int32_t>(mc.tap_slot.size());
            mc.tap_slot.push_back(ls);
            mc.tap_count.push_back(static_cast<uint32_t>(v.size()));
            for (size_t k = 0; k < 8; k++)
            {
                mc.tap_tc.push_back(k < v.size() ? tc[k] : 0.0);
                mc.tap_sc.push_back(k + 1 < v.size() ? space[k] : 0.0);
            }
        }
        std::vector<double> train;
        double rate = 0.0, poisson = 0.0;
        uint32_t tn_mask = 0;
        for (int64_t i = d.neuron_attr_ptr[gid]; i < d.neuron_attr_ptr[gid + 1]; i++)
        {
            const Attr a{&d.neuron_attrs, i, &d};
            const std::string k = a.key();
            if (k == "soma_hw_name" || k == "default_synapse_hw_name" || k == "dendrite_hw_name" || k == "log_spikes" ||
                    k == "log_potential" || k == "log_v")
                throw std::invalid_argument("Reserved neuron attribute '" + k + "' cannot be used as a model attribute. Pass it as a direct argument instead (if supported).");
            if (!(a.fwd() & SANAFE_FWD_SOMA)) continue;
            if (model == SANAFE_SOMA_LIF || model == SANAFE_SOMA_TRUENORTH)
            {
                SomaAttr sa;
                sa.key = k;
                sa.type = a.type();
                sa.num = (a.type() == SANAFE_ATTR_STRING || a.type() == SANAFE_ATTR_LIST) ? 0.0 : d.neuron_attrs.num[i];
                if (a.type() == SANAFE_ATTR_STRING) sa.str = a.as_string();
                SomaAttrEffect fx;
                apply_soma_attribute(model, sa, p, fx);
                if (fx.bias_set) mc.slot_bias[ls] = fx.bias;
                if (fx.potential_set) mc.slot_v0[ls] = fx.potential;
                if (fx.random_mask_set) tn_mask = fx.random_mask;
            }
            else
            {
                if (k == "spikes")
                {
                    if (a.type() != SANAFE_ATTR_LIST) throw std::runtime_error("Error: Attribute spikes is not a list");
                    train.assign(d.neuron_attrs.list_num + d.neuron_attrs.list_ptr[i], d.neuron_attrs.list_num + d.neuron_attrs.list_ptr[i + 1]);
                }
                else if (k == "poisson") poisson = a.as_double();
                else if (k == "rate") rate = a.as_double();
            }
        }
        if (model == SANAFE_SOMA_HOST)
        {
            // one plugin instance per (core, unit); addresses in arrival order on the unit
            uint32_t hu = 0;
            while (hu < mc.host_units.size() && !(mc.host_units[hu].core == static_cast<uint32_t>(c) &&
                           mc.host_units[hu].desc_unit == d.tmpl_unit_ptr[d.core_template[c]] + soma_unit[gid]))
                hu++;
            if (hu == mc.host_units.size())
            {
                MappedChip::HostUnit h;
                h.core = c;
                h.desc_unit = d.tmpl_unit_ptr[d.core_template[c]] + soma_unit[gid];
                h.name = su.name;
                h.model = S(d, d.unit_model[h.desc_unit]);
                h.plugin_path = S(d, d.unit_plugin[h.desc_unit]);
                h.has_energy = su.has_soma_e;
                h.has_latency = su.has_soma_l;
                for (int k = 0; k < 3; k++)
                {
                    h.energy[k] = cc.soma_energy[k];
                    h.latency[k] = cc.soma_latency[k];
                }
                mc.host_units.push_back(h);
            }
            MappedChip::HostNeuron hn;
            hn.slot = ls;
            hn.core_local = c - mc.first_core;
            hn.unit = hu;
            hn.gid = gid;
            mc.host_neurons.push_back(hn);
        }
        else if (model == SANAFE_SOMA_INPUT)
        {
            const SharedInput &sh = shared_inputs.at({c, soma_unit[gid]});
            const size_t k_members = sh.members.size();
            const size_t j_member = static_cast<size_t>(std::find(sh.members.begin(), sh.members.end(), gid) - sh.members.begin());
            train.clear();
            for (size_t b = j_member; b < sh.train.size(); b += k_members) train.push_back(sh.train[b]);
            rate = sh.rate;
            poisson = sh.poisson;
            if (poisson > 0.0) // the draw happens at every update; it can only matter when p > 0
            {
                MappedChip::ExtColumn col;
                col.slot = ls;
                col.kind = MappedChip::ExtColumn::Poisson;
                col.poisson = poisson;
                col.seed = core_input_base[c] + static_cast<uint32_t>(su.input_rank) + 1u;
                col.gen = sh.gen; // the unit's generator: its neurons draw from it one after the other
                col.unit_key = (static_cast<uint64_t>(c) << 16) | static_cast<uint64_t>(soma_unit[gid]);
                mc.ext.push_back(col);
            }
            mc.slot_aux[ls] = static_cast<uint32_t>(mc.in_train_beg.size());
            const uint32_t beg = static_cast<uint32_t>(mc.in_train_bits.size()) * 32u;
            mc.in_train_beg.push_back(beg);
            mc.in_train_len.push_back(static_cast<uint32_t>(train.size()));
            mc.in_train_bits.resize(mc.in_train_bits.size() + (train.size() + 31) / 32, 0u);
            for (size_t b = 0; b < train.size(); b++)
                if (train[b] != 0.0) mc.in_train_bits[(beg + b) >> 5] |= 1u << ((beg + b) & 31u);
            int64_t period = 0;
            if (rate > 0.0)
            {
                period = static_cast<long>(1.0 / rate); // src/models.cpp:891-892
                if (period == 0) throw std::invalid_argument("input rate > 1 makes the reference divide by zero (SURVEY quirk 14)");
            }
            mc.in_rate_period.push_back(period);
            mc.in_shared.push_back(k_members > 1 ? 1 : 0);
            mc.in_seed.push_back(core_input_base[c] + static_cast<uint32_t>(su.input_rank) + 1u);
            mc.in_unit_key.push_back((static_cast<uint64_t>(c) << 16) | static_cast<uint64_t>(soma_unit[gid]));
        }
        else
        {
            if (model == SANAFE_SOMA_TRUENORTH && tn_mask != 0)
            {
                MappedChip::ExtColumn col;
                col.slot = ls;
                col.kind = MappedChip::ExtColumn::TrueNorthRand;
                col.mask = tn_mask;
                mc.ext.push_back(col);
            }
            if (model == SANAFE_SOMA_LIF && !su.noise_path.empty())
            {
                const auto key2 = std::make_pair(c, soma_unit[gid]);
                auto sit = noise_stream_ids.find(key2);
                if (sit == noise_stream_ids.end())
                {
                    sit = noise_stream_ids.emplace(key2, static_cast<uint32_t>(mc.noise_streams.size())).first;
                    MappedChip::NoiseStream ns;
                    ns.path = su.noise_path;
                    ns.random_mask = su.noise_random_mask;
                    ns.unit_key = (static_cast<uint64_t>(c) << 16) | static_cast<uint64_t>(soma_unit[gid]);
                    mc.noise_streams.push_back(ns);
                }
                MappedChip::ExtColumn col;
                col.slot = ls;
                col.kind = MappedChip::ExtColumn::LifNoise;
                col.stream = sit->second;
                mc.ext.push_back(col);
            }
            ClassKey key;
            std::memset(&key, 0, sizeof(key));
            const sanafe_hip_soma_class canon = canonical_soma_class(p);
            std::memcpy(key.b, &canon, sizeof(canon));
            auto it = class_ids.find(key);
            if (it == class_ids.end())
            {
                it = class_ids.emplace(key, static_cast<uint32_t>(mc.soma_classes.size())).first;
                mc.soma_classes.push_back(canon);
                if (mc.soma_classes.size() > 65536) throw UnsupportedError("more than 65536 distinct soma parameter sets");
            }
            pcls = it->second;
        }
        if (persist) model = SANAFE_SOMA_PERSIST; // (its soma class is the TrueNorth parameter set: the per-event updates use it)
        mc.slot_cls[ls] = model | (static_cast<uint32_t>(kind) << 3) | (cit->second << 6) | (pcls << 16);
        if (du.model == M_TAPS && model != SANAFE_SOMA_INPUT) mc.slot_aux[ls] = static_cast<uint32_t>(taps_index[gid]);
        mc.slot_model[ls] = static_cast<uint8_t>(model);
        mc.slot_log_spikes[ls] = d.neuron_log_spikes[gid];
        mc.slot_log_potential[ls] = d.neuron_log_potential[gid];
    }
    {
        // soma addresses follow the order neurons arrive on the unit (map order), src/pipeline.cpp:79-85
        std::vector<size_t> idx(mc.host_neurons.size());
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
            return d.neuron_map_order[mc.host_neurons[a].gid] < d.neuron_map_order[mc.host_neurons[b].gid];
        });
        std::vector<uint32_t> next(mc.host_units.size(), 0);
        for (size_t i : idx) mc.host_neurons[i].addr = next[mc.host_neurons[i].unit]++;
    }
    if (!mc.ext.empty())
    {
        // columns in slot order: the order the reference's sweep reaches the neurons (core id, map order)
        std::sort(mc.ext.begin(), mc.ext.end(), [](const MappedChip::ExtColumn &a, const MappedChip::ExtColumn &b) { return a.slot < b.slot; });
        std::sort(rand_slots.begin(), rand_slots.end());
        mc.n_rand_global = rand_slots.size();
        mc.slot_ext.assign(LS, 0xffffffffu);
        for (size_t k = 0; k < mc.ext.size(); k++)
        {
            MappedChip::ExtColumn &col = mc.ext[k];
            mc.slot_ext[col.slot] = static_cast<uint32_t>(k);
            if (col.kind == MappedChip::ExtColumn::TrueNorthRand)
                col.rand_index = static_cast<uint64_t>(std::lower_bound(rand_slots.begin(), rand_slots.end(), col.slot + SO) - rand_slots.begin());
        }
    }
    if (mc.soma_classes.empty()) mc.soma_classes.push_back(sanafe_hip_soma_class{});
    if (mc.cost_classes.empty()) mc.cost_classes.push_back(sanafe_hip_cost_class{});
    // gated delay lines mature one step later; chips without delays keep two rows of the time-step buffer (this step's and
    // the next one's), so the device may deliver a quiet step's spikes from inside the neuron launch (DevImage: push delivery)
    mc.ring_slots = any_gated_delay ? 7 : any_delay_dendrite ? 6 : 2;
    if (any_taps) mc.ring_slots = 8; // the delay field doubles as the tap index (<= 7): the image contract is field < ring_slots

    lap("neurons + slots");
    // ------------------------------------------------------------------ map_connections, src/chip.cpp:334-380
    const int64_t E = d.n_edges;
    std::vector<int32_t> edge_syn_unit(E);  // unit index inside the destination core's template
    std::vector<uint8_t> edge_delay_eff;     // effective delay seen by the delay dendrite
    parallel_tasks(block_count(n_threads, E), [&](int t) {
        const int T = block_count(n_threads, E);
        int32_t memo_name = -2, memo_tmpl = -1, memo_unit = -1; // consecutive edges nearly always repeat the lookup
        for (int64_t e = E * t / T; e < E * (t + 1) / T; e++)
        {
            const int64_t dst = d.edge_dst[e];
            if (d.edge_src[e] < 0 || d.edge_src[e] >= N || dst < 0 || dst >= N) throw std::invalid_argument("edge endpoint out of range");
            int32_t hw_id = d.edge_synapse_hw[e]; // get_synapse_hw_name, src/chip.cpp:308-332
            if (hw_id < 0) hw_id = d.neuron_synapse_hw[dst];
            const int32_t tm = d.core_template[d.neuron_core[dst]];
            const Template &tp = templates[tm];
            if (hw_id != memo_name || tm != memo_tmpl)
            {
                memo_unit = find_unit(tp, S(d, hw_id), true, false, false);
                memo_name = hw_id;
                memo_tmpl = tm;
            }
            const int su = memo_unit;
            const UnitInfo &u = tp.units[su];
            edge_syn_unit[e] = su;
            if (host_core_index[d.neuron_core[dst]] >= 0) continue; // the host replays this core's message pipeline: any unit goes
            if (u.model != M_CURRENT_BASED) throw UnsupportedError("synapse unit '" + u.name + "' (plugin) is not implemented on the MI355X backend");
            if (!u.e_spike) throw std::runtime_error("Synapse unit does not simulate energy or provide a default energy cost in the architecture description.");
            if (!u.l_spike) throw std::runtime_error("Synapse unit does not simulate latency or provide a default latency cost in the architecture description.");
        }
    });
    // neurons that receive through a synapse unit flagged update_every_timestep (MappedNeuron::
    // check_for_synapse_updates_every_timestep, src/mapped.cpp:32-40): only host cores' units can make that observable
    std::vector<uint8_t> receives_forced_synapse;
    if (any_host_core)
    {
        receives_forced_synapse.assign(N, 0);
        for (int64_t e = 0; e < E; e++)
            if (tmpl_of(d.neuron_core[d.edge_dst[e]]).units[edge_syn_unit[e]].update_every_timestep) receives_forced_synapse[d.edge_dst[e]] = 1;
    }
    std::vector<uint64_t> syn_addr; // per edge: address on its synapse unit = arrival order in map_connections order
    if ((d.edge_delay && any_delay_dendrite) || any_taps || any_host_core)
    {
        // The delay dendrite stores delays[] indexed by the SYNAPSE unit's address of the
        // connection (src/mapped.cpp:60-89, src/models.cpp:133-152): connections that reach the
        // same dendrite unit through different synapse units share table entries.  Replay the
        // assignments in map_connections order to get the delay each connection really sees.
        EdgeOrder eo(E);
        std::vector<uint64_t> rank_base(d.n_groups + 1, 0);
        for (int i = 0; i < d.n_groups; i++)
            rank_base[i + 1] = rank_base[i] + (d.group_ptr[mc.group_lex_order[i] + 1] - d.group_ptr[mc.group_lex_order[i]]);
        auto src_key = [&](uint64_t e) {
            const int64_t s = d.edge_src[e];
            const int g = group_of[s];
            return rank_base[lex_rank[g]] + (s - d.group_ptr[g]);
        };
        eo.sort(src_key, N, n_threads);
        eo.release_scratch();
        std::map<std::pair<uint32_t, int>, uint64_t> syn_count;            // (core, synapse unit) -> next address
        std::map<std::pair<uint32_t, int>, std::vector<uint8_t>> delays;   // (core, dendrite unit) -> delays[]
        syn_addr.assign(E, 0);
        std::map<std::pair<uint32_t, int>, int64_t> taps_owner; // (core, taps unit) -> the neuron that receives synapses
        for (int64_t pos = 0; pos < E; pos++)
        {
            const uint64_t e = eo[pos];
            const int64_t dst = d.edge_dst[e];
            const uint32_t c = d.neuron_core[dst];
            const uint64_t addr = syn_count[{c, edge_syn_unit[e]}]++;
            syn_addr[e] = addr;
            const int edge_attr = d.edge_delay ? d.edge_delay[e] : -1;
            if (neuron_dend_kind[dst] == 3)
            {
                auto it = taps_owner.emplace(std::make_pair(c, dend_unit[dst]), dst).first;
                const bool local_dst = mc.slot_of_gid[dst] >= SO && mc.slot_of_gid[dst] < SO + LS;
                if (local_dst && taps_index[dst] < 0)
                    throw UnsupportedError("synapses into an input neuron that sits on a `taps` dendrite unit");
                if (it->second != dst)
                    throw UnsupportedError("several neurons receive synapses through one `taps` dendrite unit (they would share its RC "
                                           "line): not implemented on the MI355X backend");
            }
            if (neuron_dend_kind[dst] == 2 || neuron_dend_kind[dst] == 4)
            {
                auto &tab = delays[{c, dend_unit[dst]}];
                if (tab.size() <= addr) tab.resize(addr + 1, 0); // every forwarded attribute resizes (weight included)
                if (edge_attr >= 0 && edge_attr < 64) // (a `tap` attribute means nothing to this unit)
                {
                    if (edge_attr > 5) throw std::runtime_error("Error: delay > max delay\n");
                    tab[addr] = static_cast<uint8_t>(edge_attr);
                }
            }
            else if (neuron_dend_kind[dst] == 3 && edge_attr >= 64)
            {
                // MultiTapModel1D::set_attribute_edge "tap": synapse_to_tap[address], src/models.cpp:330-342
                auto &tab = delays[{c, dend_unit[dst]}];
                if (tab.size() <= addr) tab.resize(addr + 1, 0);
                tab[addr] = static_cast<uint8_t>(edge_attr - 64);
            }
        }
        // SANAFE_IN_LAST_DELAY: the delay the unit's synapse address 0 carries
        for (int64_t gid = 0; gid < N; gid++)
        {
            if (neuron_dend_kind[gid] != 4) continue;
            const uint32_t sl = mc.slot_of_gid[gid];
            if (sl < SO || sl >= SO + LS) continue;
            const auto it = delays.find({static_cast<uint32_t>(d.neuron_core[gid]), dend_unit[gid]});
            mc.slot_aux[sl - SO] = (it != delays.end() && !it->second.empty()) ? it->second[0] : 0u;
        }
        edge_delay_eff.assign(E, 0);
        for (int64_t e = 0; e < E; e++)
        {
            const int64_t dst = d.edge_dst[e];
            if (neuron_dend_kind[dst] != 2 && neuron_dend_kind[dst] != 3) continue;
            const auto &tab = delays[{static_cast<uint32_t>(d.neuron_core[dst]), dend_unit[dst]}];
            edge_delay_eff[e] = syn_addr[e] < tab.size() ? tab[syn_addr[e]] : 0;
            if (neuron_dend_kind[dst] == 3 && taps_index[dst] >= 0 && edge_delay_eff[e] >= mc.tap_count[taps_index[dst]])
                throw std::logic_error("Tap should be >= 0 and less than taps.\n"); // src/models.cpp:232-237 (raised at load here)
        }
    }

    lap("edge units + delays");
    // ------------------------------------------------------------------ map_axons, src/chip.cpp:382-408, 1263-1391
    // Delivery order at a destination core = (source core id, source neuron order, connection
    // order) = (pre slot, creation order): two stable counting sorts.
    EdgeOrder eo(E);
    eo.sort([&](uint64_t e) { return static_cast<size_t>(mc.slot_of_gid[d.edge_src[e]]); }, mc.n_global_slots, n_threads);
    eo.sort([&](uint64_t e) { return static_cast<size_t>(d.neuron_core[d.edge_dst[e]]); }, d.n_cores, n_threads);
    eo.release_scratch();
    lap("edge sorts");
    // The edges of the local cores are one contiguous run [local_beg, local_end) of the sorted order, so a
    // synapse's place in the image is known before its axon is: position in the run.
    auto dest_core_at = [&](int64_t k) { return static_cast<uint32_t>(d.neuron_core[d.edge_dst[eo[k]]]); };
    auto first_at_or_after = [&](uint32_t core) { // first sorted position whose destination core is >= core
        int64_t lo = 0, hi = E;
        while (lo < hi)
        {
            const int64_t mid = lo + (hi - lo) / 2;
            if (dest_core_at(mid) < core) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    };
    const int64_t local_beg = first_at_or_after(mc.first_core), local_end = first_at_or_after(mc.last_core);
    mc.syn_meta.resize(static_cast<size_t>(local_end - local_beg));
    mc.syn_weight.resize(static_cast<size_t>(local_end - local_beg));
    const bool build_log = mc.log.any && n_ranks == 1; // (ranks of a sharded chip leave the plan to the whole-chip twin)
    if (build_log) mc.log.syn_units.resize(static_cast<size_t>(local_end - local_beg));

    // Pass 1 (threads, blocks of whole axons): everything about an axon that depends only on its own
    // edges.  Pass 2 (serial, below) numbers the axons and accumulates the per-neuron aggregates in
    // delivery order, so floating-point sums do not depend on the thread count.
    struct AxonBlock
    {
        std::vector<int64_t> first; // sorted position of the axon's first edge (host cores list their synapses from it)
        std::vector<uint32_t> pre, nsyn, hops, dc;
        std::vector<uint8_t> uniform;
        std::vector<double> proc, first_lat, min_hop, e_net, e_syn, e_dend, e_hop;
    };
    const int T = block_count(n_threads, static_cast<uint64_t>(E));
    std::vector<AxonBlock> blocks(T);
    auto axon_start = [&](int64_t k) { // first position >= k that starts an axon
        while (k > 0 && k < E && dest_core_at(k) == dest_core_at(k - 1) &&
                mc.slot_of_gid[d.edge_src[eo[k]]] == mc.slot_of_gid[d.edge_src[eo[k - 1]]])
            k++;
        return k;
    };
    parallel_tasks(T, [&](int t) {
        AxonBlock &B = blocks[t];
        int64_t i = axon_start(E * t / T);
        const int64_t end = axon_start(E * (t + 1) / T);
        const size_t guess = static_cast<size_t>((end - i) / 4 + 16);
        for (auto *v : {&B.pre, &B.nsyn, &B.hops, &B.dc}) v->reserve(guess);
        for (auto *v : {&B.proc, &B.first_lat, &B.min_hop, &B.e_net, &B.e_syn, &B.e_dend}) v->reserve(guess);
        B.uniform.reserve(guess);
        while (i < end)
        {
            const uint64_t e0 = eo[i];
            const uint32_t dc = d.neuron_core[d.edge_dst[e0]];
            const uint32_t pre = mc.slot_of_gid[d.edge_src[e0]];
            int64_t j = i;
            while (j < E && static_cast<uint32_t>(d.neuron_core[d.edge_dst[eo[j]]]) == dc && mc.slot_of_gid[d.edge_src[eo[j]]] == pre) j++;
            const Template &dt = tmpl_of(dc);
            const int bp = d.core_buffer_pos[dc];
            const bool local = dc >= mc.first_core && dc < mc.last_core;
            // a core that runs on the host: its synapse / dendrite / soma costs are whatever its units return at run time
            // (host/host_cores.cpp adds them per step); nothing of it enters the static per-spike totals or the device image
            const bool host_dest = host_core_index[dc] >= 0;
            // processing delay of the message: pipeline_process_axon_in + process_message, src/chip.cpp:738-800
            if (dt.ain_l.empty()) throw std::runtime_error("core receives spike messages but has no axon_in unit");
            double proc = dt.ain_l[0];
            double e_syn = 0.0, e_dend = 0.0;
            double first_lat = 0.0;
            bool uniform_lat = true;
            for (int64_t k = i; k < j; k++)
            {
                const uint64_t e = eo[k];
                const UnitInfo &su = dt.units[edge_syn_unit[e]];
                const int64_t dst = d.edge_dst[e];
                double lat = 0.0; // execute_pipeline: total_latency
                if (host_dest)
                {
                    if (build_log) mc.log.unit_used[mc.log.core_unit_beg[dc] + edge_syn_unit[e]] = 1; // (is_used: a connection is mapped to it)
                    if (local) // a hole in the image's synapse arrays: no axon refers to it, every scan skips it (lost charge)
                    {
                        mc.syn_meta[static_cast<size_t>(k - local_beg)] = 1u << 19;
                        mc.syn_weight[static_cast<size_t>(k - local_beg)] = 0.0;
                    }
                    continue;
                }
                lat += *su.l_spike;
                e_syn += *su.e_spike;
                if (bp > SANAFE_BUF_BEFORE_DENDRITE)
                {
                    const UnitInfo &du = dt.units[dend_unit[dst]];
                    if (!du.e_update) throw std::runtime_error("Dendrite unit does not simulate energy or provide a default energy cost in the architecture description.");
                    if (!du.l_update) throw std::runtime_error("Dendrite unit does not simulate latency or provide a default latency cost in the architecture description.");
                    lat += *du.l_update;
                    e_dend += *du.e_update;
                }
                proc += lat;
                if (k == i) first_lat = lat;
                else if (lat != first_lat) uniform_lat = false;
                if (local)
                {
                    const uint32_t post = offset_in_core[dst];
                    if (post > 0xffffu) throw UnsupportedError("more than 65536 neurons on one core");
                    uint32_t meta = post;
                    if (neuron_dend_kind[dst] == 2 || neuron_dend_kind[dst] == 3) // delay value, or tap index
                        meta |= static_cast<uint32_t>(edge_delay_eff.empty() ? 0 : edge_delay_eff[e]) << 16;
                    if (neuron_dend_kind[dst] == 1) meta |= 1u << 19; // charge is lost inside a plain accumulator (quirk 1)
                    mc.syn_meta[static_cast<size_t>(k - local_beg)] = meta;
                    mc.syn_weight[static_cast<size_t>(k - local_beg)] = d.edge_weight[e];
                    if (build_log)
                    {
                        mc.log.syn_units[static_cast<size_t>(k - local_beg)] =
                                static_cast<uint16_t>(static_cast<uint32_t>(edge_syn_unit[e]) | (static_cast<uint32_t>(dend_unit[dst]) << 8));
                        mc.log.unit_used[mc.log.core_unit_beg[dc] + edge_syn_unit[e]] = 1; // (benign race: every writer stores 1)
                    }
                }
            }
            // network costs: sim_estimate_network_costs, src/chip.cpp:1127-1169
            const uint32_t sc = mc.core_of_slot[pre];
            const uint32_t st = mc.core_tile[sc], dtile = mc.core_tile[dc];
            const uint32_t sx = mc.tile_x[st], sy = mc.tile_y[st], dx = mc.tile_x[dtile], dy = mc.tile_y[dtile];
            const uint32_t xh = sx > dx ? sx - dx : dx - sx, yh = sy > dy ? sy - dy : dy - sy;
            double min_hop = 0.0, e_hop = 0.0;
            if (sx < dx)
            {
                min_hop += static_cast<double>(xh) * d.tile_hop_latency[st * 4 + SANAFE_DIR_EAST];
                e_hop += static_cast<double>(xh) * d.tile_hop_energy[dtile * 4 + SANAFE_DIR_EAST];
            }
            else
            {
                min_hop += static_cast<double>(xh) * d.tile_hop_latency[st * 4 + SANAFE_DIR_WEST];
                e_hop += static_cast<double>(xh) * d.tile_hop_energy[dtile * 4 + SANAFE_DIR_WEST];
            }
            if (sy < dy)
            {
                min_hop += static_cast<double>(yh) * d.tile_hop_latency[st * 4 + SANAFE_DIR_NORTH];
                e_hop += static_cast<double>(yh) * d.tile_hop_energy[dtile * 4 + SANAFE_DIR_NORTH];
            }
            else
            {
                min_hop += static_cast<double>(yh) * d.tile_hop_latency[st * 4 + SANAFE_DIR_SOUTH];
                e_hop += static_cast<double>(yh) * d.tile_hop_energy[dtile * 4 + SANAFE_DIR_SOUTH];
            }
            // axon energies count only when the core has exactly one such unit: the counters live on
            // unit 0 but the energy is read from the LAST unit (src/chip.cpp:1215-1221, 1248-1253)
            const Template &stt = tmpl_of(sc);
            const double e_aout = stt.aout_e.size() == 1 ? stt.aout_e[0] : 0.0;
            const double e_ain = dt.ain_e.size() == 1 ? dt.ain_e[0] : 0.0;
            B.first.push_back(i);
            B.pre.push_back(pre);
            B.nsyn.push_back(static_cast<uint32_t>(j - i));
            B.hops.push_back(xh + yh);
            B.dc.push_back(dc);
            B.uniform.push_back(uniform_lat ? 1 : 0);
            B.proc.push_back(proc);
            B.first_lat.push_back(first_lat);
            B.min_hop.push_back(min_hop);
            B.e_net.push_back((e_aout + e_hop) + e_ain);
            if (build_log) B.e_hop.push_back(e_hop);
            B.e_syn.push_back(e_syn);
            B.e_dend.push_back(e_dend);
            i = j;
        }
    });
    lap("axons + synapses (threads)");

    std::vector<uint32_t> g_packets(mc.n_global_slots, 0), g_hops(mc.n_global_slots, 0), g_events(mc.n_global_slots, 0);
    std::vector<double> g_e_net(mc.n_global_slots, 0.0), g_e_syn(mc.n_global_slots, 0.0), g_e_dend(mc.n_global_slots, 0.0);
    mc.core_syn_base.assign(LC, 0);
    std::vector<uint64_t> core_axon_beg(LC + 1, 0);
    std::vector<uint32_t> dest_axon_count(d.n_cores, 0);
    std::vector<uint64_t> out_count(static_cast<size_t>(mc.n_global_slots) + 1, 0);
    struct MsgAxonOut
    {
        uint32_t pre, dc, axon_id, hops;
        double min_hop;
        uint32_t host_core, idx_in_core;
        double e_hop;
    };
    std::vector<MsgAxonOut> mx; // inbound axons of the cores whose soma is part of the message pipeline (msg_on_device)
    const bool keep_out_tables = (n_ranks == 1);
    {
        size_t n_local_axons = 0;
        for (const AxonBlock &B : blocks)
            for (uint32_t dc : B.dc) n_local_axons += (dc >= mc.first_core && dc < mc.last_core);
        mc.ax_pre.reserve(n_local_axons);
        mc.ax_syn_beg.reserve(n_local_axons);
        mc.ax_nsyn.reserve(n_local_axons);
        mc.ax_proc_delay.reserve(n_local_axons);
        mc.ax_lat_class.reserve(n_local_axons);
        if (keep_out_tables)
        {
            mc.ax_dest_core.reserve(n_local_axons);
            mc.ax_dest_axon_id.reserve(n_local_axons);
            mc.ax_hops.reserve(n_local_axons);
            mc.ax_min_hop_delay.reserve(n_local_axons);
        }
    }
    uint64_t syn0 = 0; // synapses of the local axons numbered so far
    for (AxonBlock &B : blocks)
    {
        for (size_t a = 0; a < B.pre.size(); a++)
        {
            const uint32_t pre = B.pre[a], dc = B.dc[a], nsyn = B.nsyn[a];
            g_packets[pre] += 1;
            g_hops[pre] += B.hops[a];
            g_events[pre] += nsyn;
            g_e_net[pre] += B.e_net[a];
            g_e_syn[pre] += B.e_syn[a];
            g_e_dend[pre] += B.e_dend[a];
            const uint32_t axon_id = dest_axon_count[dc]++;
            if (dc < mc.first_core || dc >= mc.last_core) continue;
            const uint32_t lc = dc - mc.first_core;
            if (host_core_index[dc] >= 0)
            {
                // the host replays this core: the axon and its synapses go to the core's own tables, in delivery order;
                // their places in the image's synapse arrays stay holes
                MappedChip::HostCore &hc = mc.host_cores[host_core_index[dc]];
                MappedChip::HostCore::Axon ha;
                ha.pre = pre;
                ha.syn_beg = static_cast<uint32_t>(hc.synapses.size());
                ha.n_syn = nsyn;
                if (mc.msg_on_device && keep_out_tables) // the host still rebuilds the messages INTO such a core (detailed timing)
                    mx.push_back(MsgAxonOut{pre, dc, axon_id, B.hops[a], B.min_hop[a], static_cast<uint32_t>(host_core_index[dc]),
                            static_cast<uint32_t>(hc.axons.size()), build_log ? B.e_hop[a] : 0.0});
                hc.axons.push_back(ha);
                for (int64_t k = B.first[a]; k < B.first[a] + nsyn; k++)
                {
                    const uint64_t e = eo[k];
                    MappedChip::HostCore::Synapse hs;
                    hs.unit = edge_syn_unit[e];
                    hs.addr = static_cast<uint32_t>(syn_addr[e]);
                    hs.post = offset_in_core[d.edge_dst[e]];
                    hs.edge = static_cast<int64_t>(e);
                    hs.weight = d.edge_weight[e];
                    hs.pre_checks_synapses = !receives_forced_synapse.empty() && receives_forced_synapse[d.edge_src[e]] != 0;
                    hc.synapses.push_back(hs);
                }
                syn0 += nsyn;
                continue;
            }
            if (core_axon_beg[lc + 1] == 0) mc.core_syn_base[lc] = syn0;
            if (syn0 - mc.core_syn_base[lc] > 0xffffffffull) throw UnsupportedError("more than 2^32 synapses on one core");
            mc.ax_pre.push_back(pre);
            mc.ax_syn_beg.push_back(static_cast<uint32_t>(syn0 - mc.core_syn_base[lc]));
            mc.ax_nsyn.push_back(nsyn);
            mc.ax_proc_delay.push_back(B.proc[a]);
            uint8_t cls = 255;
            if (B.uniform[a])
            {
                const double first_lat = B.first_lat[a];
                size_t q = 0;
                while (q < mc.lat_class_per_event.size() && mc.lat_class_per_event[q] != first_lat) q++;
                if (q == mc.lat_class_per_event.size() && q < 255) mc.lat_class_per_event.push_back(first_lat);
                if (q < 255) cls = static_cast<uint8_t>(q);
            }
            mc.ax_lat_class.push_back(cls);
            if (build_log) mc.log.ax_e_hop.push_back(B.e_hop[a]);
            if (keep_out_tables)
            {
                mc.ax_dest_core.push_back(dc);
                mc.ax_dest_axon_id.push_back(axon_id);
                mc.ax_hops.push_back(B.hops[a]);
                mc.ax_min_hop_delay.push_back(B.min_hop[a]);
                out_count[pre + 1]++;
            }
            core_axon_beg[lc + 1]++;
            syn0 += nsyn;
        }
        B = AxonBlock(); // release as we go
    }
    if (syn0 != static_cast<uint64_t>(local_end - local_beg)) throw std::logic_error("mapper: synapse numbering out of step");
    lap("axon numbering + aggregates");
    if (mc.msg_on_device)
    {
        // the tables of the cores whose soma is part of the message pipeline, as the image carries them (sanafe_hip_image::msg_*)
        mc.msg_ax_beg.push_back(0);
        mc.msg_syn_beg.push_back(0);
        for (const MappedChip::HostCore &hc : mc.host_cores) // (ascending core id)
        {
            mc.msg_core.push_back(hc.core - mc.first_core);
            for (const MappedChip::HostCore::Axon &ax : hc.axons)
            {
                mc.msg_ax_pre.push_back(ax.pre);
                mc.msg_ax_nsyn.push_back(ax.n_syn);
            }
            for (const MappedChip::HostCore::Synapse &hs : hc.synapses)
            {
                mc.msg_syn_post.push_back(hs.post);
                mc.msg_syn_weight.push_back(hs.weight);
            }
            mc.msg_ax_beg.push_back(static_cast<uint32_t>(mc.msg_ax_pre.size()));
            mc.msg_syn_beg.push_back(static_cast<uint32_t>(mc.msg_syn_post.size()));
            sanafe_hip_msg_core_costs k{};
            k.axon_in_latency = hc.ain_latency;
            std::array<uint32_t, 3> roles{0, 0, 0}; // the core's synapse, dendrite and soma unit (one each: the capability check)
            const std::vector<UnitInfo> &units = tmpl_of(static_cast<int>(hc.core)).units;
            for (size_t q0 = 0; q0 < units.size(); q0++)
            {
                const UnitInfo &u = units[q0];
                if (u.syn) k.synapse_energy = *u.e_spike, k.synapse_latency = *u.l_spike, roles[0] = static_cast<uint32_t>(q0);
                if (u.dend) k.dendrite_energy = *u.e_update, k.dendrite_latency = *u.l_update, roles[1] = static_cast<uint32_t>(q0);
                if (u.soma)
                {
                    for (int q = 0; q < 3; q++) k.soma_energy[q] = u.se[q], k.soma_latency[q] = u.sl[q];
                    roles[2] = static_cast<uint32_t>(q0);
                }
            }
            mc.msg_costs.push_back(k);
            mc.msg_units.push_back(roles);
        }
    }
    mc.lat_class_per_event.resize(255, 0.0);
    for (uint32_t k = 0; k < LC; k++) core_axon_beg[k + 1] += core_axon_beg[k];
    // cores without inbound axons still need a valid synapse base
    {
        uint64_t run = 0;
        for (uint32_t k = 0; k < LC; k++)
        {
            if (core_axon_beg[k + 1] == core_axon_beg[k]) mc.core_syn_base[k] = run;
            else
            {
                const uint64_t last = core_axon_beg[k + 1] - 1;
                run = mc.core_syn_base[k] + mc.ax_syn_beg[last] + mc.ax_nsyn[last];
            }
        }
    }
    // per-slot aggregates of the local slots
    mc.slot_packets.assign(g_packets.begin() + SO, g_packets.begin() + SO + std::min<uint32_t>(LS, mc.n_global_slots - SO));
    mc.slot_hops.assign(g_hops.begin() + SO, g_hops.begin() + SO + std::min<uint32_t>(LS, mc.n_global_slots - SO));
    mc.slot_events.assign(g_events.begin() + SO, g_events.begin() + SO + std::min<uint32_t>(LS, mc.n_global_slots - SO));
    mc.slot_e_net.assign(g_e_net.begin() + SO, g_e_net.begin() + SO + std::min<uint32_t>(LS, mc.n_global_slots - SO));
    mc.slot_e_syn.assign(g_e_syn.begin() + SO, g_e_syn.begin() + SO + std::min<uint32_t>(LS, mc.n_global_slots - SO));
    mc.slot_e_dend.assign(g_e_dend.begin() + SO, g_e_dend.begin() + SO + std::min<uint32_t>(LS, mc.n_global_slots - SO));
    for (auto *v : {&mc.slot_packets, &mc.slot_hops, &mc.slot_events}) v->resize(LS, 0);
    for (auto *v : {&mc.slot_e_net, &mc.slot_e_syn, &mc.slot_e_dend}) v->resize(LS, 0.0);

    // ------------------------------------------------------------------ delivery slices
    {
        const uint64_t A = mc.ax_pre.size();
        // Slice size: a multiple of 1,024 axons (one 256-axon chunk per wavefront of the delivery workgroup); large slices
        // a multiple of 8,192 (one run of 8 chunks per wavefront, deliver_kernel), so that the four wavefronts of a
        // workgroup finish together -- 17 k-axon slices cost 3 % against 16 k, 10 k-axon ones 8 % against 8 k (measured).
        uint64_t chunk = std::max<uint64_t>(std::max<uint32_t>(4, min_slice_axons & ~3u), ((A / std::max<uint32_t>(1, target_slices)) + 1023) & ~1023ull);
        if (chunk > 8192) chunk = (chunk + 8191) & ~8191ull;
        const char *align_env = std::getenv("SANAFE_SLICE_ALIGN"); // 0: cut every core by axon count (A/B runs)
        const bool align_to_source_slots = !(align_env != nullptr && std::atoi(align_env) == 0);
        for (uint32_t k = 0; k < LC; k++)
        {
            uint64_t b = core_axon_beg[k];
            const uint64_t e = core_axon_beg[k + 1];
            // A core that hears from most of the neurons of a wide span of the chip (C3: 92-99 % of them) gets BITMAP axon records
            // on the device (deliver_kernel): a run of a wavefront is then 8 windows of 256 SOURCE SLOTS, whatever the number of
            // axons in them, so such a core is cut where the source slot passes a multiple of 8,192 -- every slice holds whole
            // runs, the same number for each of the four wavefronts of its workgroup (cut by axon count, 6 slices of 170.7
            // windows make 22 runs: 6, 6, 5, 5 per wavefront).
            if (e - b > chunk && chunk >= 8192 && align_to_source_slots) // (small chips keep small slices: one chunk per wavefront)
            {
                const uint64_t u0 = mc.ax_pre[b] >> 13, units = (mc.ax_pre[e - 1] >> 13) - u0 + 1;
                bool ascending = true;
                for (uint64_t a = b + 1; a < e && ascending; a++) ascending = mc.ax_pre[a] > mc.ax_pre[a - 1];
                if (ascending && units >= 2 && (e - b) * 4ull >= units * 8192ull)
                {
                    const uint64_t n_cuts = std::min<uint64_t>(units, std::max<uint64_t>(1, (e - b + chunk / 2) / chunk));
                    for (uint64_t i = 0; i < n_cuts; i++)
                    {
                        const uint64_t slot_end = (u0 + ((i + 1) * units) / n_cuts) << 13;
                        const uint64_t cut = (i + 1 == n_cuts) ? e
                                                               : static_cast<uint64_t>(std::lower_bound(mc.ax_pre.begin() + b, mc.ax_pre.begin() + e,
                                                                                               static_cast<uint32_t>(std::min<uint64_t>(slot_end, 0xffffffffull))) - mc.ax_pre.begin());
                        if (cut == b) continue; // (no axon in these units)
                        mc.slice_core.push_back(k);
                        mc.slice_axon_beg.push_back(b);
                        mc.slice_axon_end.push_back(cut);
                        b = cut;
                    }
                    continue;
                }
            }
            while (b < e)
            {
                // later slices start on a multiple of 4 axons so the 16-byte ax_pre loads stay aligned
                uint64_t n = std::min<uint64_t>(chunk, e - b);
                if (b + n < e) n -= (b + n) & 3ull;
                mc.slice_core.push_back(k);
                mc.slice_axon_beg.push_back(b);
                mc.slice_axon_end.push_back(b + n);
                b += n;
            }
        }
    }
    // ------------------------------------------------------------------ out tables (host message reconstruction)
    if (keep_out_tables)
    {
        mc.out_ptr.assign(out_count.begin(), out_count.end());
        for (size_t s = 0; s < mc.n_global_slots; s++) mc.out_ptr[s + 1] += mc.out_ptr[s];
        mc.n_device_axons = mc.ax_pre.size();
        if (mx.empty())
        {
        mc.out_axon.resize(mc.ax_pre.size());
        std::vector<uint64_t> cur(mc.out_ptr.begin(), mc.out_ptr.end() - 1);
        for (uint64_t a = 0; a < mc.ax_pre.size(); a++) mc.out_axon[cur[mc.ax_pre[a]]++] = a; // ascending destination core
        }
        else
        {
            // chips with msg cores (small ones): the device's axons and the msg cores' in one list per source neuron, ascending
            // destination core (a neuron has one axon per destination core)
            const uint64_t A = mc.ax_pre.size();
            std::vector<uint32_t> msg_base(mc.host_cores.size() + 1, 0);
            for (size_t k = 0; k < mc.host_cores.size(); k++) msg_base[k + 1] = msg_base[k] + static_cast<uint32_t>(mc.host_cores[k].axons.size());
            std::vector<uint32_t> all_pre(mc.ax_pre.begin(), mc.ax_pre.end());
            mc.ax_dest_core.resize(A + mx.size());
            mc.ax_dest_axon_id.resize(A + mx.size());
            mc.ax_hops.resize(A + mx.size());
            mc.ax_min_hop_delay.resize(A + mx.size());
            if (build_log) mc.log.ax_e_hop.resize(A + mx.size());
            all_pre.resize(A + mx.size());
            for (const MsgAxonOut &m : mx)
            {
                const uint64_t id = A + msg_base[m.host_core] + m.idx_in_core;
                mc.ax_dest_core[id] = m.dc;
                mc.ax_dest_axon_id[id] = m.axon_id;
                mc.ax_hops[id] = m.hops;
                mc.ax_min_hop_delay[id] = m.min_hop;
                if (build_log) mc.log.ax_e_hop[id] = m.e_hop;
                all_pre[id] = m.pre;
                out_count[m.pre + 1]++;
            }
            mc.out_ptr.assign(out_count.begin(), out_count.end());
            for (size_t s = 0; s < mc.n_global_slots; s++) mc.out_ptr[s + 1] += mc.out_ptr[s];
            std::vector<uint64_t> ids(A + mx.size());
            std::iota(ids.begin(), ids.end(), 0ull);
            std::stable_sort(ids.begin(), ids.end(), [&](uint64_t x, uint64_t y) {
                return all_pre[x] != all_pre[y] ? all_pre[x] < all_pre[y] : mc.ax_dest_core[x] < mc.ax_dest_core[y];
            });
            mc.out_axon = ids;
        }
    }
}
} // namespace sanafe_amd
