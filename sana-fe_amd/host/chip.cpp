// chip.cpp -- host runtime of the MI355X path: implements include/sanafe_host.h.
//
// Owns the mapped network, the device chip (libsanafe_hip.so) and the host-side
// timing models.  Mirrors SpikingChip::sim / step / reset / get_* (src/chip.cpp:477-621,
// 1766-1831) and, for `detailed` timing, schedule_messages_timestep_detailed
// (src/schedule.cpp:208-620), which is an inherently serial discrete-event simulation the
// reference also runs on the CPU.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <thread>
#include <type_traits>
#include <list>
#include <memory>
#include <queue>
#include <sstream>
#include <string>
#include <vector>

#include <dlfcn.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>

#include "../../include/sanafe_host.h"
#include "comm.hpp"
#include "host_cores.hpp"
#include "mapper.hpp"
#include "plugin_abi/pipeline.hpp"

using sanafe_amd::MappedChip;

namespace
{
thread_local std::string g_err;
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
} // namespace
namespace sanafe_amd
{
int host_fail(int code, const std::string &msg) { return fail(code, msg); } // for the other translation units
}
namespace
{
#define DEV(expr)                                                       \
    do                                                                  \
    {                                                                   \
        int rc_ = (expr);                                               \
        if (rc_ != 0) return fail(rc_, sanafe_hip_last_error());        \
    } while (0)

constexpr double NEG_INF = -std::numeric_limits<double>::infinity();

// glibc's rand() (TYPE_3 additive feedback generator, r[i] = r[i-3] + r[i-31], seed 1 unless srand was
// called -- the reference never calls it): the sequence std::rand() yields in the reference's process
// (src/models.cpp:752-758), kept private so the host process's own rand() state is left alone.
class GlibcRand
{
public:
    explicit GlibcRand(uint32_t seed = 1)
    {
        int32_t word = seed == 0 ? 1 : static_cast<int32_t>(seed);
        r_[0] = static_cast<uint32_t>(word);
        for (int i = 1; i < 31; i++)
        {
            const long hi = word / 127773, lo = word % 127773; // 16807 * word % 2147483647 without overflow
            long w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            word = static_cast<int32_t>(w);
            r_[i] = static_cast<uint32_t>(word);
        }
        f_ = 3;
        b_ = 0;
        for (int i = 0; i < 310; i++) (void) next();
    }
    uint32_t next()
    {
        r_[f_] += r_[b_];
        const uint32_t out = r_[f_] >> 1;
        f_ = (f_ + 1) % 31;
        b_ = (b_ + 1) % 31;
        return out;
    }

private:
    uint32_t r_[31];
    int f_, b_;
};

// Host-side sources of the external value streams (include/sanafe_hip.h, slot_ext): one row of values
// per timestep, generated in the order the reference's sweep would have consumed them.
struct ExtStreams
{
    const MappedChip *mc{nullptr};
    std::vector<std::mt19937> poisson_gen; // per input unit instance (src/models.hpp:366-374)
    std::vector<uint64_t> poisson_gen_key; // ... its (core, unit) key
    std::vector<size_t> poisson_col_gen;   // per Poisson column: its unit's generator
    std::uniform_real_distribution<double> uni{0.0, 1.0};
    GlibcRand rand;
    std::vector<std::vector<int>> noise_values; // per stream: the file's entries
    std::vector<size_t> noise_pos;

    void init(const MappedChip &m)
    {
        mc = &m;
        std::map<uint64_t, size_t> gen_of_unit; // one generator per input unit instance
        for (const MappedChip::ExtColumn &c : m.ext)
            if (c.kind == MappedChip::ExtColumn::Poisson)
            {
                if (!gen_of_unit.count(c.unit_key))
                {
                    gen_of_unit[c.unit_key] = poisson_gen.size();
                    poisson_gen.emplace_back(c.seed);
                    // (a double from std::uniform_real_distribution takes two 32-bit draws: generate_canonical<double, 53>)
                    if (c.skip_updates > 0) poisson_gen.back().discard(2ull * static_cast<unsigned long long>(c.skip_updates));
                    poisson_gen_key.push_back(c.unit_key);
                }
                poisson_col_gen.push_back(gen_of_unit[c.unit_key]);
            }
        for (const MappedChip::NoiseStream &ns : m.noise_streams)
        {
            // LoihiLifModel::set_attribute_hw "noise", src/models.cpp:354-366
            FILE *f = std::fopen(ns.path.c_str(), "rb");
            if (!f) throw std::runtime_error("Failed to open noise stream");
            std::string text;
            char buf[65536];
            size_t n;
            while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
            std::fclose(f);
            // loihi_read_noise_stream, src/models.cpp:589-627: one getline per update, `iss >> int`
            // (0 when the entry does not parse), rewinding at end of file
            std::vector<int> vals;
            size_t pos = 0;
            while (pos < text.size())
            {
                size_t nl = text.find('\n', pos);
                if (nl == std::string::npos) nl = text.size();
                int v = 0;
                std::istringstream iss(text.substr(pos, nl - pos));
                if (!(iss >> v)) v = 0;
                vals.push_back(v);
                pos = nl + 1;
            }
            noise_values.push_back(std::move(vals));
            noise_pos.push_back(0);
        }
    }
    // load(net, overwrite=false) after timesteps (src/chip.cpp:129-138): every programmed unit keeps its state -- the process's
    // std::rand() sequence goes on, a unit's std::mt19937 and its noise file's read position stay where they are; units the new
    // network brings into use start fresh.
    void carry_from(const ExtStreams &old)
    {
        rand = old.rand;
        for (size_t k = 0; k < poisson_gen.size(); k++)
            for (size_t j = 0; j < old.poisson_gen.size(); j++)
                if (old.poisson_gen_key[j] == poisson_gen_key[k]) poisson_gen[k] = old.poisson_gen[j];
        for (size_t k = 0; k < noise_pos.size(); k++)
            for (size_t j = 0; j < old.noise_pos.size(); j++)
                if (old.mc->noise_streams[j].unit_key == mc->noise_streams[k].unit_key) noise_pos[k] = old.noise_pos[j];
    }
    void fill_row(int32_t *row)
    {
        size_t pg = 0;
        uint64_t rand_cursor = 0;
        for (size_t k = 0; k < mc->ext.size(); k++)
        {
            const MappedChip::ExtColumn &c = mc->ext[k];
            if (c.kind == MappedChip::ExtColumn::Poisson)
            {
                row[k] = (c.poisson > uni(poisson_gen[poisson_col_gen[pg++]])) ? 1 : 0;
            }
            else if (c.kind == MappedChip::ExtColumn::TrueNorthRand)
            {
                for (; rand_cursor < c.rand_index; rand_cursor++) (void) rand.next(); // neurons of other ranks
                row[k] = static_cast<int32_t>(rand.next() & c.mask);
                rand_cursor++;
            }
            else
            {
                std::vector<int> &vals = noise_values[c.stream];
                if (vals.empty()) throw std::runtime_error("Couldn't read noise entry from file");
                size_t &p = noise_pos[c.stream];
                if (p >= vals.size()) p = 0;
                const MappedChip::NoiseStream &ns = mc->noise_streams[c.stream];
                long r = vals[p++]; // loihi_generate_noise, src/models.cpp:629-651
                const long sign = r & ns.sign_mask;
                r &= ns.random_mask;
                if (sign != 0) r |= ~ns.random_mask;
                row[k] = static_cast<int32_t>(r);
            }
        }
        for (; rand_cursor < mc->n_rand_global; rand_cursor++) (void) rand.next();
    }
};

struct Msg : sanafe_message
{
    bool in_noc{false};
};
// What the NoC schedule itself reads and writes of a message: 96 bytes instead of the 240 of the traced record.
// Used when no message trace is kept (same member names, so the scheduler is one template).
struct SchedMsg
{
    double generation_delay, processing_delay, min_hop_delay;
    double sent_timestamp, received_timestamp, processed_timestamp;
    double blocking_delay, network_delay, messages_along_route;
    uint32_t src_core_id, dest_core_id;
    uint16_t hops, src_x, src_y, dest_x, dest_y, src_core_offset;
    uint8_t placeholder, in_noc;
};
} // namespace

struct sanafe_chip
{
    MappedChip mc;
    sanafe_hip_chip *dev{nullptr};
    int device{-1}, n_ranks{1}, rank{0};
    sanafe_amd::Exchange xc; // the per-step spike exchange of a tile-sharded chip
    int64_t n_neurons{0};
    int64_t total_timesteps{0};
    int64_t total_messages_sent{0};
    int scheduler_threads{0}; // schedule_create_threads, src/schedule.cpp:169-179
    ExtStreams ext;
    std::vector<int32_t> ext_rows;
    // ---- parameter patches between sim() calls (MappedNeuron::set_attributes, src/mapped.cpp:113-166) ----
    std::map<std::string, uint32_t> class_ids; // bytes of a canonical soma class -> index in mc.soma_classes
    std::vector<uint32_t> dirty_slots;
    bool classes_dirty{false};
    bool inputs_dirty{false};
    int64_t ext_hook_steps{0};         // rows drawn through sanafe_chip_generate_ext (the host-side test hook) instead of sim()
    bool structure_dirty{false};       // a value-stream column came or went (Poisson rate / random_mask set after load()): rebuild_device()
    std::vector<uint8_t> input_rewind; // per input: its train was replaced since the last commit
    static std::string class_key(const sanafe_hip_soma_class &c) { return std::string(reinterpret_cast<const char *>(&c), sizeof(c)); }
    int rebuild_device();
    // A value-stream column of slot `ls` comes (col != nullptr) or goes: the columns stay in slot order -- the order the
    // reference's sweep reaches the neurons; the device tables follow at the next commit (rebuild_device).
    void change_ext_column(uint32_t ls, const MappedChip::ExtColumn *col)
    {
        auto at = std::lower_bound(mc.ext.begin(), mc.ext.end(), ls, [](const MappedChip::ExtColumn &c, uint32_t slot) { return c.slot < slot; });
        if (col != nullptr) mc.ext.insert(at, *col);
        else if (at != mc.ext.end() && at->slot == ls) mc.ext.erase(at);
        // (slot_ext is rebuilt with the device; until then no lookup may trust it)
        mc.slot_ext.assign(mc.ext.empty() ? 0 : mc.n_slots, 0xffffffffu);
        uint64_t r = 0;
        for (size_t k = 0; k < mc.ext.size(); k++)
        {
            mc.slot_ext[mc.ext[k].slot] = static_cast<uint32_t>(k);
            if (mc.ext[k].kind == MappedChip::ExtColumn::TrueNorthRand) mc.ext[k].rand_index = r++;
        }
        mc.n_rand_global = r;
        structure_dirty = true;
    }
    int commit_attributes()
    {
        if (structure_dirty)
            if (int rc = rebuild_device()) return rc;
        if (classes_dirty)
        {
            if (sanafe_hip_write_soma_classes(dev, static_cast<uint32_t>(mc.soma_classes.size()), mc.soma_classes.data()) != 0)
                return fail(SANAFE_HIP_ERR_INVALID, sanafe_hip_last_error());
            classes_dirty = false;
        }
        std::sort(dirty_slots.begin(), dirty_slots.end());
        dirty_slots.erase(std::unique(dirty_slots.begin(), dirty_slots.end()), dirty_slots.end());
        for (size_t i = 0; i < dirty_slots.size();)
        {
            size_t j = i + 1;
            while (j < dirty_slots.size() && dirty_slots[j] == dirty_slots[j - 1] + 1) j++;
            if (sanafe_hip_write_slot_class(dev, dirty_slots[i], static_cast<uint32_t>(j - i), &mc.slot_cls[dirty_slots[i]]) != 0)
                return fail(SANAFE_HIP_ERR_INVALID, sanafe_hip_last_error());
            i = j;
        }
        dirty_slots.clear();
        if (inputs_dirty)
        {
            // compact the train storage (replaced trains leave holes), then replace the device tables
            std::vector<uint32_t> bits, beg(mc.in_train_beg.size());
            for (size_t a = 0; a < mc.in_train_beg.size(); a++)
            {
                const uint32_t nb = static_cast<uint32_t>(bits.size()) * 32u;
                beg[a] = nb;
                bits.resize(bits.size() + (mc.in_train_len[a] + 31u) / 32u, 0u);
                for (uint32_t b = 0; b < mc.in_train_len[a]; b++)
                {
                    const uint32_t src = mc.in_train_beg[a] + b;
                    if ((mc.in_train_bits[src >> 5] >> (src & 31u)) & 1u) bits[(nb + b) >> 5] |= 1u << ((nb + b) & 31u);
                }
            }
            mc.in_train_beg.swap(beg);
            mc.in_train_bits.swap(bits);
            if (sanafe_hip_write_inputs(dev, static_cast<uint32_t>(mc.in_train_beg.size()), mc.in_train_beg.data(), mc.in_train_len.data(),
                        mc.in_rate_period.data(), mc.in_train_bits.data(), mc.in_train_bits.size(), input_rewind.data()) != 0)
                return fail(SANAFE_HIP_ERR_INVALID, sanafe_hip_last_error());
            inputs_dirty = false;
            std::fill(input_rewind.begin(), input_rewind.end(), 0);
        }
        return 0;
    }
    // InputModel::set_attribute_neuron after load(), src/models.cpp:832-853
    int set_input_attribute(uint32_t ls, const std::string &key, int type, double num, const double *list, int64_t n_list)
    {
        const uint32_t a = mc.slot_aux[ls];
        if (a < mc.in_shared.size() && mc.in_shared[a])
            return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: attributes of neurons that share one `input` unit cannot change after "
                                                    "load() on the MI355X backend (the unit's train is interleaved over them)");
        if (input_rewind.size() != mc.in_train_beg.size()) input_rewind.assign(mc.in_train_beg.size(), 0);
        if (key == "spikes")
        {
            if (type != SANAFE_ATTR_LIST) return fail(SANAFE_HIP_ERR_INVALID, "Error: Attribute spikes is not a list");
            const uint32_t nb = static_cast<uint32_t>(mc.in_train_bits.size()) * 32u;
            mc.in_train_beg[a] = nb;
            mc.in_train_len[a] = static_cast<uint32_t>(n_list);
            mc.in_train_bits.resize(mc.in_train_bits.size() + (static_cast<size_t>(n_list) + 31) / 32, 0u);
            for (int64_t b = 0; b < n_list; b++)
                if (list[b] != 0.0) mc.in_train_bits[(nb + b) >> 5] |= 1u << ((nb + b) & 31u);
            input_rewind[a] = 1;
            inputs_dirty = true;
        }
        else if (key == "rate")
        {
            if (type != SANAFE_ATTR_DOUBLE && type != SANAFE_ATTR_INT) return fail(SANAFE_HIP_ERR_INVALID, "Error: Attribute rate cannot be cast to a double");
            int64_t period = 0;
            if (num > 0.0)
            {
                period = static_cast<long>(1.0 / num);
                if (period == 0) return fail(SANAFE_HIP_ERR_INVALID, "input rate > 1 makes the reference divide by zero (SURVEY quirk 14)");
            }
            mc.in_rate_period[a] = period;
            inputs_dirty = true;
        }
        else if (key == "poisson")
        {
            if (type != SANAFE_ATTR_DOUBLE && type != SANAFE_ATTR_INT) return fail(SANAFE_HIP_ERR_INVALID, "Error: Attribute poisson cannot be cast to a double");
            const bool has_column = !mc.slot_ext.empty() && mc.slot_ext[ls] != 0xffffffffu;
            if (has_column) mc.ext[mc.slot_ext[ls]].poisson = num;
            else if (num > 0.0)
            {
                // no column yet: the unit's generator has drawn at every update so far all the same (src/models.cpp:876)
                if (n_ranks != 1 || a >= mc.in_seed.size())
                    return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: an input neuron that had poisson == 0 at load() cannot get a "
                                                            "Poisson rate later on a tile-sharded chip");
                MappedChip::ExtColumn col;
                col.slot = ls;
                col.kind = MappedChip::ExtColumn::Poisson;
                col.poisson = num;
                col.seed = mc.in_seed[a];
                col.unit_key = mc.in_unit_key[a];
                col.skip_updates = total_timesteps + ext_hook_steps;
                change_ext_column(ls, &col);
            }
        }
        return 0;
    }
    // Generates and queues the external stream values of the next `steps` timesteps.
    int queue_ext(int64_t steps)
    {
        const size_t n = mc.ext.size();
        if (n == 0 || steps <= 0) return 0;
        ext_rows.resize(static_cast<size_t>(steps) * n);
        try
        {
            for (int64_t s = 0; s < steps; s++) ext.fill_row(ext_rows.data() + static_cast<size_t>(s) * n);
        }
        catch (const std::exception &e)
        {
            return fail(SANAFE_HIP_ERR_INVALID, e.what());
        }
        if (sanafe_hip_write_ext(dev, steps, ext_rows.data()) != 0) return fail(SANAFE_HIP_ERR_INVALID, sanafe_hip_last_error());
        return 0;
    }
    double total_energy{0.0}, total_sim_time{0.0};
    // Tile-sharded chips: a mapped-only, single-rank twin of the WHOLE chip (sanafe_chip_attach_whole).  The detailed NoC
    // schedule and the message trace are whole-chip host algorithms (one global event queue, src/schedule.cpp:208-292): every
    // rank gathers the statuses of all neurons and runs them on the twin's tables, as the reference runs them in its one process.
    sanafe_chip *whole{nullptr};
    // records of the last sim()
    bool have_records{false};
    bool rec_bits_global{false}; // rec_spike_bits rows cover the GLOBAL slots (records gathered from the ranks of a sharded chip)
    int64_t rec_first_timestep{0}, rec_count{0};
    std::vector<sanafe_hip_totals> rec_totals;
    std::vector<std::vector<Msg>> rec_messages; // per recorded step, per-source-core order
    std::vector<std::vector<uint32_t>> rec_spike_bits;
    std::vector<std::vector<double>> rec_optional; // per recorded step: the optional perf-trace columns (mc.log)
    // state log (potential / neuron traces): neurons whose potential / input current every recorded step keeps
    std::vector<int64_t> log_v_gids, log_u_gids;
    // tile-sharded chips: every rank logs the neurons it holds; per requested neuron (v list, then u list) the rank that
    // holds it and its column in that rank's rows -- the ranks' rows of a chunk are gathered and laid out in request order
    std::vector<int32_t> log_owner;
    std::vector<uint32_t> log_column;
    std::vector<uint32_t> log_rank_columns; // [n_ranks] columns (v + u) each rank logs
    std::vector<double> rec_state; // [recorded steps][log_v_gids + log_u_gids]
    // host copies of the per-slot cost classes (for generation delays)
    std::vector<double> slot_lat[3];
    std::vector<uint32_t> msg_ax_core; // per inbound axon of the message-pipeline cores on the device (msg_ax_*): index of its core

    // ---- cores that run on the host: soma inside the message pipeline, plugin synapse / dendrite units (host/host_cores.hpp) ----
    std::unique_ptr<sanafe_amd::HostCores> hcores;
    std::vector<uint32_t> hc_bits;                       // the step's spike bitmap for the replay of their message pipelines
    std::vector<sanafe_hip_host_core_costs> hc_costs;
    // ---- host-evaluated (plugin) soma units: plugin_get_hw, src/plugins.cpp:45-98 ----
    std::vector<std::unique_ptr<sanafe::PipelineUnit>> plugin_units; // parallel to mc.host_units
    std::vector<void *> plugin_handles;
    std::vector<uint32_t> h_slots, h_cores;
    std::vector<uint8_t> h_status, h_has;
    std::vector<double> h_cur, h_energy, h_latency;

    ~sanafe_chip()
    {
        delete whole;
        plugin_units.clear(); // destroy the objects before their code is unloaded
        for (void *h : plugin_handles) dlclose(h);
    }

    static sanafe::ModelAttribute to_model_attribute(const sanafe_desc &d, const sanafe_attr_table &t, int64_t i)
    {
        sanafe::ModelAttribute a;
        a.name = std::string(d.strings[t.key[i]]);
        a.forward_to_synapse = t.fwd ? (t.fwd[i] & SANAFE_FWD_SYNAPSE) != 0 : true;
        a.forward_to_dendrite = t.fwd ? (t.fwd[i] & SANAFE_FWD_DENDRITE) != 0 : true;
        a.forward_to_soma = t.fwd ? (t.fwd[i] & SANAFE_FWD_SOMA) != 0 : true;
        switch (t.type[i])
        {
        case SANAFE_ATTR_BOOL: a.value = (t.num[i] != 0.0); break;
        case SANAFE_ATTR_INT: a.value = static_cast<int>(t.num[i]); break;
        case SANAFE_ATTR_DOUBLE: a.value = t.num[i]; break;
        case SANAFE_ATTR_STRING: a.value = std::string(t.str[i] >= 0 ? d.strings[t.str[i]] : ""); break;
        default:
        {
            std::vector<sanafe::ModelAttribute> v;
            for (int64_t k = t.list_ptr[i]; k < t.list_ptr[i + 1]; k++)
            {
                sanafe::ModelAttribute e;
                const double x = t.list_num[k];
                if (x == static_cast<double>(static_cast<int>(x))) e.value = static_cast<int>(x);
                else e.value = x;
                v.push_back(e);
            }
            a.value = v;
        }
        }
        return a;
    }

    void load_plugins(const sanafe_desc &d)
    {
        std::map<std::string, void *> libs;
        for (const MappedChip::HostUnit &hu : mc.host_units)
        {
            void *&lib = libs[hu.plugin_path];
            if (!lib)
            {
                lib = dlopen(hu.plugin_path.c_str(), RTLD_LAZY | RTLD_LOCAL);
                if (!lib) throw std::runtime_error(std::string("Error: Couldn't load library ") + hu.plugin_path + ": " + dlerror());
                plugin_handles.push_back(lib);
            }
            using Factory = sanafe::PipelineUnit *(*) ();
            const std::string sym = "create_" + hu.model;
            dlerror();
            auto create = reinterpret_cast<Factory>(dlsym(lib, sym.c_str()));
            if (!create) throw std::runtime_error("Error: Couldn't load symbol " + sym + " from " + hu.plugin_path);
            std::unique_ptr<sanafe::PipelineUnit> unit(create());
            if (!unit->implements_soma || unit->implements_synapse || unit->implements_dendrite)
                throw std::runtime_error("plugin unit '" + hu.name + "' must implement the soma interface only");
            unit->name = hu.name;
            unit->model = hu.model;
            unit->plugin_lib = hu.plugin_path;
            for (int64_t i = d.unit_attr_ptr[hu.desc_unit]; i < d.unit_attr_ptr[hu.desc_unit + 1]; i++)
            {
                const sanafe::ModelAttribute a = to_model_attribute(d, d.unit_attrs, i);
                unit->model_attributes[*a.name] = a;
            }
            for (const auto &kv : unit->model_attributes) unit->set_attribute_hw(kv.first, kv.second); // key order
            plugin_units.push_back(std::move(unit));
        }
        for (const MappedChip::HostNeuron &hn : mc.host_neurons)
        {
            sanafe::PipelineUnit &u = *plugin_units[hn.unit];
            u.add_neuron();
            for (int64_t i = d.neuron_attr_ptr[hn.gid]; i < d.neuron_attr_ptr[hn.gid + 1]; i++)
            {
                const sanafe::ModelAttribute a = to_model_attribute(d, d.neuron_attrs, i);
                if (a.forward_to_soma) u.set_attribute_neuron(hn.addr, *a.name, a);
            }
            h_slots.push_back(hn.slot);
            h_cores.push_back(hn.core_local);
        }
        const size_t n = mc.host_neurons.size();
        h_status.assign(n, 0);
        h_has.assign(n, 0);
        h_cur.assign(n, 0.0);
        h_energy.assign(n, 0.0);
        h_latency.assign(n, 0.0);
    }

    // One timestep of the host-evaluated somas: soma `update`, then the default costing of
    // process_soma_output (src/pipeline.hpp:453-458, 631-714).
    void run_plugins(int64_t timestep)
    {
        for (size_t i = 0; i < mc.host_neurons.size(); i++)
        {
            const MappedChip::HostNeuron &hn = mc.host_neurons[i];
            const MappedChip::HostUnit &hu = mc.host_units[hn.unit];
            sanafe::PipelineUnit &u = *plugin_units[hn.unit];
            const std::optional<double> cur = h_has[i] ? std::optional<double>(h_cur[i]) : std::nullopt;
            sanafe::PipelineResult r = u.update(static_cast<size_t>(hn.addr), cur, static_cast<long int>(timestep));
            if (r.status == sanafe::neuron_state_unset) throw std::runtime_error("Soma output; should return valid neuron state.");
            if (r.energy.has_value() && hu.has_energy)
                throw std::runtime_error("Error: Soma unit simulates energy and also has default energy metrics set. Remove the default energy metrics from the architecture description.");
            if (r.latency.has_value() && hu.has_latency)
                throw std::runtime_error("Error: Soma unit simulates latency and also has default latency costs set. Remove the default latency metrics from the architecture description");
            const int k = static_cast<int>(r.status) - 1;
            if (hu.has_energy) r.energy = hu.energy[k];
            if (hu.has_latency) r.latency = hu.latency[k];
            if (!r.energy.has_value()) throw std::runtime_error("Soma unit does not simulate energy or provide default energy costs in the architecture description.");
            if (!r.latency.has_value()) throw std::runtime_error("Soma unit does not simulate latency or provide default latency costs in the architecture description.");
            h_status[i] = static_cast<uint8_t>(r.status);
            h_energy[i] = *r.energy;
            h_latency[i] = *r.latency;
            if (!slot_lat[0].empty()) // exact generation delays (detailed model): dendrite of the neuron pipeline + the soma
            {
                const sanafe_hip_cost_class &cc = mc.cost_classes[(mc.slot_cls[hn.slot] >> 6) & 1023u];
                slot_lat[k][mc.slot_offset + hn.slot] = (0.0 + cc.dendrite_latency) + *r.latency;
            }
        }
    }

    // ------------------------------------------------------------------------------
    // Rebuild one timestep's messages from the spike bitmap + per-slot status.
    // process_neurons / pipeline_process_axon_out / receive_message, src/chip.cpp:624-654,
    // 694-708, 802-834: per core, neurons in mapped order; latencies accumulate into
    // next_message_generation_delay; the first message of a firing neuron carries it.
    // ------------------------------------------------------------------------------
    template <typename M>
    void build_messages(int64_t timestep, const std::vector<uint8_t> &status, std::vector<std::vector<M>> &per_core,
            int64_t mid_base, const uint16_t *msg_fired = nullptr) const
    {
        constexpr bool FULL = std::is_same<M, Msg>::value; // the traced record carries every field of `Message`
        int64_t next_mid = mid_base;
        per_core.resize(mc.n_cores);
        for (auto &q : per_core) q.clear(); // capacity is kept: the caller reuses the queues from step to step
        for (uint32_t c = 0; c < mc.n_cores; c++)
        {
            double next_delay = 0.0;
            const uint32_t base = mc.core_nbase[c];
            const uint32_t st = mc.core_tile[c];
            {
                size_t n_msgs = 1; // + the placeholder
                for (uint32_t k = 0; k < mc.core_ncount[c]; k++)
                    if (status[base + k] == 3) n_msgs += mc.out_ptr[base + k + 1] - mc.out_ptr[base + k];
                per_core[c].reserve(n_msgs);
            }
            const double lat_access = mc.core_axon_out_latency[c];
            int64_t last_gid = -1;
            for (uint32_t k = 0; k < mc.core_ncount[c]; k++)
            {
                const uint32_t s = base + k;
                const uint8_t stat = status[s];
                if (stat == 0) continue;
                last_gid = mc.gid_of_slot[s];
                next_delay += slot_lat[stat - 1][s];
                if (stat != 3) continue;
                for (uint64_t o = mc.out_ptr[s]; o < mc.out_ptr[s + 1]; o++)
                {
                    const uint64_t a = mc.out_axon[o];
                    M m{};
                    const uint32_t dc = mc.ax_dest_core[a];
                    const uint32_t dt = mc.core_tile[dc];
                    const bool into_msg_core = a >= mc.n_device_axons; // a core whose soma is part of the message pipeline
                    if constexpr (FULL)
                    {
                        m.timestep = timestep;
                        m.mid = next_mid++;
                        m.src_neuron = mc.gid_of_slot[s];
                        m.src_tile = st;
                        m.dest_tile = dt;
                        m.dest_core_offset = mc.core_offset[dc];
                        m.dest_axon_id = mc.ax_dest_axon_id[a];
                        m.spikes = into_msg_core ? mc.msg_ax_nsyn[a - mc.n_device_axons] : mc.ax_nsyn[a];
                    }
                    m.src_core_id = c;
                    m.src_core_offset = mc.core_offset[c];
                    m.src_x = mc.tile_x[st];
                    m.src_y = mc.tile_y[st];
                    m.dest_core_id = dc;
                    m.dest_x = mc.tile_x[dt];
                    m.dest_y = mc.tile_y[dt];
                    m.hops = mc.ax_hops[a];
                    m.placeholder = 0;
                    m.generation_delay = next_delay + lat_access;
                    next_delay = 0.0;
                    if (!into_msg_core) m.processing_delay = mc.ax_proc_delay[a];
                    else
                    {
                        // process_message, src/chip.cpp:738-789: the axon-in latency, then per synaptic event the synapse's, the
                        // dendrite's and the soma's latency BY THE STATUS ITS UPDATE RETURNED -- the device counted, per message,
                        // the updates that fired (msgsoma_kernel)
                        const uint64_t i = a - mc.n_device_axons;
                        const sanafe_hip_msg_core_costs &k = mc.msg_costs[msg_ax_core[i]];
                        if (msg_fired == nullptr) throw std::logic_error("messages into a message-pipeline core need the step's fired counts");
                        const uint32_t n_ev = mc.msg_ax_nsyn[i], n_fired = std::min<uint32_t>(msg_fired[i], n_ev);
                        const double updated = (0.0 + k.synapse_latency + k.dendrite_latency) + (k.soma_latency[0] + k.soma_latency[1]);
                        const double fired = (0.0 + k.synapse_latency + k.dendrite_latency) + ((k.soma_latency[0] + k.soma_latency[1]) + k.soma_latency[2]);
                        double lat = k.axon_in_latency;
                        for (uint32_t q = 0; q < n_ev - n_fired; q++) lat += updated;
                        for (uint32_t q = 0; q < n_fired; q++) lat += fired;
                        m.processing_delay = 0.0 + lat;
                    }
                    m.min_hop_delay = mc.ax_min_hop_delay[a];
                    m.sent_timestamp = m.received_timestamp = m.processed_timestamp = NEG_INF;
                    per_core[c].push_back(m);
                }
            }
            if (next_delay != 0.0) // placeholder, src/chip.cpp:640-652
            {
                M m{};
                if constexpr (FULL)
                {
                    m.timestep = timestep;
                    m.mid = -1;
                    m.src_neuron = last_gid;
                    m.src_tile = st;
                }
                m.src_core_id = c;
                m.src_core_offset = mc.core_offset[c];
                m.src_x = mc.tile_x[st];
                m.src_y = mc.tile_y[st];
                m.placeholder = 1;
                m.generation_delay = next_delay;
                m.sent_timestamp = m.received_timestamp = m.processed_timestamp = NEG_INF;
                per_core[c].push_back(m);
            }
        }
    }

    // ------------------------------------------------------------------------------
    // Optional perf-trace columns of one timestep (sim_trace_get_optional_traces, src/chip.cpp:1541-1579) from the
    // status of every slot.  Per-unit energy is accumulated by the reference one pipeline call at a time
    // (PipelineUnit::process, src/pipeline.cpp:87-105): first every neuron of every core in order, then every
    // message in (source core, neuron, destination core) order with its synapses in connection order -- the same
    // sequence of additions is replayed here.  unit.latency accumulates ENERGY (src/pipeline.cpp:102), so a
    // `.latency` column repeats the `.energy` value.  Core and tile energies follow sim_calculate_core_energy /
    // _tile_energy (src/chip.cpp:1188-1261); a tile's hop energy is summed per message here (per direction there).
    // ------------------------------------------------------------------------------
    std::vector<double> optional_columns(const uint8_t *status, const uint16_t *msg_fired = nullptr) const
    {
        const MappedChip::LogPlan &lg = mc.log;
        std::vector<double> unit_e(lg.unit_used.size(), 0.0), tile_hop(mc.n_tiles, 0.0);
        std::vector<int64_t> msgs_in(mc.n_cores, 0), packets_out(mc.n_cores, 0);
        for (uint32_t c = 0; c < mc.n_cores; c++) // neuron processing
            for (uint32_t k = 0; k < mc.core_ncount[c]; k++)
            {
                const uint32_t s = mc.core_nbase[c] + k;
                const uint8_t stt = status[s];
                if (stt == 0) continue;
                const sanafe_hip_cost_class &cc = mc.cost_classes[(mc.slot_cls[s] >> 6) & 1023u];
                if (lg.core_bp[c] <= SANAFE_BUF_INSIDE_DENDRITE) unit_e[lg.core_unit_beg[c] + lg.slot_dend_unit[s]] += cc.dendrite_energy;
                unit_e[lg.core_unit_beg[c] + lg.slot_soma_unit[s]] += cc.soma_energy[stt - 1];
            }
        for (uint32_t c = 0; c < mc.n_cores; c++) // message processing
            for (uint32_t k = 0; k < mc.core_ncount[c]; k++)
            {
                const uint32_t s = mc.core_nbase[c] + k;
                if (status[s] != 3) continue;
                for (uint64_t o = mc.out_ptr[s]; o < mc.out_ptr[s + 1]; o++)
                {
                    const uint64_t a = mc.out_axon[o];
                    const uint32_t dc = mc.ax_dest_core[a];
                    packets_out[c]++;
                    msgs_in[dc]++;
                    tile_hop[mc.core_tile[dc]] += lg.ax_e_hop[a];
                    if (a >= mc.n_device_axons)
                    {
                        // into a core whose soma is part of the message pipeline (on the device): per synaptic event the
                        // synapse, the dendrite and the soma unit, the soma by the status its update returned -- the device
                        // counted the updates that fired per message (the reference interleaves them: same terms, other order)
                        const uint64_t i = a - mc.n_device_axons;
                        const uint32_t k = msg_ax_core[i];
                        const sanafe_hip_msg_core_costs &mk = mc.msg_costs[k];
                        const uint32_t ub = lg.core_unit_beg[dc], n_ev = mc.msg_ax_nsyn[i];
                        const uint32_t n_fired = msg_fired ? std::min<uint32_t>(msg_fired[i], n_ev) : 0u;
                        const double updated = mk.soma_energy[0] + mk.soma_energy[1], fired = updated + mk.soma_energy[2];
                        for (uint32_t q = 0; q < n_ev; q++)
                        {
                            unit_e[ub + mc.msg_units[k][0]] += mk.synapse_energy;
                            unit_e[ub + mc.msg_units[k][1]] += mk.dendrite_energy;
                            unit_e[ub + mc.msg_units[k][2]] += q < n_ev - n_fired ? updated : fired;
                        }
                        continue;
                    }
                    const uint64_t s0 = mc.core_syn_base[dc] + mc.ax_syn_beg[a];
                    for (uint32_t q = 0; q < mc.ax_nsyn[a]; q++)
                    {
                        const uint16_t pair = lg.syn_units[s0 + q];
                        unit_e[lg.core_unit_beg[dc] + (pair & 0xffu)] += lg.unit_e_spike[lg.core_unit_beg[dc] + (pair & 0xffu)];
                        if (lg.core_bp[dc] > SANAFE_BUF_BEFORE_DENDRITE)
                            unit_e[lg.core_unit_beg[dc] + (pair >> 8)] += lg.unit_e_update[lg.core_unit_beg[dc] + (pair >> 8)];
                    }
                }
            }
        std::vector<double> core_e(mc.n_cores, 0.0), tile_e(tile_hop);
        for (uint32_t c = 0; c < mc.n_cores; c++)
        {
            double e = static_cast<double>(msgs_in[c]) * lg.core_e_ain[c], pipeline = 0.0;
            for (uint32_t u = lg.core_unit_beg[c]; u < lg.core_unit_beg[c + 1]; u++)
                if (lg.unit_used[u]) pipeline += unit_e[u];
            e += pipeline;
            // AxonOutUnit::energy accumulates energy_access once per packet, src/chip.cpp:826-829
            double aout = 0.0;
            for (int64_t p = 0; p < packets_out[c]; p++) aout += lg.core_e_aout[c];
            e += aout;
            core_e[c] = e;
            tile_e[mc.core_tile[c]] += e;
        }
        std::vector<double> out;
        out.reserve(lg.columns.size());
        for (const MappedChip::LogPlan::Column &col : lg.columns)
            out.push_back(col.kind == 0 ? tile_e[col.tile] : col.kind == 1 ? core_e[col.core] : unit_e[lg.core_unit_beg[col.core] + col.unit]);
        return out;
    }

    // ---- detailed timing model: src/schedule.cpp:208-620 ----
    struct Noc
    {
        size_t w, h, max_cpt;
        std::vector<double> density, core_finished;
        double mean_delay{0.0};
        long in_noc{0};
        size_t idx(size_t x, size_t y, size_t link) const
        {
            const size_t lpr = max_cpt + 4; // src/schedule.hpp:190-195
            return (x * h * lpr) + (y * lpr) + link;
        }
    };
    template <typename M, typename F> static void walk_route(const Noc &noc, const M &m, F &&visit)
    {
        // dimension-order route; NocInfo::update_message_density / calculate_route_congestion,
        // src/schedule.cpp:478-611 (directions: north 0, east 1, south 2, west 3).  The cell index
        // idx(x, y, link) = (x * h + y) * links + link is advanced by its strides instead of recomputed.
        const int64_t sx = m.src_x, sy = m.src_y, dx = m.dest_x, dy = m.dest_y;
        const int64_t lpr = static_cast<int64_t>(noc.max_cpt) + 4, stride_x = static_cast<int64_t>(noc.h) * lpr;
        const size_t own = 4 + m.src_core_offset;
        size_t prev = own;
        int64_t cell = sx * stride_x + sy * lpr; // (x, y) of the walk, link 0
        if (sx != dx)
        {
            const int64_t step = (sx < dx) ? stride_x : -stride_x;
            const size_t dir = (sx < dx) ? 1 : 3;
            visit(static_cast<size_t>(cell) + own);
            cell += step;
            for (int64_t x = sx + ((sx < dx) ? 1 : -1); x != dx; x += (sx < dx) ? 1 : -1, cell += step) visit(static_cast<size_t>(cell) + dir);
            prev = dir;
        }
        if (sy != dy)
        {
            const int64_t step = (sy < dy) ? lpr : -lpr;
            const size_t dir = (sy < dy) ? 0 : 2;
            visit(static_cast<size_t>(cell) + prev); // at (dest_x, src_y): the own link when there was no x leg
            cell += step;
            for (int64_t y = sy + ((sy < dy) ? 1 : -1); y != dy; y += (sy < dy) ? 1 : -1, cell += step) visit(static_cast<size_t>(cell) + dir);
            prev = dir;
        }
        visit(static_cast<size_t>(cell) + prev);
    }
    template <typename M> static void check_bounds(const Noc &noc, const M &m)
    {
        if (static_cast<size_t>(m.src_x) > noc.w || static_cast<size_t>(m.dest_x) > noc.w)
            throw std::runtime_error("Message x > NoC width");
        if (static_cast<size_t>(m.src_y) > noc.h || static_cast<size_t>(m.dest_y) > noc.h)
            throw std::runtime_error("Message y > NoC height");
    }
    template <typename M> static void track(Noc &noc, const M &m, bool entering)
    {
        check_bounds(noc, m);
        double adjust = 1.0 / (2.0 + static_cast<double>(m.hops));
        if (!entering) adjust *= -1.0;
        walk_route(noc, m, [&](size_t i) { noc.density[i] += adjust; });
        rolling_average(noc, m, entering);
    }
    template <typename M> static void rolling_average(Noc &noc, const M &m, bool entering)
    {
        if (entering) // update_rolling_averages, src/schedule.cpp:449-476
        {
            noc.mean_delay += (m.processing_delay - noc.mean_delay) / (static_cast<double>(noc.in_noc) + 1.0);
            noc.in_noc++;
        }
        else
        {
            if (noc.in_noc > 1) noc.mean_delay += (noc.mean_delay - m.processing_delay) / (static_cast<double>(noc.in_noc) - 1.0);
            else noc.mean_delay = 0.0;
            noc.in_noc--;
        }
    }
    // schedule_messages_timestep_detailed, src/schedule.cpp:208-292, with the same arithmetic in the same order but
    // without its O(cores) scan per scheduled message: the reference walks every core's list of in-flight messages
    // at each pop (noc_update_all_tracked_messages, :380-400) and retires those received by `now`, in (core, list)
    // order.  Here the in-flight messages sit in a min-heap on their received time; the ones due are popped,
    // put back into (destination core, arrival) order and retired -- the identical sequence of density and
    // rolling-average updates, so every timestamp is bit-identical (checked against the oracle, which keeps the scan).
    // Scratch of one scheduler thread, reused from timestep to timestep (no allocation in the steady state).
    template <typename M> struct SchedScratch
    {
        Noc noc;
        std::vector<size_t> head;
        std::vector<std::vector<M>> per_core;
    };
    template <typename M> double schedule_detailed(std::vector<std::vector<M>> &per_core, bool keep_order, SchedScratch<M> &scratch) const
    {
        using Msg = M; // (the body below is written against the member names both message types share)
        Noc &noc = scratch.noc;
        noc.w = mc.noc_width;
        noc.h = mc.noc_height;
        noc.max_cpt = mc.max_cores_per_tile;
        noc.mean_delay = 0.0;
        noc.in_noc = 0;
        noc.core_finished.assign(mc.n_cores, 0.0);
        noc.density.assign(static_cast<size_t>(mc.noc_height) * mc.noc_width * (4 + mc.max_cores_per_tile), 0.0);
        std::vector<size_t> &head = scratch.head;
        head.assign(mc.n_cores, 0);
        std::vector<std::vector<const Msg *>> sched_order(keep_order ? mc.n_cores : 0); // pop order per source core
        // The send queue holds handles; its order depends only on the comparisons, which are the reference's
        // (CompareMessagesBySentTime, src/message.cpp:61-65), so ties break exactly as there.
        struct Pending
        {
            double sent;
            Msg *m;
        };
        struct BySentHandle
        {
            bool operator()(const Pending &a, const Pending &b) const noexcept { return a.sent > b.sent; }
        };
        std::priority_queue<Pending, std::vector<Pending>, BySentHandle> pq;
        // messages are scheduled in place, in their source core's FIFO (the vectors do not move while this runs)
        struct InFlight // 32 bytes: everything retirement needs without touching the message again but for its route
        {
            double received;
            double adjust;   // the density share 1 / (2 + hops) this message added along its route
            const Msg *m;
            uint32_t seq;    // arrival order at its destination core's list
            uint32_t dest;   // destination core
        };
        // 4-ary min-heap on the received time: half the levels of a binary heap for the few hundred to few thousand
        // messages in flight.  Which of several equal keys pops first does not matter: what is due is re-ordered below.
        struct Heap
        {
            std::vector<InFlight> v;
            bool empty() const { return v.empty(); }
            const InFlight &top() const { return v.front(); }
            void push(const InFlight &e)
            {
                size_t i = v.size();
                v.push_back(e);
                while (i > 0)
                {
                    const size_t p = (i - 1) >> 2;
                    if (!(v[p].received > e.received)) break;
                    v[i] = v[p];
                    i = p;
                }
                v[i] = e;
            }
            void pop()
            {
                const InFlight e = v.back();
                v.pop_back();
                const size_t n = v.size();
                if (n == 0) return;
                size_t i = 0;
                for (;;)
                {
                    const size_t c0 = 4 * i + 1;
                    if (c0 >= n) break;
                    size_t best = c0;
                    const size_t c1 = std::min(n, c0 + 4);
                    for (size_t c = c0 + 1; c < c1; c++)
                        if (v[c].received < v[best].received) best = c;
                    if (!(v[best].received < e.received)) break;
                    v[i] = v[best];
                    i = best;
                }
                v[i] = e;
            }
        } in_flight;
        std::vector<InFlight> due;
        uint32_t seq = 0;
        for (uint32_t c = 0; c < mc.n_cores; c++)
            if (!per_core[c].empty())
            {
                Msg &m = per_core[c][head[c]++];
                m.sent_timestamp = m.generation_delay;
                pq.push(Pending{m.sent_timestamp, &m});
            }
        double last = 0.0;
        while (!pq.empty())
        {
            Msg &m = *pq.top().m;
            pq.pop();
            last = std::max(last, m.sent_timestamp);
            const double tnow = m.sent_timestamp;
            due.clear();
            while (!in_flight.empty() && tnow >= in_flight.top().received)
            {
                due.push_back(in_flight.top());
                in_flight.pop();
            }
            if (due.size() > 1)
                std::sort(due.begin(), due.end(), [](const InFlight &a, const InFlight &b) { return a.dest != b.dest ? a.dest < b.dest : a.seq < b.seq; });
            for (const InFlight &r : due) // NocInfo::update_message_density(leaving) + update_rolling_averages
            {
                const double leave = r.adjust * -1.0;
                walk_route(noc, *r.m, [&](size_t i) { noc.density[i] += leave; });
                rolling_average(noc, *r.m, false);
            }
            if (!m.placeholder) // schedule_handle_message
            {
                const size_t dc = m.dest_core_id;
                // one walk reads the densities along the route (calculate_route_congestion) and adds this message's own
                // share to them (update_message_density): each cell is read before it is changed, as in two walks
                check_bounds(noc, m);
                const double adjust = 1.0 / (2.0 + static_cast<double>(m.hops));
                double flow = 0.0;
                walk_route(noc, m, [&](size_t i) {
                    flow += noc.density[i];
                    noc.density[i] += adjust;
                });
                m.messages_along_route = flow;
                const double cap = static_cast<double>((m.hops + 1UL) * mc.noc_buffer);
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
                in_flight.push(InFlight{m.received_timestamp, adjust, &m, seq++, static_cast<uint32_t>(dc)});
                rolling_average(noc, m, true);
                last = std::max(last, m.processed_timestamp);
            }
            const size_t sc = m.src_core_id;
            if (head[sc] < per_core[sc].size()) // schedule_push_next_message
            {
                Msg &nx = per_core[sc][head[sc]++];
                nx.sent_timestamp = m.sent_timestamp + nx.generation_delay;
                pq.push(Pending{nx.sent_timestamp, &nx});
                last = std::max(last, nx.sent_timestamp);
            }
            if (keep_order) sched_order[sc].push_back(&m);
        }
        if (keep_order) // the message trace lists a core's messages in the order they were scheduled
        {
            std::vector<std::vector<Msg>> scheduled(mc.n_cores);
            for (uint32_t c = 0; c < mc.n_cores; c++)
            {
                scheduled[c].reserve(sched_order[c].size());
                for (const Msg *pm : sched_order[c]) scheduled[c].push_back(*pm);
            }
            per_core.swap(scheduled);
        }
        return last + mc.sync_delay;
    }
};

extern "C" const char *sanafe_last_error(void) { return g_err.c_str(); }

extern "C" int sanafe_chip_create(const sanafe_desc *desc, int device, int n_ranks, int rank, sanafe_chip **out)
{
    if (!desc || !out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    *out = nullptr;
    auto chip = std::make_unique<sanafe_chip>();
    try
    {
        // SANAFE_TARGET_SLICES: delivery work items to aim for (tests use it to force multi-slice cores)
        uint32_t target_slices = 6144; // 4 x the 1,536 delivery workgroups resident at six wavefronts per SIMD (bitmap records; no
                                       // partial last round), few enough that a big core's write-back is shared by 6-12 slices
                                       // (C3 1,024 x 256: 3,072 / 6,144 slices 0.281 ms, 4,096 0.294, 8,192 0.308; delta records
                                       // and formats 0 / 6: 6,144 is 1-5 % faster than 8,192 as well)
        if (const char *env = std::getenv("SANAFE_TARGET_SLICES")) target_slices = static_cast<uint32_t>(std::max(1L, std::atol(env)));
        uint32_t min_slice_axons = 1024; // one 256-axon chunk per wavefront: small chips still spread over many CUs
        if (const char *env = std::getenv("SANAFE_MIN_SLICE_AXONS")) min_slice_axons = static_cast<uint32_t>(std::max(4L, std::atol(env)));
        sanafe_amd::map_and_lower(*desc, n_ranks, rank, target_slices, min_slice_axons, chip->mc);
    }
    catch (const sanafe_amd::UnsupportedError &e)
    {
        return fail(SANAFE_HIP_ERR_UNSUPPORTED, std::string("UnsupportedError: ") + e.what());
    }
    catch (const sanafe_amd::HardwareMappingError &e)
    {
        return fail(SANAFE_HIP_ERR_INVALID, std::string("HardwareMappingError: ") + e.what());
    }
    catch (const std::exception &e)
    {
        return fail(SANAFE_HIP_ERR_INVALID, e.what());
    }
    chip->n_neurons = desc->n_neurons;
    chip->device = device;
    chip->n_ranks = n_ranks;
    chip->rank = rank;
    chip->xc.n_ranks = n_ranks;
    chip->xc.rank = rank;
    chip->xc.slot_begin = chip->mc.rank_slot_begin;
    if (n_ranks > 1 && (chip->mc.rank_slot_begin[rank] != chip->mc.slot_offset ||
                               chip->mc.rank_slot_begin[rank + 1] - chip->mc.rank_slot_begin[rank] != chip->mc.n_slots))
        return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: a rank of a tile-sharded chip holds no mapped neurons");
    try
    {
        chip->ext.init(chip->mc); // opens the LIF noise files: a missing file fails the load, as in the reference
    }
    catch (const std::exception &e)
    {
        return fail(SANAFE_HIP_ERR_INVALID, e.what());
    }
    const sanafe_hip_image im = chip->mc.image();
    if (device >= 0) DEV(sanafe_hip_chip_create(&im, device, &chip->dev)); // device < 0: map only (CPU-side checks)
    // per-slot neuron-pipeline latency by status, for the host-side generation delays
    const MappedChip &mc = chip->mc;
    if (n_ranks == 1)
    {
        for (int k = 0; k < 3; k++) chip->slot_lat[k].assign(mc.n_global_slots, 0.0);
        for (uint32_t s = 0; s < mc.n_slots; s++)
        {
            const uint32_t cls = mc.slot_cls[s];
            if ((cls & 7u) == SANAFE_SOMA_NONE) continue;
            const sanafe_hip_cost_class &cc = mc.cost_classes[(cls >> 6) & 1023u];
            for (int k = 0; k < 3; k++) chip->slot_lat[k][s] = (0.0 + cc.dendrite_latency) + cc.soma_latency[k];
        }
    }
    if (chip->mc.msg_on_device)
        for (size_t k = 0; k + 1 < mc.msg_ax_beg.size(); k++) chip->msg_ax_core.insert(chip->msg_ax_core.end(), mc.msg_ax_beg[k + 1] - mc.msg_ax_beg[k], static_cast<uint32_t>(k));
    if (!chip->mc.host_cores.empty() && !chip->mc.msg_on_device) // (msg_on_device: these cores run on the device, msgsoma_kernel)
    {
        try
        {
            chip->hcores = std::make_unique<sanafe_amd::HostCores>(chip->mc, *desc);
        }
        catch (const std::exception &e)
        {
            sanafe_hip_chip_destroy(chip->dev);
            chip->dev = nullptr;
            return fail(SANAFE_HIP_ERR_INVALID, e.what());
        }
    }
    if (!chip->mc.host_units.empty())
    {
        try
        {
            chip->load_plugins(*desc);
        }
        catch (const std::exception &e)
        {
            sanafe_hip_chip_destroy(chip->dev);
            chip->dev = nullptr;
            return fail(SANAFE_HIP_ERR_INVALID, e.what());
        }
    }
    *out = chip.release();
    return 0;
}

// A structural patch (a value-stream column came or went): the device chip is created again from the patched tables and the
// run-time state moves over slot by slot -- the layout is the same.  The generators and file positions of the value streams
// stay where they are (ExtStreams::carry_from); a new Poisson generator skips the draws its unit has made so far.
int sanafe_chip::rebuild_device()
{
    if (!dev) // mapped only (host-side checks): the value streams follow the new columns, there is nothing else to re-create
    {
        try
        {
            ExtStreams streams;
            streams.init(mc);
            streams.carry_from(ext);
            ext = std::move(streams);
        }
        catch (const std::exception &e)
        {
            return fail(SANAFE_HIP_ERR_INVALID, e.what());
        }
        structure_dirty = false;
        return 0;
    }
    if (n_ranks != 1 || xc.kind != sanafe_amd::Exchange::None || !mc.host_neurons.empty() || hcores || !mc.tap_slot.empty())
        return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: a Poisson rate / random_mask that comes or goes after load() needs a single-rank "
                                                "chip without plugin / host-side units or `taps` dendrites");
    const size_t n = mc.n_slots;
    std::vector<double> v(n, 0.0), u(n, 0.0), ring(static_cast<size_t>(mc.ring_slots) * n, 0.0);
    std::vector<int32_t> rf(n, 0);
    std::vector<uint8_t> stt(n, 0), rv(static_cast<size_t>(mc.ring_slots) * n, 0), arr(n, 0);
    std::vector<uint32_t> last(n, 0), pos(std::max<size_t>(1, mc.in_train_beg.size()), 0);
    sanafe_hip_state s{0, v.data(), u.data(), rf.data(), stt.data(), ring.data(), rv.data(), arr.data(), last.data(), pos.data()};
    if (sanafe_hip_export_state(dev, &s) != 0) return fail(SANAFE_HIP_ERR_INVALID, sanafe_hip_last_error());
    if (inputs_dirty) // replaced trains start over; the new image holds the patched trains already
        for (size_t a = 0; a < input_rewind.size() && a < pos.size(); a++)
            if (input_rewind[a]) pos[a] = 0;
    const sanafe_hip_image im = mc.image();
    sanafe_hip_chip *fresh = nullptr;
    if (sanafe_hip_chip_create(&im, device, &fresh) != 0) return fail(SANAFE_HIP_ERR_INVALID, sanafe_hip_last_error());
    if (sanafe_hip_import_state(fresh, &s) != 0)
    {
        sanafe_hip_chip_destroy(fresh);
        return fail(SANAFE_HIP_ERR_INVALID, sanafe_hip_last_error());
    }
    sanafe_hip_chip_destroy(dev);
    dev = fresh;
    try
    {
        ExtStreams streams;
        streams.init(mc);
        streams.carry_from(ext);
        ext = std::move(streams);
    }
    catch (const std::exception &e)
    {
        return fail(SANAFE_HIP_ERR_INVALID, e.what());
    }
    structure_dirty = classes_dirty = inputs_dirty = false; // (the new image holds every patch made so far)
    dirty_slots.clear();
    std::fill(input_rewind.begin(), input_rewind.end(), 0);
    if (!log_v_gids.empty() || !log_u_gids.empty())
    {
        const std::vector<int64_t> lv = log_v_gids, lu = log_u_gids;
        if (int rc = sanafe_chip_set_state_log(this, static_cast<int64_t>(lv.size()), lv.data(), static_cast<int64_t>(lu.size()), lu.data())) return rc;
    }
    return 0;
}

// load(net, overwrite=false) on a chip that has already simulated timesteps (src/chip.cpp:129-138: the new neurons are
// mapped next to the programmed ones, every unit keeps its state): the combined network was lowered into `to`; the
// programmed neurons keep their global ids, so their state moves slot by slot -- potentials, LIF input currents,
// refractory counters, statuses, the time-step buffers / delay lines (by the future step a row belongs to), spike-train
// cursors -- and the step numbering continues.
extern "C" int sanafe_chip_carry_state(sanafe_chip *to, sanafe_chip *from)
{
    if (!to || !from || !to->dev || !from->dev) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    const MappedChip &a = from->mc, &b = to->mc;
    if (from->n_ranks != 1 || to->n_ranks != 1) return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: adding a network to a tile-sharded chip after timesteps have run");
    for (const sanafe_chip *c : {from, to})
        if (!c->mc.host_neurons.empty() || c->hcores || !c->mc.tap_slot.empty())
            return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: load(net, overwrite=False) after timesteps have been simulated is not "
                                                    "available with plugin / host-side units or `taps` dendrites "
                                                    "(their state lives in host objects that a new lowering re-creates)");
    if (from->n_neurons > to->n_neurons) return fail(SANAFE_HIP_ERR_INVALID, "the new chip does not hold the programmed neurons");
    auto buffers = [](const MappedChip &m, std::vector<double> &v, std::vector<double> &u, std::vector<int32_t> &rf, std::vector<uint8_t> &stt,
                           std::vector<double> &ring, std::vector<uint8_t> &rv, std::vector<uint8_t> &arr, std::vector<uint32_t> &last,
                           std::vector<uint32_t> &pos, sanafe_hip_state &s) {
        const size_t n = m.n_slots;
        v.assign(n, 0.0);
        u.assign(n, 0.0);
        rf.assign(n, 0);
        stt.assign(n, 0);
        ring.assign(static_cast<size_t>(m.ring_slots) * n, 0.0);
        rv.assign(static_cast<size_t>(m.ring_slots) * n, 0);
        arr.assign(n, 0);
        last.assign(n, 0);
        pos.assign(std::max<size_t>(1, m.in_train_beg.size()), 0);
        s = sanafe_hip_state{0, v.data(), u.data(), rf.data(), stt.data(), ring.data(), rv.data(), arr.data(), last.data(), pos.data()};
    };
    std::vector<double> av, au, aring, bv, bu, bring;
    std::vector<int32_t> arf, brf;
    std::vector<uint8_t> ast, arv, aarr, bst, brv, barr;
    std::vector<uint32_t> alast, apos, blast, bpos;
    sanafe_hip_state sa{}, sb{};
    buffers(a, av, au, arf, ast, aring, arv, aarr, alast, apos, sa);
    buffers(b, bv, bu, brf, bst, bring, brv, barr, blast, bpos, sb);
    DEV(sanafe_hip_export_state(from->dev, &sa));
    DEV(sanafe_hip_export_state(to->dev, &sb)); // the new neurons' initial values stay
    const int64_t t = sa.timesteps;
    const uint32_t ra = a.ring_slots, rb = b.ring_slots;
    for (int64_t g = 0; g < from->n_neurons; g++)
    {
        const uint32_t s0 = a.slot_of_gid[g], s1 = b.slot_of_gid[g];
        bv[s1] = av[s0];
        bu[s1] = au[s0];
        brf[s1] = arf[s0];
        bst[s1] = ast[s0];
        barr[s1] = aarr[s0];
        if (alast[s0] != 0u)
        {
            // a pending last-event entry is 1 + the event's position among the core's inbound synapses: only meaningful while
            // that list is what it was.  A network that adds inbound synapses to such a core shifts the positions.
            const uint32_t ca = a.core_of_slot[s0], cb = b.core_of_slot[s1];
            auto core_synapses = [](const MappedChip &m, uint32_t core) {
                const uint32_t k = core - m.first_core;
                const uint64_t end = (k + 1 < m.core_syn_base.size()) ? m.core_syn_base[k + 1] : m.syn_meta.size();
                return end - m.core_syn_base[k];
            };
            if (core_synapses(a, ca) != core_synapses(b, cb))
                return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: load(net, overwrite=False) adds inbound synapses to a core whose time-step "
                                                        "buffer (before the dendrite unit) still holds an event: its position cannot be carried");
        }
        blast[s1] = alast[s0];
        // what the time-step buffer / delay line holds for the steps to come: step t + 1 + k sits in row (t + 1 + k) % R
        for (uint32_t k = 0; k < ra; k++)
        {
            const size_t from_at = static_cast<size_t>((t + 1 + k) % ra) * a.n_slots + s0;
            if (!arv[from_at]) continue;
            if (k >= rb) return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: pending synaptic input does not fit the new chip's delay ring");
            const size_t to_at = static_cast<size_t>((t + 1 + k) % rb) * b.n_slots + s1;
            bring[to_at] = aring[from_at];
            brv[to_at] = 1;
        }
        if (a.slot_model[s0] == SANAFE_SOMA_INPUT && b.slot_model[s1] == SANAFE_SOMA_INPUT) bpos[b.slot_aux[s1]] = apos[a.slot_aux[s0]];
    }
    sb.timesteps = t;
    DEV(sanafe_hip_import_state(to->dev, &sb));
    to->ext.carry_from(from->ext); // stochastic value streams: the generators and file positions of the programmed units
    to->total_timesteps = from->total_timesteps;
    to->total_messages_sent = from->total_messages_sent;
    to->total_energy = from->total_energy;
    to->total_sim_time = from->total_sim_time;
    return 0;
}

extern "C" int sanafe_chip_attach_whole(sanafe_chip *chip, const sanafe_desc *desc)
{
    if (!chip || !desc) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    if (desc->n_neurons != chip->n_neurons) return fail(SANAFE_HIP_ERR_INVALID, "the description does not hold the chip's neurons");
    sanafe_chip *twin = nullptr;
    if (int rc = sanafe_chip_create(desc, -1, 1, 0, &twin)) return rc; // mapped only: no device, one rank
    if (twin->mc.n_global_slots != chip->mc.n_global_slots)
    {
        delete twin;
        return fail(SANAFE_HIP_ERR_INVALID, "the description maps to a different slot layout than this chip");
    }
    delete chip->whole;
    chip->whole = twin;
    return 0;
}

extern "C" void sanafe_chip_destroy(sanafe_chip *chip)
{
    if (!chip) return;
    sanafe_hip_chip_destroy(chip->dev);
    delete chip;
}

extern "C" int sanafe_chip_get_info(sanafe_chip *chip, sanafe_chip_info *o)
{
    if (!chip || !o) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    const MappedChip &mc = chip->mc;
    o->n_cores = mc.n_cores;
    o->n_local_cores = mc.last_core - mc.first_core;
    o->n_slots = mc.n_slots;
    o->n_global_slots = mc.n_global_slots;
    o->ring_slots = mc.ring_slots;
    o->n_slices = static_cast<uint32_t>(mc.slice_core.size());
    o->n_neurons = chip->n_neurons;
    o->n_axons = mc.ax_pre.size();
    o->n_synapses = mc.syn_meta.size();
    o->mapped_tiles = mc.mapped_tiles;
    o->mapped_cores = mc.mapped_cores;
    o->n_soma_classes = static_cast<uint32_t>(mc.soma_classes.size());
    o->n_cost_classes = static_cast<uint32_t>(mc.cost_classes.size());
    o->sync_delay = mc.sync_delay;
    o->image_bytes = mc.ax_pre.size() * 20ull + mc.syn_meta.size() * 12ull + static_cast<uint64_t>(mc.n_slots) * 80ull;
    return 0;
}

extern "C" sanafe_hip_chip *sanafe_chip_device(sanafe_chip *chip) { return chip ? chip->dev : nullptr; }

extern "C" int sanafe_chip_get_image(sanafe_chip *chip, sanafe_hip_image *out)
{
    if (!chip || !out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    *out = chip->mc.image();
    return 0;
}
extern "C" int sanafe_chip_get_slot_map(sanafe_chip *chip, uint32_t *slot_of_neuron)
{
    if (!chip || !slot_of_neuron) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    std::copy(chip->mc.slot_of_gid.begin(), chip->mc.slot_of_gid.end(), slot_of_neuron);
    return 0;
}

static void add_totals(sanafe_hip_totals &a, const sanafe_hip_totals &b)
{
    a.timesteps += 1;
    a.spikes += b.spikes;
    a.packets_sent += b.packets_sent;
    a.neurons_updated += b.neurons_updated;
    a.neurons_fired += b.neurons_fired;
    a.total_hops += b.total_hops;
    a.total_energy += b.total_energy;
    a.synapse_energy += b.synapse_energy;
    a.dendrite_energy += b.dendrite_energy;
    a.soma_energy += b.soma_energy;
    a.network_energy += b.network_energy;
    a.sim_time += b.sim_time;
}

// One rank of a tile-sharded chip: the enqueue loop of a chunk of timesteps -- neurons, gather of the spike windows (in
// line, or with RCCL on its own stream beside the delivery of the slices fed by local neurons), then the remaining
// slices.  The simple timing model needs the largest per-core delay over ALL ranks of every step, so each rank logs its
// own per-step maximum on the device and the logs are max-reduced once per chunk.
struct ShardedRun
{
    sanafe_chip *chip;
    sanafe_amd::Exchange &xc;
    int64_t cap{1}, next{0};
    double *d_log{nullptr};
    void *global_bits{nullptr};
    std::vector<uint32_t> h_local, h_global;
    std::vector<double> maxima; // [m] of the last chunk: largest per-core delay of every step over all ranks
    int64_t t_next{1};          // number of the next timestep (plugin somas are told)

    explicit ShardedRun(sanafe_chip *c) : chip(c), xc(c->xc) {}
    int begin(int64_t timesteps)
    {
        if (xc.kind == sanafe_amd::Exchange::None)
            return fail(SANAFE_HIP_ERR_INVALID, "this chip holds rank " + std::to_string(chip->rank) + " of " + std::to_string(chip->n_ranks) +
                            " of a tile-sharded chip: set the spike exchange up first (sanafe_chip_comm_init_rccl / _callback)");
        // the device keeps one entry per step in a ring of `cap` entries; the ring never shrinks, so `cap` is what the
        // device reports, not what this call asked for (a shorter sim() after a longer one)
        cap = std::max<int64_t>(1, std::min<int64_t>(timesteps, 1 << 16));
        DEV(sanafe_hip_delay_log(chip->dev, cap, &d_log, &cap, &next));
        void *local_bits = nullptr;
        uint64_t local_bytes = 0, global_bytes = 0;
        DEV(sanafe_hip_spike_buffers(chip->dev, &local_bits, &local_bytes, &global_bits, &global_bytes));
        if (xc.kind == sanafe_amd::Exchange::Callback)
        {
            h_local.resize(local_bytes / 4);
            h_global.resize(global_bytes / 4);
        }
        maxima.assign(static_cast<size_t>(cap), 0.0);
        t_next = chip->total_timesteps + 1;
        return 0;
    }
    // m <= cap steps; record_bits as sanafe_hip_step's (0: none).  Synchronises; fills maxima[0, m) when simple_timing.
    int chunk(int64_t m, int simple_timing, int record_bits)
    {
        if (int rc = chip->queue_ext(m)) return rc;
        DEV(sanafe_hip_record_begin(chip->dev, m, record_bits));
        DEV(sanafe_hip_delay_log(chip->dev, cap, &d_log, &cap, &next)); // flushed: `next` is where this chunk starts
        for (int64_t s = 0; s < m; s++, t_next++)
        {
            DEV(sanafe_hip_step_neurons(chip->dev));
            if (!chip->h_slots.empty())
            {
                // plugin somas of THIS rank's tiles (a core's units live where the core does), evaluated by the host between the
                // neuron launch and the exchange: their spikes are in the rank's bitmap before it is gathered
                const uint32_t n = static_cast<uint32_t>(chip->h_slots.size());
                try
                {
                    DEV(sanafe_hip_read_host_inputs(chip->dev, n, chip->h_slots.data(), chip->h_cur.data(), chip->h_has.data()));
                    chip->run_plugins(t_next);
                    DEV(sanafe_hip_write_host_status(chip->dev, n, chip->h_slots.data(), chip->h_status.data(), chip->h_cores.data(),
                            chip->h_energy.data(), chip->h_latency.data()));
                }
                catch (const std::exception &e)
                {
                    return fail(SANAFE_HIP_ERR_INVALID, e.what());
                }
            }
            if (xc.kind == sanafe_amd::Exchange::Rccl_ && xc.overlap)
            {
                if (xc.gather_spikes_rccl(global_bits)) return fail(SANAFE_HIP_ERR_HIP, xc.error);
                DEV(sanafe_hip_step_deliver_local(chip->dev));
                if (xc.wait_gathered()) return fail(SANAFE_HIP_ERR_HIP, xc.error);
                DEV(sanafe_hip_step_deliver_remote(chip->dev, simple_timing));
                continue;
            }
            if (xc.kind == sanafe_amd::Exchange::Rccl_)
            {
                if (xc.gather_spikes_rccl(global_bits)) return fail(SANAFE_HIP_ERR_HIP, xc.error);
            }
            else
            {
                DEV(sanafe_hip_export_spikes(chip->dev, h_local.data()));
                if (xc.gather_spikes_host(h_local.data(), h_global.data())) return fail(SANAFE_HIP_ERR_INVALID, xc.error);
                DEV(sanafe_hip_import_spikes(chip->dev, h_global.data()));
            }
            DEV(sanafe_hip_step_deliver(chip->dev, simple_timing, 0));
        }
        DEV(sanafe_hip_synchronize(chip->dev));
        if (!simple_timing) return 0;
        // this chunk's per-step maxima sit at [next, next + m) modulo cap: at most two pieces
        for (int64_t at = 0; at < m;)
        {
            const int64_t first = (next + at) % cap, len = std::min(m - at, cap - first);
            if (xc.kind == sanafe_amd::Exchange::Callback)
            {
                DEV(sanafe_hip_read_delay_log(chip->dev, first, len, maxima.data() + at));
            }
            if (xc.max_over_ranks(d_log + first, maxima.data() + at, static_cast<size_t>(len))) return fail(SANAFE_HIP_ERR_HIP, xc.error);
            at += len;
        }
        return 0;
    }
    int end() { return sanafe_hip_record_begin(chip->dev, 0, 0) != 0 ? fail(SANAFE_HIP_ERR_INVALID, sanafe_hip_last_error()) : 0; }
};

// sim() of one rank of a tile-sharded chip under the simple timing model: whole chunks stay on the device.  Counters
// and energies of the ranks are added in rank order.
static int sim_sharded(sanafe_chip *chip, int64_t timesteps, sanafe_hip_totals &run, int record, bool want_state)
{
    ShardedRun sr(chip);
    if (int rc = sr.begin(timesteps)) return rc;
    sanafe_amd::Exchange &xc = chip->xc;
    MappedChip &mc = chip->mc;
    DEV(sanafe_hip_reset_totals(chip->dev));
    const int64_t cap = sr.cap;
    std::vector<double> &maxima = sr.maxima;
    double sim_time = 0.0;
    const size_t n_ext = mc.ext.size();
    // Recorded runs (spike / perf traces): every rank records its own window on the device -- per-step totals and spike
    // bitmap rows -- and the ranks' records of a chunk of steps are gathered once per chunk: counters and energies added in
    // rank order, sim_time from the per-step maximum over the ranks, bitmap rows laid side by side in global slot order.
    uint32_t widest = 0;
    for (int k = 0; k < xc.n_ranks; k++) widest = std::max(widest, xc.slot_begin[k + 1] - xc.slot_begin[k]);
    const size_t row_words = widest / 32, local_words = mc.n_slots / 32, global_words = mc.n_global_slots / 32;
    const size_t rec_bytes = sizeof(sanafe_hip_totals) + row_words * sizeof(uint32_t); // per step and rank
    int64_t rec_cap = record ? std::max<int64_t>(1, (int64_t{64} << 20) / static_cast<int64_t>(rec_bytes * xc.n_ranks)) : cap;
    std::vector<unsigned char> rec_send, rec_recv;
    std::vector<sanafe_hip_totals> rec_local;
    std::vector<uint32_t> rows_local;
    for (int64_t done = 0, m = 0; done < timesteps; done += m)
    {
        m = std::min(std::min(cap, rec_cap), timesteps - done);
        if (n_ext != 0) m = std::min<int64_t>(m, std::max<int64_t>(1, (int64_t{16} << 20) / static_cast<int64_t>(n_ext)));
        if (int rc = sr.chunk(m, 1, (record ? 1 : 0) | ((want_state && chip->log_rank_columns[chip->rank] > 0) ? 8 : 0))) return rc; // (bit 3: only ranks that hold a logged neuron)
        for (int64_t s = 0; s < m; s++) sim_time += maxima[s] + mc.sync_delay;
        if (want_state)
        {
            // potential / neuron traces (src/chip.cpp:1766-1831): every rank sampled the neurons it holds on the device, one row
            // per step; the ranks' rows of this chunk are gathered once and laid out in the order the neurons were given
            const size_t mine = chip->log_rank_columns[chip->rank];
            size_t widest_row = 0;
            for (uint32_t w : chip->log_rank_columns) widest_row = std::max<size_t>(widest_row, w);
            std::vector<double> local(static_cast<size_t>(m) * std::max<size_t>(mine, 1), 0.0);
            if (mine > 0) DEV(sanafe_hip_read_step_state(chip->dev, 0, m, local.data()));
            std::vector<double> send(static_cast<size_t>(m) * widest_row, 0.0);
            for (int64_t s = 0; s < m; s++)
                std::copy(local.begin() + static_cast<size_t>(s) * mine, local.begin() + static_cast<size_t>(s + 1) * mine, send.begin() + static_cast<size_t>(s) * widest_row);
            std::vector<unsigned char> recv;
            if (xc.gather_bytes(send.data(), send.size() * sizeof(double), recv)) return fail(SANAFE_HIP_ERR_HIP, xc.error);
            const size_t row = chip->log_owner.size();
            const size_t base = chip->rec_state.size();
            chip->rec_state.resize(base + static_cast<size_t>(m) * row);
            const double *all = reinterpret_cast<const double *>(recv.data());
            for (int64_t s = 0; s < m; s++)
                for (size_t j = 0; j < row; j++)
                    chip->rec_state[base + static_cast<size_t>(s) * row + j] =
                            all[static_cast<size_t>(chip->log_owner[j]) * send.size() + static_cast<size_t>(s) * widest_row + chip->log_column[j]];
        }
        if (record)
        {
            rec_local.resize(m);
            rows_local.resize(static_cast<size_t>(m) * local_words);
            DEV(sanafe_hip_read_step_totals(chip->dev, 0, m, rec_local.data()));
            DEV(sanafe_hip_read_step_spike_rows(chip->dev, 0, m, rows_local.data()));
            rec_send.assign(static_cast<size_t>(m) * rec_bytes, 0);
            for (int64_t s = 0; s < m; s++)
            {
                unsigned char *at = rec_send.data() + static_cast<size_t>(s) * rec_bytes;
                std::memcpy(at, &rec_local[s], sizeof(sanafe_hip_totals));
                std::memcpy(at + sizeof(sanafe_hip_totals), rows_local.data() + static_cast<size_t>(s) * local_words, local_words * sizeof(uint32_t));
            }
            if (xc.gather_bytes(rec_send.data(), rec_send.size(), rec_recv)) return fail(SANAFE_HIP_ERR_HIP, xc.error);
            for (int64_t s = 0; s < m; s++)
            {
                sanafe_hip_totals ts{};
                std::vector<uint32_t> bits(global_words, 0u);
                for (int k = 0; k < xc.n_ranks; k++) // rank order: reproducible sums
                {
                    const unsigned char *at = rec_recv.data() + static_cast<size_t>(k) * rec_send.size() + static_cast<size_t>(s) * rec_bytes;
                    sanafe_hip_totals part;
                    std::memcpy(&part, at, sizeof(part));
                    add_totals(ts, part);
                    std::memcpy(bits.data() + xc.slot_begin[k] / 32, at + sizeof(sanafe_hip_totals),
                            static_cast<size_t>(xc.slot_begin[k + 1] - xc.slot_begin[k]) / 8);
                }
                ts.timesteps = chip->total_timesteps + s + 1;
                ts.sim_time = maxima[s] + mc.sync_delay;
                chip->rec_totals.push_back(ts);
                chip->rec_spike_bits.push_back(std::move(bits));
            }
        }
        chip->total_timesteps += m;
    }
    if (int rc = sr.end()) return rc;
    if (record && timesteps > 0)
    {
        chip->have_records = true;
        chip->rec_bits_global = true;
        chip->rec_count = timesteps;
    }
    sanafe_hip_totals mine{};
    DEV(sanafe_hip_read_totals(chip->dev, &mine));
    std::vector<sanafe_hip_totals> all;
    if (xc.gather_totals(mine, sanafe_hip_run_totals_device(chip->dev), all)) return fail(SANAFE_HIP_ERR_HIP, xc.error);
    run = sanafe_hip_totals{};
    for (const sanafe_hip_totals &t : all) // rank order
    {
        run.spikes += t.spikes;
        run.packets_sent += t.packets_sent;
        run.neurons_updated += t.neurons_updated;
        run.neurons_fired += t.neurons_fired;
        run.total_hops += t.total_hops;
        run.total_energy += t.total_energy;
        run.synapse_energy += t.synapse_energy;
        run.dendrite_energy += t.dendrite_energy;
        run.soma_energy += t.soma_energy;
        run.network_energy += t.network_energy;
    }
    run.sim_time = sim_time;
    chip->total_messages_sent += run.packets_sent;
    return 0;
}

extern "C" int sanafe_comm_unique_id(uint8_t id[SANAFE_COMM_ID_BYTES])
{
    if (!id) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    if (sanafe_amd::Exchange::unique_id(id) != 0)
        return fail(SANAFE_HIP_ERR_HIP, "ncclGetUniqueId failed: " + sanafe_amd::Exchange::library_error());
    return 0;
}
extern "C" int sanafe_chip_comm_init_rccl(sanafe_chip *chip, const uint8_t id[SANAFE_COMM_ID_BYTES])
{
    if (!chip || !id) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    if (!chip->dev) return fail(SANAFE_HIP_ERR_NO_DEVICE, "the chip has no device (mapped only)");
    if (chip->xc.init_rccl(id, chip->device, sanafe_hip_stream(chip->dev))) return fail(SANAFE_HIP_ERR_HIP, chip->xc.error);
    return 0;
}
extern "C" int sanafe_chip_comm_init_callback(sanafe_chip *chip, sanafe_allgather_fn fn, void *ctx)
{
    if (!chip) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    if (chip->xc.init_callback(fn, ctx)) return fail(SANAFE_HIP_ERR_INVALID, chip->xc.error);
    return 0;
}

extern "C" int sanafe_chip_sim(sanafe_chip *chip, int64_t timesteps, int timing_model, int record, sanafe_hip_totals *rd)
{
    if (!chip || timesteps < 0) return fail(SANAFE_HIP_ERR_INVALID, "bad arguments");
    if (!chip->dev) return fail(SANAFE_HIP_ERR_NO_DEVICE, "chip was mapped without a device (device < 0); there is no CPU execution path");
    if (timing_model == SANAFE_TIMING_CYCLE)
        return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: the cycle-accurate (Booksim2) timing model is out of scope");
    if (timing_model != SANAFE_TIMING_SIMPLE && timing_model != SANAFE_TIMING_DETAILED)
        return fail(SANAFE_HIP_ERR_INVALID, "unknown timing model");
    if (int rc = chip->commit_attributes()) return rc;
    MappedChip &mc = chip->mc;
    sanafe_hip_totals run{};
    chip->have_records = false;
    chip->rec_bits_global = false;
    chip->rec_totals.clear();
    chip->rec_messages.clear();
    chip->rec_spike_bits.clear();
    chip->rec_optional.clear();
    chip->rec_first_timestep = chip->total_timesteps + 1;
    chip->rec_count = 0;
    const bool host_units = !chip->mc.host_neurons.empty() || chip->hcores != nullptr;
    const bool want_messages = (record & SANAFE_RECORD_MESSAGES) != 0;
    if (chip->mc.msg_on_device && host_units && (timing_model != SANAFE_TIMING_SIMPLE || want_messages || ((record & SANAFE_RECORD_STEPS) && chip->mc.log.any)))
        return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: a chip with plugin soma units AND cores whose soma is part of the message pipeline "
                                                "is simulated under the simple timing model, without message traces or optional perf columns "
                                                "(the per-message fired counts are logged by batched runs only)");
    if (chip->mc.msg_on_device && (timing_model != SANAFE_TIMING_SIMPLE || want_messages) && (chip->n_ranks > 1 || chip->xc.kind != sanafe_amd::Exchange::None))
        return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: detailed timing and message traces of a chip with cores whose soma is part of "
                                                "the message pipeline need a single-rank chip");
    const bool want_state = (record & SANAFE_RECORD_STATE) != 0;
    if (want_state && chip->log_v_gids.empty() && chip->log_u_gids.empty())
        return fail(SANAFE_HIP_ERR_INVALID, "SANAFE_RECORD_STATE needs the neurons to log (sanafe_chip_set_state_log)");
    record = (record & (SANAFE_RECORD_STEPS | SANAFE_RECORD_MESSAGES | SANAFE_RECORD_STATE)) ? 1 : 0;
    const size_t state_row = chip->log_v_gids.size() + chip->log_u_gids.size();
    chip->rec_state.clear();
    // (an exchange set up on a single-rank chip is honoured as well: the same loop with a one-rank gather)
    if (chip->n_ranks > 1 || chip->xc.kind != sanafe_amd::Exchange::None)
    {
        if (chip->hcores)
            return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: cores that run on the host are not available on a tile-sharded chip");
        if (record && chip->mc.log.any && (host_units || want_state))
            return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: optional perf columns of a tile-sharded chip are not available with "
                                                    "plugin units or potential traces");
        if (host_units && (timing_model != SANAFE_TIMING_SIMPLE || want_messages || want_state))
            return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: plugin soma units on a tile-sharded chip need the simple timing model, "
                                                    "without a message trace or potential traces (their latencies and potentials live on "
                                                    "the rank that holds them)");
        if (want_state && (timing_model != SANAFE_TIMING_SIMPLE || want_messages))
            return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: potential / neuron traces of a tile-sharded chip need the simple timing "
                                                    "model without a message trace");
        if ((timing_model != SANAFE_TIMING_SIMPLE || want_messages || (record && chip->mc.log.any)) && chip->whole == nullptr)
            return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: detailed timing, message traces and optional perf columns on a tile-sharded "
                                                    "chip need the whole chip's tables on the host: call sanafe_chip_attach_whole with the "
                                                    "complete description first");
    }
    const bool sharded = chip->n_ranks > 1 || chip->xc.kind != sanafe_amd::Exchange::None;
    if (sharded && timing_model == SANAFE_TIMING_SIMPLE && !want_messages && !(record && chip->mc.log.any))
    {
        if (int rc = sim_sharded(chip, timesteps, run, record, want_state)) return rc;
    }
    else if (!sharded && timing_model == SANAFE_TIMING_SIMPLE && !host_units && !want_messages && !(record && chip->mc.log.any))
    {
        // Whole run stays on the device; nothing comes back per step unless recorded.  With external value
        // streams the run is cut into chunks whose stream rows fit a bounded upload (<= 64 MiB).
        DEV(sanafe_hip_reset_totals(chip->dev));
        const size_t n_ext = mc.ext.size();
        int64_t chunk_cap = n_ext == 0 ? std::max<int64_t>(timesteps, 1) : std::max<int64_t>(1, (int64_t{16} << 20) / static_cast<int64_t>(n_ext));
        // recorded runs keep a spike bitmap and a totals record per step on the device: bounded chunks (<= 64 MiB)
        if (record) chunk_cap = std::min<int64_t>(chunk_cap, std::max<int64_t>(1, (int64_t{64} << 20) / static_cast<int64_t>(mc.n_slots / 8 + 96 + (want_state ? state_row * 8 : 0))));
        if (const char *env = std::getenv("SANAFE_EXT_CHUNK_STEPS")) // tests: force several chunks
            if (n_ext != 0) chunk_cap = std::max<int64_t>(1, std::atol(env));
        for (int64_t done = 0, m = 0; done < timesteps; done += m)
        {
            m = std::min(chunk_cap, timesteps - done);
            if (int rc = chip->queue_ext(m)) return rc;
            DEV(sanafe_hip_step(chip->dev, m, 1, record | (want_state ? 8 : 0)));
            DEV(sanafe_hip_synchronize(chip->dev));
            if (want_state && m > 0)
            {
                chip->rec_state.resize(static_cast<size_t>(done + m) * state_row);
                DEV(sanafe_hip_read_step_state(chip->dev, 0, m, chip->rec_state.data() + static_cast<size_t>(done) * state_row));
            }
            if (record && m > 0)
            {
                chip->rec_totals.resize(done + m);
                DEV(sanafe_hip_read_step_totals(chip->dev, 0, m, chip->rec_totals.data() + done));
                chip->rec_spike_bits.resize(done + m);
                const size_t words = mc.n_slots / 32;
                std::vector<uint32_t> rows(static_cast<size_t>(m) * words); // the device log is contiguous: one copy per chunk
                DEV(sanafe_hip_read_step_spike_rows(chip->dev, 0, m, rows.data()));
                for (int64_t s = 0; s < m; s++)
                    chip->rec_spike_bits[done + s].assign(rows.begin() + static_cast<size_t>(s) * words, rows.begin() + static_cast<size_t>(s + 1) * words);
            }
        }
        DEV(sanafe_hip_read_totals(chip->dev, &run));
        if (record && timesteps > 0)
        {
            chip->have_records = true;
            chip->rec_count = timesteps;
        }
        chip->total_timesteps += timesteps;
        chip->total_messages_sent += run.packets_sent; // every message takes an id under any timing model, src/chip.cpp:811
    }
    else
    {
        // `detailed`: functional step on the GPU, NoC discrete-event schedule on the host
        // (serial by construction, src/schedule.cpp:234-281), one step at a time.
        // The same stepwise loop serves plugin (host-evaluated) soma units under either timing model.
        const bool detailed = (timing_model == SANAFE_TIMING_DETAILED);
        if (chip->hcores && (detailed || want_messages || (record && chip->mc.log.any)))
            return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: cores that run on the host (buffer inside the soma unit / before axon_out, "
                                                    "plugin synapse or dendrite units) are simulated under the simple timing model, without "
                                                    "message traces or optional perf columns: their message costs are only known at run time");
        // H: whose tables rebuild and schedule the messages -- the chip itself, or the whole-chip twin of a sharded one
        sanafe_chip *H = sharded ? chip->whole : chip;
        const MappedChip &hmc = H->mc;
        if ((detailed || want_messages) && hmc.out_ptr.empty())
            return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: detailed timing and message traces need a single-rank chip");
        std::vector<uint8_t> status(hmc.n_slots);
        struct Job
        {
            sanafe_hip_totals ts{};
            int64_t mid_base{0};
            std::vector<uint8_t> status;
            std::vector<uint32_t> bits;
            std::vector<uint16_t> msg_fired; // chips with message-pipeline cores on the device: fired updates per message into them
            std::vector<Msg> flat;
            std::string error;
        };
        std::deque<Job> jobs; // stable addresses
        std::deque<Job *> ready;
        std::mutex qmutex;
        std::condition_variable qcv;
        bool done_submitting = false;
        auto run_job = [&](Job &job) {
            try
            {
                if (want_messages) // the traced record of every message, in scheduling order
                {
                    thread_local sanafe_chip::SchedScratch<Msg> scratch;
                    H->build_messages(job.ts.timesteps, job.status, scratch.per_core, job.mid_base, job.msg_fired.empty() ? nullptr : job.msg_fired.data());
                    job.ts.sim_time = H->schedule_detailed(scratch.per_core, true, scratch);
                    for (auto &q : scratch.per_core) job.flat.insert(job.flat.end(), q.begin(), q.end());
                }
                else // only sim_time is wanted: the compact message, queues and NoC state reused by this thread
                {
                    thread_local sanafe_chip::SchedScratch<SchedMsg> scratch;
                    H->build_messages(job.ts.timesteps, job.status, scratch.per_core, job.mid_base, job.msg_fired.empty() ? nullptr : job.msg_fired.data());
                    job.ts.sim_time = H->schedule_detailed(scratch.per_core, false, scratch);
                }
            }
            catch (const std::exception &e)
            {
                job.error = e.what();
            }
            std::vector<uint8_t>().swap(job.status);
        };
        std::vector<std::thread> pool;
        // plugin units update slot latencies every step, so their runs schedule inline
        const int n_workers = (detailed && !host_units) ? chip->scheduler_threads : 0;
        for (int w = 0; w < n_workers; w++)
            pool.emplace_back([&] {
                for (;;)
                {
                    Job *job = nullptr;
                    {
                        std::unique_lock<std::mutex> lock(qmutex);
                        qcv.wait(lock, [&] { return !ready.empty() || done_submitting; });
                        if (ready.empty()) return;
                        job = ready.front();
                        ready.pop_front();
                    }
                    run_job(*job);
                }
            });
        struct PoolGuard // never leave joinable threads behind on an error return
        {
            std::vector<std::thread> &pool;
            std::mutex &m;
            std::condition_variable &cv;
            bool &done;
            ~PoolGuard()
            {
                {
                    std::lock_guard<std::mutex> lock(m);
                    done = true;
                }
                cv.notify_all();
                for (std::thread &th : pool)
                    if (th.joinable()) th.join();
            }
        } guard{pool, qmutex, qcv, done_submitting};
        // one finished timestep: totals + the status of every slot -> records, or a job for the scheduler threads
        // (final_bits / msg_fired: chips with message-pipeline cores on the device -- the spike record of the END of the step, which
        // the message pipeline's soma calls latch too, and the fired updates per message into such a core)
        const size_t n_msg_ax = chip->mc.msg_on_device ? chip->mc.msg_ax_pre.size() : 0;
        auto process_step = [&](sanafe_hip_totals ts, const uint8_t *st_bytes, const uint32_t *final_bits = nullptr, const uint16_t *msg_fired = nullptr) -> int {
            chip->total_timesteps += 1;
            ts.timesteps = chip->total_timesteps;
            const int64_t mid_base = chip->total_messages_sent;
            chip->total_messages_sent += ts.packets_sent;
            std::vector<uint32_t> bits;
            if (record)
            {
                bits.assign(hmc.n_slots / 32, 0);
                if (final_bits != nullptr) bits.assign(final_bits, final_bits + hmc.n_slots / 32);
                else
                    for (uint32_t k = 0; k < hmc.n_slots; k++)
                        if (st_bytes[k] == 3) bits[k >> 5] |= 1u << (k & 31u);
                if (hmc.log.any) chip->rec_optional.push_back(H->optional_columns(st_bytes, msg_fired));
            }
            if (!detailed)
            {
                add_totals(run, ts);
                if (record)
                {
                    chip->rec_totals.push_back(ts);
                    chip->rec_messages.emplace_back();
                    if (want_messages)
                    {
                        // schedule_messages_timestep_simple, src/schedule.cpp:61-102: the messages stay in their source
                        // cores' FIFOs; no blocking is modelled and the network delay is the minimum hop delay
                        std::vector<uint8_t> st_copy(st_bytes, st_bytes + hmc.n_slots);
                        std::vector<std::vector<Msg>> per_core;
                        H->build_messages(ts.timesteps, st_copy, per_core, mid_base, msg_fired);
                        std::vector<Msg> &flat = chip->rec_messages.back();
                        for (auto &q : per_core)
                            for (Msg &m : q)
                            {
                                m.blocking_delay = 0.0;
                                m.network_delay = m.min_hop_delay;
                                flat.push_back(m);
                            }
                    }
                    chip->rec_spike_bits.push_back(std::move(bits));
                }
                return 0;
            }
            // The NoC model of one timestep depends only on that timestep's messages (NocInfo is rebuilt per
            // step, src/schedule.cpp:208-222), so timesteps are handed to scheduler threads like the
            // reference's `-S n` does (src/schedule.cpp:182-206, 622-661) while the GPU simulates ahead.
            jobs.emplace_back();
            Job &job = jobs.back();
            job.ts = ts;
            job.mid_base = mid_base;
            job.status.assign(st_bytes, st_bytes + hmc.n_slots);
            if (msg_fired != nullptr) job.msg_fired.assign(msg_fired, msg_fired + n_msg_ax);
            job.bits = std::move(bits);
            if (pool.empty())
            {
                run_job(job);
                if (!job.error.empty()) return fail(SANAFE_HIP_ERR_INVALID, job.error);
            }
            else
            {
                // the GPU runs far ahead of the schedulers: keep at most ~1k finished steps (with their status arrays)
                // waiting, like the reference's bound on buffered timesteps (src/schedule.cpp:182-206)
                for (;;)
                {
                    {
                        std::lock_guard<std::mutex> lock(qmutex);
                        if (ready.size() < 1024)
                        {
                            ready.push_back(&job);
                            break;
                        }
                    }
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                }
                qcv.notify_one();
            }
            return 0;
        };
        if (sharded)
        {
            // Tile-sharded chip: K split steps per chunk with the spike exchange in between, every rank recording the totals
            // and the NeuronStatus of its own slots; the ranks' records are gathered once per chunk and every rank replays the
            // whole chip's messages on the twin's tables (the same deterministic schedule on every rank).
            ShardedRun sr(chip);
            if (int rc = sr.begin(timesteps)) return rc;
            sanafe_amd::Exchange &xc = chip->xc;
            uint32_t widest = 0;
            for (int k = 0; k < xc.n_ranks; k++) widest = std::max(widest, xc.slot_begin[k + 1] - xc.slot_begin[k]);
            const size_t rec_bytes = sizeof(sanafe_hip_totals) + widest; // per step and rank: totals + one status byte per slot
            const int64_t k_cap = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(64, sr.cap), (int64_t{64} << 20) / static_cast<int64_t>(rec_bytes * xc.n_ranks)));
            std::vector<sanafe_hip_totals> tsv;
            std::vector<uint8_t> stv, st_global(hmc.n_slots);
            std::vector<unsigned char> send, recv;
            for (int64_t s = 0; s < timesteps;)
            {
                const int64_t k_steps = std::min(k_cap, timesteps - s);
                if (int rc = sr.chunk(k_steps, detailed ? 0 : 1, 3)) return rc;
                tsv.resize(k_steps);
                stv.resize(static_cast<size_t>(k_steps) * mc.n_slots);
                DEV(sanafe_hip_read_step_totals(chip->dev, 0, k_steps, tsv.data()));
                DEV(sanafe_hip_read_step_status(chip->dev, 0, k_steps, stv.data()));
                send.assign(static_cast<size_t>(k_steps) * rec_bytes, 0);
                for (int64_t k = 0; k < k_steps; k++)
                {
                    unsigned char *at = send.data() + static_cast<size_t>(k) * rec_bytes;
                    std::memcpy(at, &tsv[k], sizeof(sanafe_hip_totals));
                    std::memcpy(at + sizeof(sanafe_hip_totals), stv.data() + static_cast<size_t>(k) * mc.n_slots, mc.n_slots);
                }
                if (xc.gather_bytes(send.data(), send.size(), recv)) return fail(SANAFE_HIP_ERR_HIP, xc.error);
                for (int64_t k = 0; k < k_steps; k++)
                {
                    sanafe_hip_totals ts{};
                    for (int r = 0; r < xc.n_ranks; r++) // rank order: reproducible sums
                    {
                        const unsigned char *at = recv.data() + static_cast<size_t>(r) * send.size() + static_cast<size_t>(k) * rec_bytes;
                        sanafe_hip_totals part;
                        std::memcpy(&part, at, sizeof(part));
                        add_totals(ts, part);
                        std::memcpy(st_global.data() + xc.slot_begin[r], at + sizeof(sanafe_hip_totals), xc.slot_begin[r + 1] - xc.slot_begin[r]);
                    }
                    ts.sim_time = detailed ? 0.0 : sr.maxima[k] + mc.sync_delay;
                    if (int rc = process_step(ts, st_global.data())) return rc;
                }
                s += k_steps;
            }
            if (int rc = sr.end()) return rc;
            if (record) chip->rec_bits_global = true;
        }
        else if (!host_units)
        {
            // `detailed` without plugin units: the device runs K steps back to back and logs every step's totals
            // and slot statuses; the host fetches them in one go (no round trip per step) and rebuilds the messages.
            const int64_t k_cap = std::max<int64_t>(1, std::min<int64_t>(64, (int64_t{64} << 20) / std::max<uint32_t>(1, mc.n_slots)));
            std::vector<sanafe_hip_totals> tsv;
            std::vector<uint8_t> stv;
            std::vector<uint32_t> rowv;
            std::vector<uint16_t> firedv;
            for (int64_t s = 0; s < timesteps;)
            {
                const int64_t k_steps = std::min(k_cap, timesteps - s);
                if (int rc = chip->queue_ext(k_steps)) return rc;
                DEV(sanafe_hip_step(chip->dev, k_steps, detailed ? 0 : 1, 3 | (want_state ? 8 : 0)));
                DEV(sanafe_hip_synchronize(chip->dev));
                if (want_state)
                {
                    chip->rec_state.resize(static_cast<size_t>(s + k_steps) * state_row);
                    DEV(sanafe_hip_read_step_state(chip->dev, 0, k_steps, chip->rec_state.data() + static_cast<size_t>(s) * state_row));
                }
                tsv.resize(k_steps);
                stv.resize(static_cast<size_t>(k_steps) * mc.n_slots);
                DEV(sanafe_hip_read_step_totals(chip->dev, 0, k_steps, tsv.data()));
                DEV(sanafe_hip_read_step_status(chip->dev, 0, k_steps, stv.data()));
                if (n_msg_ax != 0)
                {
                    rowv.resize(static_cast<size_t>(k_steps) * (mc.n_slots / 32));
                    firedv.resize(static_cast<size_t>(k_steps) * n_msg_ax);
                    DEV(sanafe_hip_read_step_spike_rows(chip->dev, 0, k_steps, rowv.data()));
                    DEV(sanafe_hip_read_step_msg_fired(chip->dev, 0, k_steps, firedv.data()));
                }
                for (int64_t k = 0; k < k_steps; k++)
                    if (int rc = process_step(tsv[k], stv.data() + static_cast<size_t>(k) * mc.n_slots,
                                n_msg_ax ? rowv.data() + static_cast<size_t>(k) * (mc.n_slots / 32) : nullptr,
                                n_msg_ax ? firedv.data() + static_cast<size_t>(k) * n_msg_ax : nullptr))
                        return rc;
                s += k_steps;
            }
        }
        else
        for (int64_t s = 0; s < timesteps; s++)
        {
            DEV(sanafe_hip_reset_totals(chip->dev));
            if (int rc = chip->queue_ext(1)) return rc;
            {
                const uint32_t n = static_cast<uint32_t>(chip->h_slots.size());
                const long t_now = static_cast<long>(chip->total_timesteps + 1);
                sanafe_amd::HostCores *hcs = chip->hcores.get();
                DEV(sanafe_hip_step_neurons(chip->dev));
                try
                {
                    if (n > 0)
                    {
                        DEV(sanafe_hip_read_host_inputs(chip->dev, n, chip->h_slots.data(), chip->h_cur.data(), chip->h_has.data()));
                        chip->run_plugins(chip->total_timesteps + 1);
                        DEV(sanafe_hip_write_host_status(chip->dev, n, chip->h_slots.data(), chip->h_status.data(), chip->h_cores.data(),
                                chip->h_energy.data(), chip->h_latency.data()));
                    }
                    if (hcs)
                    {
                        // neuron pipelines of the cores that run on the host; their statuses complete the step's spike bitmap
                        hcs->begin_step();
                        hcs->process_neurons(t_now);
                        DEV(sanafe_hip_write_host_core_status(chip->dev, static_cast<uint32_t>(hcs->slots().size()), hcs->slots().data(),
                                hcs->status().data(), hcs->slot_cores().data()));
                        chip->hc_bits.resize(mc.n_slots / 32);
                        DEV(sanafe_hip_export_spikes(chip->dev, chip->hc_bits.data()));
                    }
                    DEV(sanafe_hip_step_deliver(chip->dev, detailed ? 0 : 1, 0));
                    if (hcs)
                    {
                        // their message pipelines, replayed while the device delivers to its own cores
                        hcs->process_messages(t_now, chip->hc_bits.data());
                        hcs->forced_updates(t_now);
                        hcs->end_step();
                        chip->hc_costs.resize(hcs->partials().size());
                        for (size_t k = 0; k < hcs->partials().size(); k++)
                        {
                            const sanafe_amd::HostCores::Partial &p = hcs->partials()[k];
                            sanafe_hip_host_core_costs &o = chip->hc_costs[k];
                            o.core = hcs->core_ids()[k];
                            o.pad = 0;
                            o.synapse_energy = p.e_syn;
                            o.dendrite_energy = p.e_dend;
                            o.soma_energy = p.e_soma;
                            o.neuron_latency = p.neuron_latency;
                            o.processing_delay = p.processing;
                            o.neurons_updated = p.updated;
                            o.neurons_fired = p.fired;
                        }
                        DEV(sanafe_hip_write_host_core_costs(chip->dev, static_cast<uint32_t>(chip->hc_costs.size()), chip->hc_costs.data()));
                    }
                }
                catch (const std::exception &e)
                {
                    return fail(SANAFE_HIP_ERR_INVALID, e.what());
                }
            }
            sanafe_hip_totals ts{};
            DEV(sanafe_hip_read_totals(chip->dev, &ts));
            DEV(sanafe_hip_read_status(chip->dev, status.data()));
            if (chip->hcores) // the status a neuron of a host core holds at the END of the step (latched by the message pipeline too)
                for (size_t k = 0; k < chip->hcores->slots().size(); k++) status[chip->hcores->slots()[k]] = chip->hcores->final_status()[k];
            if (want_state) // plugin somas keep their own potentials: sampled on the host
            {
                std::vector<double> v(static_cast<size_t>(chip->n_neurons)), u(static_cast<size_t>(chip->n_neurons));
                if (int rc = sanafe_chip_get_potentials(chip, v.data())) return rc;
                if (!chip->log_u_gids.empty())
                    if (int rc = sanafe_chip_get_input_current(chip, u.data())) return rc;
                for (int64_t g : chip->log_v_gids) chip->rec_state.push_back(v[g]);
                for (int64_t g : chip->log_u_gids) chip->rec_state.push_back(u[g]);
            }
            if (int rc = process_step(ts, status.data())) return rc;
        }
        if (detailed)
        {
            {
                std::lock_guard<std::mutex> lock(qmutex);
                done_submitting = true;
            }
            qcv.notify_all();
            for (std::thread &th : pool) th.join();
            for (Job &job : jobs) // retire in timestep order: deterministic accumulation (flush_timestep_data)
            {
                if (!job.error.empty()) return fail(SANAFE_HIP_ERR_INVALID, job.error);
                add_totals(run, job.ts);
                if (record)
                {
                    chip->rec_totals.push_back(job.ts);
                    chip->rec_messages.push_back(std::move(job.flat));
                    chip->rec_spike_bits.push_back(std::move(job.bits));
                }
            }
        }
        if (record)
        {
            chip->have_records = true;
            chip->rec_count = timesteps;
        }
    }
    chip->total_energy += run.total_energy;
    chip->total_sim_time += run.sim_time;
    run.timesteps = timesteps;
    if (rd) *rd = run;
    return 0;
}

extern "C" int sanafe_chip_set_scheduler_threads(sanafe_chip *chip, int n_threads)
{
    if (!chip) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    if (n_threads < 0 || n_threads > 256) return fail(SANAFE_HIP_ERR_INVALID, "scheduler_threads must be in 0..256");
    chip->scheduler_threads = n_threads;
    return 0;
}

// Self-check hook: the first n values of the host's rand() restatement (tests compare with libc).
extern "C" void sanafe_test_glibc_rand(uint32_t seed, int64_t n, uint32_t *out)
{
    GlibcRand g(seed);
    for (int64_t i = 0; i < n; i++) out[i] = g.next();
}

// Test / profiling hook: rebuilds the messages of one timestep from `status` (one NeuronStatus byte per local slot)
// and runs the detailed NoC schedule `reps` times on the calling thread -- no device involved, so the host-side cost
// per message can be measured on any machine.  Returns sim_time of the step and the number of messages.
static int test_schedule(sanafe_chip *chip, const uint8_t *status, const uint16_t *msg_fired, int reps, double *sim_time, int64_t *n_messages,
        double *build_seconds, double *schedule_seconds);
extern "C" int sanafe_test_schedule(sanafe_chip *chip, const uint8_t *status, int reps, double *sim_time, int64_t *n_messages,
        double *build_seconds, double *schedule_seconds)
{
    return test_schedule(chip, status, nullptr, reps, sim_time, n_messages, build_seconds, schedule_seconds);
}
// ... for chips with message-pipeline cores on the device: `msg_fired` = per message into such a core (sanafe_hip_image::msg_ax_*
// order) how many of its synaptic events made the soma fire -- what sanafe_hip_read_step_msg_fired returns for a recorded step.
extern "C" int sanafe_test_schedule_msg(sanafe_chip *chip, const uint8_t *status, const uint16_t *msg_fired, double *sim_time, int64_t *n_messages)
{
    return test_schedule(chip, status, msg_fired, 1, sim_time, n_messages, nullptr, nullptr);
}
// Test hook: the optional perf columns of one timestep (sim_trace_get_optional_traces, src/chip.cpp:1541-1579) from the statuses
// the neuron loop left and, on chips with message-pipeline cores on the device, the fired counts per message -- what a
// recorded sim() computes from the device's step logs.  out: sanafe_chip_perf_columns() values.
extern "C" int sanafe_test_optional_columns(sanafe_chip *chip, const uint8_t *status, const uint16_t *msg_fired, double *out)
{
    if (!chip || !status || !out || chip->mc.out_ptr.empty() || !chip->mc.log.any)
        return fail(SANAFE_HIP_ERR_INVALID, "needs a single-rank chip with log flags, a status array and an output array");
    try
    {
        const std::vector<double> cols = chip->optional_columns(status, msg_fired);
        std::copy(cols.begin(), cols.end(), out);
    }
    catch (const std::exception &e)
    {
        return fail(SANAFE_HIP_ERR_INVALID, e.what());
    }
    return 0;
}
static int test_schedule(sanafe_chip *chip, const uint8_t *status, const uint16_t *msg_fired, int reps, double *sim_time, int64_t *n_messages,
        double *build_seconds, double *schedule_seconds)
{
    if (!chip || !status || chip->mc.out_ptr.empty()) return fail(SANAFE_HIP_ERR_INVALID, "needs a single-rank chip and a status array");
    try
    {
        const std::vector<uint8_t> st(status, status + chip->mc.n_slots);
        sanafe_chip::SchedScratch<SchedMsg> scratch;
        double tb = 0.0, tsch = 0.0, last = 0.0;
        int64_t n = 0;
        for (int r = 0; r < reps; r++)
        {
            const auto t0 = std::chrono::steady_clock::now();
            chip->build_messages(1, st, scratch.per_core, 0, msg_fired);
            const auto t1 = std::chrono::steady_clock::now();
            last = chip->schedule_detailed(scratch.per_core, false, scratch);
            const auto t2 = std::chrono::steady_clock::now();
            tb += std::chrono::duration<double>(t1 - t0).count();
            tsch += std::chrono::duration<double>(t2 - t1).count();
        }
        for (const auto &q : scratch.per_core) n += static_cast<int64_t>(q.size());
        if (sim_time) *sim_time = last;
        if (n_messages) *n_messages = n;
        if (build_seconds) *build_seconds = tb;
        if (schedule_seconds) *schedule_seconds = tsch;
    }
    catch (const std::exception &e)
    {
        return fail(SANAFE_HIP_ERR_INVALID, e.what());
    }
    return 0;
}

extern "C" int sanafe_chip_generate_ext(sanafe_chip *chip, int64_t steps, int32_t *out)
{
    if (!chip || steps < 0 || (steps > 0 && !out && !chip->mc.ext.empty())) return fail(SANAFE_HIP_ERR_INVALID, "bad arguments");
    if (chip->structure_dirty)
        if (int rc = chip->rebuild_device()) return rc;
    const size_t n = chip->mc.ext.size();
    chip->ext_hook_steps += steps; // (the units' update count, for columns that arrive later: see set_input_attribute)
    try
    {
        for (int64_t s = 0; s < steps && n > 0; s++) chip->ext.fill_row(out + static_cast<size_t>(s) * n);
    }
    catch (const std::exception &e)
    {
        return fail(SANAFE_HIP_ERR_INVALID, e.what());
    }
    return 0;
}

extern "C" int sanafe_chip_reset(sanafe_chip *chip)
{
    if (!chip) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    DEV(sanafe_hip_reset(chip->dev));
    for (auto &u : chip->plugin_units) u->reset();
    if (chip->hcores) chip->hcores->reset();
    return 0;
}

extern "C" double sanafe_chip_get_power(sanafe_chip *chip) // src/chip.cpp:607-621
{
    if (!chip || !(chip->total_sim_time > 0.0)) return 0.0;
    return chip->total_energy / chip->total_sim_time;
}

template <typename T, typename F> static int gather_by_gid(sanafe_chip *chip, T *out, F &&read)
{
    const MappedChip &mc = chip->mc;
    std::vector<T> slots(mc.n_slots);
    int rc = read(slots.data());
    if (rc != 0) return fail(rc, sanafe_hip_last_error());
    for (int64_t g = 0; g < chip->n_neurons; g++)
    {
        const uint32_t s = mc.slot_of_gid[g];
        out[g] = (s >= mc.slot_offset && s < mc.slot_offset + mc.n_slots) ? slots[s - mc.slot_offset] : T(0);
    }
    return 0;
}

extern "C" int sanafe_chip_get_status(sanafe_chip *chip, uint8_t *out)
{
    if (!chip || !out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    const int rc = gather_by_gid(chip, out, [&](uint8_t *p) { return sanafe_hip_read_status(chip->dev, p); });
    if (rc != 0) return rc;
    if (chip->hcores) // cores that run on the host: MappedNeuron::status as the END of the step left it
    {
        size_t i = 0;
        for (const MappedChip::HostCore &hc : chip->mc.host_cores)
            for (const MappedChip::HostCore::Neuron &hn : hc.neurons) out[hn.gid] = chip->hcores->final_status()[i++];
    }
    return 0;
}
extern "C" int sanafe_chip_get_potentials(sanafe_chip *chip, double *out)
{
    if (!chip || !out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    const int rc = gather_by_gid(chip, out, [&](double *p) { return sanafe_hip_read_potentials(chip->dev, p); });
    if (rc != 0) return rc;
    for (const MappedChip::HostNeuron &hn : chip->mc.host_neurons) // plugin somas keep their own state
        out[hn.gid] = chip->plugin_units[hn.unit]->get_potential(hn.addr);
    if (chip->hcores) // so do the units of the cores that run on the host
    {
        size_t i = 0;
        for (const MappedChip::HostCore &hc : chip->mc.host_cores)
            for (const MappedChip::HostCore::Neuron &hn : hc.neurons) out[hn.gid] = chip->hcores->potential(i++);
    }
    return 0;
}
extern "C" int sanafe_chip_get_input_current(sanafe_chip *chip, double *out)
{
    if (!chip || !out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    return gather_by_gid(chip, out, [&](double *p) { return sanafe_hip_read_input_current(chip->dev, p); });
}

extern "C" int sanafe_chip_get_step_totals(sanafe_chip *chip, int64_t first, int64_t count, sanafe_hip_totals *out)
{
    if (!chip || !out || !chip->have_records || first < 0 || count < 0 || first + count > chip->rec_count)
        return fail(SANAFE_HIP_ERR_INVALID, "step records not available (sim with record=1)");
    std::copy(chip->rec_totals.begin() + first, chip->rec_totals.begin() + first + count, out);
    return 0;
}

extern "C" int sanafe_chip_set_state_log(sanafe_chip *chip, int64_t n_v, const int64_t *neurons_v, int64_t n_u, const int64_t *neurons_u)
{
    if (!chip || n_v < 0 || n_u < 0 || (n_v > 0 && !neurons_v) || (n_u > 0 && !neurons_u)) return fail(SANAFE_HIP_ERR_INVALID, "bad arguments");
    if (!chip->dev) return fail(SANAFE_HIP_ERR_NO_DEVICE, "the chip has no device (mapped only)");
    const MappedChip &mc = chip->mc;
    std::vector<uint32_t> sv, su;
    // a tile-sharded chip logs the neurons THIS rank holds (all ranks are given the same lists); sim() gathers the rows
    const int n_ranks = std::max(1, chip->n_ranks);
    std::vector<uint32_t> n_v_rank(n_ranks, 0), n_u_rank(n_ranks, 0);
    std::vector<int32_t> owner;
    std::vector<uint32_t> column;
    auto owner_of = [&](uint32_t slot) {
        int r = 0;
        while (r + 1 < n_ranks && !mc.rank_slot_begin.empty() && slot >= mc.rank_slot_begin[r + 1]) r++;
        return r;
    };
    auto to_slots = [&](const int64_t *g, int64_t n, std::vector<uint32_t> &out, std::vector<uint32_t> &per_rank) {
        for (int64_t i = 0; i < n; i++)
        {
            if (g[i] < 0 || g[i] >= chip->n_neurons) return false;
            const uint32_t s = mc.slot_of_gid[g[i]];
            const int r = n_ranks > 1 ? owner_of(s) : 0;
            owner.push_back(r);
            column.push_back(per_rank[r]++);
            if (s >= mc.slot_offset && s < mc.slot_offset + mc.n_slots) out.push_back(s - mc.slot_offset);
            else if (n_ranks == 1) return false;
        }
        return true;
    };
    if (!to_slots(neurons_v, n_v, sv, n_v_rank) || !to_slots(neurons_u, n_u, su, n_u_rank))
        return fail(SANAFE_HIP_ERR_INVALID, "logged neuron id out of range");
    // a rank's row: its potentials, then its currents -- the u columns follow the rank's v columns
    for (int64_t i = 0; i < n_u; i++) column[static_cast<size_t>(n_v + i)] += n_v_rank[owner[static_cast<size_t>(n_v + i)]];
    chip->log_owner = owner;
    chip->log_column = column;
    chip->log_rank_columns.assign(n_ranks, 0);
    for (int r = 0; r < n_ranks; r++) chip->log_rank_columns[r] = n_v_rank[r] + n_u_rank[r];
    DEV(sanafe_hip_set_state_log(chip->dev, static_cast<uint32_t>(sv.size()), sv.data(), static_cast<uint32_t>(su.size()), su.data()));
    chip->log_v_gids.assign(neurons_v, neurons_v + n_v);
    chip->log_u_gids.assign(neurons_u, neurons_u + n_u);
    return 0;
}
extern "C" int sanafe_chip_get_step_state(sanafe_chip *chip, int64_t first, int64_t count, double *out)
{
    const size_t row = chip ? chip->log_v_gids.size() + chip->log_u_gids.size() : 0;
    if (!chip || !out || first < 0 || count < 0 || row == 0 || static_cast<size_t>(first + count) * row > chip->rec_state.size())
        return fail(SANAFE_HIP_ERR_INVALID, "state records not available (sim with SANAFE_RECORD_STATE)");
    std::copy(chip->rec_state.begin() + first * row, chip->rec_state.begin() + (first + count) * row, out);
    return 0;
}

// (a tile-sharded chip's optional columns are the whole chip's: computed on the twin's tables from the gathered statuses)
static const MappedChip::LogPlan &log_plan(const sanafe_chip *chip) { return (chip->n_ranks > 1 && chip->whole) ? chip->whole->mc.log : chip->mc.log; }

extern "C" int sanafe_chip_wants_perf_columns(sanafe_chip *chip) { return (chip && chip->mc.log.any) ? 1 : 0; }
extern "C" int64_t sanafe_chip_perf_columns(sanafe_chip *chip, char *names, int64_t cap)
{
    if (!chip) return -1;
    int64_t pos = 0;
    for (const MappedChip::LogPlan::Column &col : log_plan(chip).columns)
    {
        const int64_t n = static_cast<int64_t>(col.name.size()) + 1;
        if (names && pos + n <= cap) std::memcpy(names + pos, col.name.c_str(), static_cast<size_t>(n));
        pos += n;
    }
    return static_cast<int64_t>(log_plan(chip).columns.size());
}
extern "C" int sanafe_chip_get_step_optional(sanafe_chip *chip, int64_t first, int64_t count, double *out)
{
    if (!chip || !out || !chip->have_records || first < 0 || count < 0 || first + count > static_cast<int64_t>(chip->rec_optional.size()))
        return fail(SANAFE_HIP_ERR_INVALID, "optional perf columns not recorded (sim with record=1 on an architecture with log flags)");
    const size_t n = log_plan(chip).columns.size();
    for (int64_t k = 0; k < count; k++) std::copy(chip->rec_optional[first + k].begin(), chip->rec_optional[first + k].end(), out + k * n);
    return 0;
}

extern "C" int sanafe_chip_get_step_fired(sanafe_chip *chip, int64_t index, uint8_t *out)
{
    if (!chip || !out || !chip->have_records || index < 0 || index >= chip->rec_count)
        return fail(SANAFE_HIP_ERR_INVALID, "step records not available (sim with record=1)");
    const MappedChip &mc = chip->mc;
    const std::vector<uint32_t> &bits = chip->rec_spike_bits[index];
    for (int64_t g = 0; g < chip->n_neurons; g++)
    {
        const uint32_t s = mc.slot_of_gid[g];
        if (chip->rec_bits_global) // gathered from all ranks: every neuron of the chip
        {
            out[g] = (bits[s >> 5] >> (s & 31u)) & 1u;
            continue;
        }
        if (s < mc.slot_offset || s >= mc.slot_offset + mc.n_slots)
        {
            out[g] = 0;
            continue;
        }
        const uint32_t ls = s - mc.slot_offset;
        out[g] = (bits[ls >> 5] >> (ls & 31u)) & 1u;
    }
    return 0;
}

extern "C" int64_t sanafe_chip_get_step_messages(sanafe_chip *chip, int64_t index, sanafe_message *out, int64_t cap)
{
    if (!chip || !chip->have_records || index < 0 || index >= static_cast<int64_t>(chip->rec_messages.size())) return -1;
    const std::vector<Msg> &v = chip->rec_messages[index];
    if (out)
        for (int64_t i = 0; i < std::min<int64_t>(cap, v.size()); i++) out[i] = static_cast<const sanafe_message &>(v[i]);
    return static_cast<int64_t>(v.size());
}

extern "C" int sanafe_chip_set_bias(sanafe_chip *chip, int64_t count, const int64_t *neurons, const double *bias)
{
    if (!chip || (count > 0 && (!neurons || !bias))) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    const MappedChip &mc = chip->mc;
    for (int64_t i = 0; i < count; i++)
    {
        if (neurons[i] < 0 || neurons[i] >= chip->n_neurons) return fail(SANAFE_HIP_ERR_INVALID, "neuron id out of range");
        const uint32_t s = mc.slot_of_gid[neurons[i]];
        if (s < mc.slot_offset || s >= mc.slot_offset + mc.n_slots) continue;
        DEV(sanafe_hip_write_bias(chip->dev, s - mc.slot_offset, 1, &bias[i]));
    }
    return 0;
}

extern "C" int sanafe_chip_set_attribute(sanafe_chip *chip, int64_t neuron, const char *key, int type, double num, const char *str)
{
    if (!chip || !key) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    // (mapped-only chips take the attributes that only touch the host's value streams: the host-side checks of those)
    if (!chip->dev && std::strcmp(key, "poisson") != 0 && std::strcmp(key, "random_mask") != 0)
        return fail(SANAFE_HIP_ERR_INVALID, "the chip has no device (mapped only)");
    if (neuron < 0 || neuron >= chip->n_neurons) return fail(SANAFE_HIP_ERR_INVALID, "neuron id out of range");
    MappedChip &mc = chip->mc;
    const uint32_t s = mc.slot_of_gid[neuron];
    if (s < mc.slot_offset || s >= mc.slot_offset + mc.n_slots) return 0; // another rank's neuron
    const uint32_t ls = s - mc.slot_offset;
    const uint32_t model = mc.slot_model[ls];
    try
    {
        if (model == SANAFE_SOMA_LIF || model == SANAFE_SOMA_TRUENORTH || model == SANAFE_SOMA_PERSIST)
        {
            sanafe_amd::SomaAttr a;
            a.key = key;
            a.type = type;
            a.num = num;
            if (type == SANAFE_ATTR_STRING) a.str = str ? str : "";
            if (type == SANAFE_ATTR_LIST) return 0; // no list-valued attribute on these models: ignored like any unknown key
            const uint32_t cls = mc.slot_cls[ls];
            sanafe_hip_soma_class p = mc.soma_classes[cls >> 16];
            sanafe_amd::SomaAttrEffect fx;
            sanafe_amd::apply_soma_attribute(model == SANAFE_SOMA_PERSIST ? static_cast<uint32_t>(SANAFE_SOMA_TRUENORTH) : model, a, p, fx);
            if (fx.random_mask_set)
            {
                // TrueNorthModel::update draws std::rand() only for a neuron with a mask (src/models.cpp:752-758): a mask that
                // comes or goes changes which neurons draw -- the value-stream columns follow
                const bool has_column = !mc.slot_ext.empty() && mc.slot_ext[ls] != 0xffffffffu;
                const uint32_t old_mask = has_column ? mc.ext[mc.slot_ext[ls]].mask : 0u;
                if (fx.random_mask != old_mask)
                {
                    if (has_column && fx.random_mask != 0u) mc.ext[mc.slot_ext[ls]].mask = fx.random_mask;
                    else if (chip->n_ranks != 1)
                        return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: a random_mask cannot come or go after load() on a tile-sharded "
                                                                "chip (it defines the chip's rand() schedule)");
                    else if (has_column) chip->change_ext_column(ls, nullptr);
                    else
                    {
                        MappedChip::ExtColumn col;
                        col.slot = ls;
                        col.kind = MappedChip::ExtColumn::TrueNorthRand;
                        col.mask = fx.random_mask;
                        chip->change_ext_column(ls, &col);
                    }
                }
            }
            if (fx.bias_set)
            {
                mc.slot_bias[ls] = fx.bias;
                DEV(sanafe_hip_write_bias(chip->dev, ls, 1, &fx.bias));
            }
            if (fx.potential_set) DEV(sanafe_hip_write_potential(chip->dev, ls, 1, &fx.potential));
            const sanafe_hip_soma_class canon = sanafe_amd::canonical_soma_class(p);
            if (chip->class_ids.empty())
                for (size_t k = 0; k < mc.soma_classes.size(); k++) chip->class_ids.emplace(sanafe_chip::class_key(mc.soma_classes[k]), static_cast<uint32_t>(k));
            auto it = chip->class_ids.find(sanafe_chip::class_key(canon));
            if (it == chip->class_ids.end())
            {
                if (mc.soma_classes.size() >= 65536) return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: more than 65536 distinct soma parameter sets");
                it = chip->class_ids.emplace(sanafe_chip::class_key(canon), static_cast<uint32_t>(mc.soma_classes.size())).first;
                mc.soma_classes.push_back(canon);
                chip->classes_dirty = true;
            }
            const uint32_t new_cls = (cls & 0xffffu) | (it->second << 16);
            if (new_cls != cls)
            {
                mc.slot_cls[ls] = new_cls;
                chip->dirty_slots.push_back(ls);
            }
            return 0;
        }
        if (model == SANAFE_SOMA_HOST)
        {
            if (chip->hcores)
            {
                sanafe::ModelAttribute ma;
                ma.name = key;
                if (type == SANAFE_ATTR_BOOL) ma.value = (num != 0.0);
                else if (type == SANAFE_ATTR_INT) ma.value = static_cast<int>(num);
                else if (type == SANAFE_ATTR_DOUBLE) ma.value = num;
                else if (type == SANAFE_ATTR_STRING) ma.value = std::string(str ? str : "");
                else return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: list attributes cannot be patched on a mapped neuron");
                if (chip->hcores->set_attribute(ls, ma)) return 0;
            }
            for (const MappedChip::HostNeuron &hn : mc.host_neurons)
                if (hn.slot == ls)
                {
                    sanafe::ModelAttribute ma;
                    ma.name = key;
                    if (type == SANAFE_ATTR_BOOL) ma.value = (num != 0.0);
                    else if (type == SANAFE_ATTR_INT) ma.value = static_cast<int>(num);
                    else if (type == SANAFE_ATTR_DOUBLE) ma.value = num;
                    else if (type == SANAFE_ATTR_STRING) ma.value = std::string(str ? str : "");
                    else return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: list attributes cannot be patched on a mapped neuron");
                    chip->plugin_units[hn.unit]->set_attribute_neuron(hn.addr, std::string(key), ma);
                    return 0;
                }
            return 0;
        }
        if (model == SANAFE_SOMA_INPUT) return chip->set_input_attribute(ls, key, type, num, nullptr, 0);
    }
    catch (const std::exception &e)
    {
        return fail(SANAFE_HIP_ERR_INVALID, e.what());
    }
    return 0;
}

extern "C" int sanafe_chip_set_attribute_list(sanafe_chip *chip, int64_t neuron, const char *key, const double *values, int64_t count)
{
    if (!chip || !key || count < 0 || (count > 0 && !values)) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    if (!chip->dev) return fail(SANAFE_HIP_ERR_INVALID, "the chip has no device (mapped only)");
    if (neuron < 0 || neuron >= chip->n_neurons) return fail(SANAFE_HIP_ERR_INVALID, "neuron id out of range");
    MappedChip &mc = chip->mc;
    const uint32_t s = mc.slot_of_gid[neuron];
    if (s < mc.slot_offset || s >= mc.slot_offset + mc.n_slots) return 0; // another rank's neuron
    const uint32_t ls = s - mc.slot_offset;
    if (mc.slot_model[ls] == SANAFE_SOMA_INPUT) return chip->set_input_attribute(ls, key, SANAFE_ATTR_LIST, 0.0, values, count);
    if (mc.slot_model[ls] == SANAFE_SOMA_HOST)
        return fail(SANAFE_HIP_ERR_UNSUPPORTED, "UnsupportedError: list attributes cannot be patched on a plugin neuron after load()");
    return 0; // LIF / TrueNorth have no list-valued attribute: ignored like any unknown key
}

extern "C" int sanafe_chip_commit_attributes(sanafe_chip *chip)
{
    if (!chip) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    return chip->dev ? chip->commit_attributes() : 0;
}

extern "C" int sanafe_chip_step_neurons(sanafe_chip *chip)
{
    if (!chip) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    if (int rc = chip->commit_attributes()) return rc;
    if (int rc = chip->queue_ext(1)) return rc;
    DEV(sanafe_hip_step_neurons(chip->dev));
    return 0;
}
extern "C" int sanafe_chip_step_deliver(sanafe_chip *chip, int timing_model)
{
    if (!chip) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    DEV(sanafe_hip_step_deliver(chip->dev, timing_model == SANAFE_TIMING_SIMPLE, 0));
    chip->total_timesteps += 1;
    return 0;
}
extern "C" int sanafe_chip_spike_buffers(sanafe_chip *chip, void **local_bits, uint64_t *local_bytes, void **global_bits,
        uint64_t *global_bytes, uint64_t *local_offset_bytes)
{
    if (!chip) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    DEV(sanafe_hip_spike_buffers(chip->dev, local_bits, local_bytes, global_bits, global_bytes));
    if (local_offset_bytes) *local_offset_bytes = chip->mc.slot_offset / 8;
    return 0;
}
extern "C" int sanafe_chip_synchronize(sanafe_chip *chip)
{
    if (!chip) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    DEV(sanafe_hip_synchronize(chip->dev));
    return 0;
}
extern "C" int64_t sanafe_chip_total_timesteps(sanafe_chip *chip) { return chip ? chip->total_timesteps : 0; }
extern "C" int sanafe_chip_read_totals(sanafe_chip *chip, sanafe_hip_totals *out)
{
    if (!chip || !out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    DEV(sanafe_hip_read_totals(chip->dev, out));
    return 0;
}
