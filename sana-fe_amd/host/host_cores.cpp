// host_cores.cpp -- see host_cores.hpp.
#include "host_cores.hpp"

#include <dlfcn.h>

#include <map>
#include <stdexcept>

namespace sanafe_amd
{
namespace
{
using sanafe::ModelAttribute;
using sanafe::PipelineResult;

ModelAttribute to_model_attribute(const sanafe_desc &d, const sanafe_attr_table &t, int64_t i)
{
    ModelAttribute a;
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
        std::vector<ModelAttribute> v;
        for (int64_t k = t.list_ptr[i]; k < t.list_ptr[i + 1]; k++)
        {
            ModelAttribute e;
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
} // namespace

HostCores::HostCores(const MappedChip &mc, const sanafe_desc &d)
{
    std::map<std::string, void *> libs;
    for (const MappedChip::HostCore &hc : mc.host_cores)
    {
        cores_.emplace_back();
        CoreRt &c = cores_.back();
        c.hc = &hc;
        // ---- Core::create_pipeline_unit for every unit of the core, src/core.cpp:196-231 ----
        for (const MappedChip::HostCore::Unit &info : hc.units)
        {
            UnitRt u;
            u.info = &info;
            if (!info.plugin_path.empty())
            {
                void *&lib = libs[info.plugin_path];
                if (!lib)
                {
                    lib = dlopen(info.plugin_path.c_str(), RTLD_LAZY | RTLD_LOCAL);
                    if (!lib) throw std::runtime_error(std::string("Error: Couldn't load library ") + info.plugin_path + ": " + dlerror());
                    plugin_handles_.push_back(lib);
                }
                using Factory = sanafe::PipelineUnit *(*) ();
                const std::string sym = "create_" + info.model;
                dlerror();
                auto create = reinterpret_cast<Factory>(dlsym(lib, sym.c_str()));
                if (!create) throw std::runtime_error("Error: Couldn't load symbol " + sym + " from " + info.plugin_path);
                u.obj.reset(create());
                u.obj->plugin_lib = info.plugin_path;
            }
            else
            {
                u.obj = make_builtin_host_unit(info.model);
            }
            // check_implemented, src/pipeline.cpp:20-57
            if (u.obj->implements_synapse != info.syn || u.obj->implements_dendrite != info.dend || u.obj->implements_soma != info.soma)
                throw std::runtime_error("Unit '" + info.name + "' (" + info.model + ") is listed in a hardware section it does not implement");
            // set_attributes_hw, src/pipeline.cpp:151-175
            u.obj->name = info.name;
            u.obj->model = info.model;
            u.obj->update_every_timestep = info.update_every_timestep;
            u.obj->default_energy_process_spike = info.e_spike;
            u.obj->default_latency_process_spike = info.l_spike;
            u.obj->default_energy_update = info.e_update;
            u.obj->default_latency_update = info.l_update;
            if (info.has_soma_e) u.obj->default_soma_energy_metrics = sanafe::SomaEnergyMetrics{info.se[1], info.se[0], info.se[2]};
            if (info.has_soma_l) u.obj->default_soma_latency_metrics = sanafe::SomaLatencyMetrics{info.sl[1], info.sl[0], info.sl[2]};
            for (int64_t i = d.unit_attr_ptr[info.desc_unit]; i < d.unit_attr_ptr[info.desc_unit + 1]; i++)
            {
                const ModelAttribute a = to_model_attribute(d, d.unit_attrs, i);
                u.obj->model_attributes[*a.name] = a;
            }
            for (const auto &kv : u.obj->model_attributes) u.obj->set_attribute_hw(kv.first, kv.second); // key order
            c.units.push_back(std::move(u));
        }
        // ---- Core::map_neuron + MappedNeuron::set_attributes, in mapped order ----
        c.first = slots_.size();
        c.buffer.assign(hc.neurons.size(), PipelineResult{});
        for (const MappedChip::HostCore::Neuron &hn : hc.neurons)
        {
            NeuronRt n;
            n.soma_unit = hn.soma_unit;
            n.dend_unit = hn.dend_unit;
            n.soma_addr = hn.soma_addr;
            n.dend_addr = hn.dend_addr;
            UnitRt &du = c.units.at(hn.dend_unit), &su = c.units.at(hn.soma_unit);
            du.obj->add_neuron();
            du.used = true;
            if (hn.soma_unit != hn.dend_unit) su.obj->add_neuron();
            su.used = true;
            for (int64_t i = d.neuron_attr_ptr[hn.gid]; i < d.neuron_attr_ptr[hn.gid + 1]; i++)
            {
                const ModelAttribute a = to_model_attribute(d, d.neuron_attrs, i);
                if (a.forward_to_dendrite) du.obj->set_attribute_neuron(hn.dend_addr, *a.name, a);
                if (a.forward_to_soma) su.obj->set_attribute_neuron(hn.soma_addr, *a.name, a);
            }
            // build_neuron_processing_pipeline, src/mapped.cpp:168-188
            bool dend_added = false;
            if (hc.bp <= SANAFE_BUF_INSIDE_DENDRITE)
            {
                n.pipeline.push_back(hn.dend_unit);
                dend_added = true;
            }
            if (hc.bp <= SANAFE_BUF_INSIDE_SOMA && (hn.soma_unit != hn.dend_unit || !dend_added)) n.pipeline.push_back(hn.soma_unit);
            c.neurons.push_back(std::move(n));
            slots_.push_back(hn.slot);
            slot_cores_.push_back(hc.core - mc.first_core);
        }
        // ---- Core::map_connection + MappedConnection::set_attributes, src/core.cpp:170-184, src/mapped.cpp:27-89 ----
        for (size_t k = 0; k < hc.synapses.size(); k++)
        {
            const MappedChip::HostCore::Synapse &hs = hc.synapses[k];
            UnitRt &sy = c.units.at(hs.unit);
            NeuronRt &post = c.neurons.at(hs.post);
            sy.used = true;
            sy.obj->connection_count = std::max<long>(sy.obj->connection_count, static_cast<long>(hs.addr) + 1);
            sy.obj->is_used = true;
            sy.obj->track_connection(hs.addr, static_cast<size_t>(d.edge_src[hs.edge]), static_cast<size_t>(d.edge_dst[hs.edge]));
            post.check_synapse_updates = post.check_synapse_updates || sy.obj->update_every_timestep;
            UnitRt &de = c.units.at(post.dend_unit);
            auto forward = [&](const ModelAttribute &a) {
                if (a.forward_to_synapse) sy.obj->set_attribute_edge(hs.addr, *a.name, a);
                if (a.forward_to_dendrite) de.obj->set_attribute_edge(hs.addr, *a.name, a);
            };
            ModelAttribute w;
            w.name = "weight";
            w.value = d.edge_weight[hs.edge];
            forward(w);
            if (d.edge_delay && d.edge_delay[hs.edge] >= 0)
            {
                ModelAttribute dl;
                const int v = d.edge_delay[hs.edge];
                dl.name = v >= 64 ? "tap" : "delay"; // include/sanafe_desc.h: 64 + tap index
                dl.value = v >= 64 ? v - 64 : v;
                forward(dl);
            }
            if (d.edge_attr_ptr)
                for (int64_t i = d.edge_attr_ptr[hs.edge]; i < d.edge_attr_ptr[hs.edge + 1]; i++) forward(to_model_attribute(d, d.edge_attrs, i));
        }
        core_ids_.push_back(hc.core - mc.first_core);
    }
    status_.assign(slots_.size(), 0);
    final_status_.assign(slots_.size(), 0);
    partials_.assign(cores_.size(), Partial{});
}

HostCores::~HostCores()
{
    cores_.clear(); // destroy the unit objects before their code is unloaded
    for (void *h : plugin_handles_) dlclose(h);
}

void HostCores::begin_step()
{
    for (CoreRt &c : cores_)
        for (UnitRt &u : c.units)
        {
            u.obj->energy = u.obj->latency = 0.0;
            u.obj->spikes_processed = u.obj->neurons_updated = u.obj->neurons_fired = 0;
        }
    for (Partial &p : partials_) p = Partial{};
}

// PipelineUnit::process, src/pipeline.cpp:87-105: the input interface of the unit's FIRST role, the output (costing)
// interface of its LAST role (src/pipeline.hpp:313-404), default costs from the architecture (src/pipeline.hpp:511-731).
PipelineResult HostCores::process(CoreRt &c, int32_t unit, long t, NeuronRt &n, const MappedChip::HostCore::Synapse *con, const PipelineResult &in)
{
    UnitRt &u = c.units[unit];
    sanafe::PipelineUnit &hw = *u.obj;
    PipelineResult out;
    if (hw.implements_synapse)
    {
        out = hw.update(con ? static_cast<size_t>(con->addr) : 0UL, con != nullptr, static_cast<long int>(t));
        ++hw.spikes_processed;
    }
    else if (hw.implements_dendrite)
    {
        out = hw.update(static_cast<size_t>(n.dend_addr), in.current, con ? std::optional<size_t>(con->addr) : std::nullopt, static_cast<long int>(t));
    }
    else
    {
        out = hw.update(static_cast<size_t>(n.soma_addr), in.current, static_cast<long int>(t));
    }
    auto apply = [&](const char *what, bool has_default_e, double def_e, bool has_default_l, double def_l) {
        if (out.energy.has_value() && has_default_e)
            throw std::runtime_error(std::string(what) + " unit simulates energy and also has default energy metrics set.");
        if (has_default_e) out.energy = def_e;
        if (out.latency.has_value() && has_default_l)
            throw std::runtime_error(std::string(what) + " unit simulates latency and also has default latency metrics set. Remove the default metric from the architecture description.");
        if (has_default_l) out.latency = def_l;
        if (!out.energy.has_value())
            throw std::runtime_error(std::string(what) + " unit does not simulate energy or provide a default energy cost in the architecture description.");
        if (!out.latency.has_value())
            throw std::runtime_error(std::string(what) + " unit does not simulate latency or provide a default latency cost in the architecture description.");
    };
    if (hw.implements_soma)
    {
        // calculate_soma_default_energy_latency + update_soma_activity: the defaults of the NEURON's soma unit
        const MappedChip::HostCore::Unit &si = *c.units[n.soma_unit].info;
        double e = si.se[0], l = si.sl[0];
        if (out.status == sanafe::updated || out.status == sanafe::fired) e += si.se[1], l += si.sl[1];
        if (out.status == sanafe::fired) e += si.se[2], l += si.sl[2];
        apply("Soma", si.has_soma_e, e, si.has_soma_l, l);
        sanafe::PipelineUnit &soma = *c.units[n.soma_unit].obj;
        if (out.status == sanafe::updated || out.status == sanafe::fired) soma.neurons_updated++;
        if (out.status == sanafe::fired) soma.neurons_fired++;
    }
    else if (hw.implements_dendrite)
    {
        const MappedChip::HostCore::Unit &di = *c.units[n.dend_unit].info;
        apply("Dendrite", di.e_update.has_value(), di.e_update.value_or(0.0), di.l_update.has_value(), di.l_update.value_or(0.0));
    }
    else
    {
        const MappedChip::HostCore::Unit &yi = con ? *c.units[con->unit].info : *u.info;
        apply("Synapse", yi.e_spike.has_value(), yi.e_spike.value_or(0.0), yi.l_spike.has_value(), yi.l_spike.value_or(0.0));
    }
    hw.energy += out.energy.value_or(0.0);
    hw.latency += out.energy.value_or(0.0); // (sic: src/pipeline.cpp:102)
    return out;
}

// execute_pipeline, src/chip.cpp:766-789
PipelineResult HostCores::execute(CoreRt &c, const int32_t *pipeline, size_t len, long t, NeuronRt &n, const MappedChip::HostCore::Synapse *con,
        const PipelineResult &in)
{
    double energy = 0.0, latency = 0.0;
    PipelineResult out{in};
    for (size_t k = 0; k < len; k++)
    {
        out = process(c, pipeline[k], t, n, con, out);
        energy += out.energy.value_or(0.0);
        latency += out.latency.value_or(0.0);
        if (out.status != sanafe::neuron_state_unset) n.status = out.status;
    }
    out.energy = energy;
    out.latency = latency;
    return out;
}

// process_neurons / process_neuron for the host cores, src/chip.cpp:624-654, 710-736
void HostCores::process_neurons(long t)
{
    for (size_t ci = 0; ci < cores_.size(); ci++)
    {
        CoreRt &c = cores_[ci];
        const bool kernel_buffer = c.hc->bp == SANAFE_BUF_BEFORE_DENDRITE || c.hc->bp == SANAFE_BUF_BEFORE_SOMA;
        for (size_t k = 0; k < c.neurons.size(); k++)
        {
            NeuronRt &n = c.neurons[k];
            PipelineResult in{};
            if (kernel_buffer)
            {
                in = c.buffer[k];
                c.buffer[k] = PipelineResult{};
            }
            const PipelineResult out = execute(c, n.pipeline.data(), n.pipeline.size(), t, n, nullptr, in);
            partials_[ci].neuron_latency += out.latency.value_or(0.0);
            status_[c.first + k] = static_cast<uint8_t>(n.status); // (a status no unit refreshed persists: buffer before axon_out)
        }
    }
}

// process_messages / process_message for the host cores, src/chip.cpp:656-692, 738-764: inbound axons in delivery order
void HostCores::process_messages(long t, const uint32_t *bits)
{
    for (size_t ci = 0; ci < cores_.size(); ci++)
    {
        CoreRt &c = cores_[ci];
        const MappedChip::HostCore &hc = *c.hc;
        for (const MappedChip::HostCore::Axon &ax : hc.axons)
        {
            if (!((bits[ax.pre >> 5] >> (ax.pre & 31u)) & 1u)) continue;
            double latency = hc.ain_latency; // pipeline_process_axon_in
            for (uint32_t k = ax.syn_beg; k < ax.syn_beg + ax.n_syn; k++)
            {
                const MappedChip::HostCore::Synapse &hs = hc.synapses[k];
                NeuronRt &n = c.neurons[hs.post];
                int32_t pipe[3];
                size_t len = 0;
                pipe[len++] = hs.unit; // build_message_processing_pipeline, src/mapped.cpp:27-58
                if (hc.bp > SANAFE_BUF_BEFORE_DENDRITE && n.dend_unit != hs.unit) pipe[len++] = n.dend_unit;
                if (hc.bp > SANAFE_BUF_BEFORE_SOMA && n.soma_unit != n.dend_unit) pipe[len++] = n.soma_unit;
                const PipelineResult out = execute(c, pipe, len, t, n, &hs, PipelineResult{});
                c.buffer[hs.post] = out;
                latency += out.latency.value_or(0.0);
            }
            partials_[ci].processing += latency;
        }
    }
}

void HostCores::forced_updates(long t)
{
    for (CoreRt &c : cores_)
    {
        const MappedChip::HostCore &hc = *c.hc;
        // synapse units flagged update_every_timestep: the reference walks the connections of every neuron that itself
        // receives through such a unit (src/chip.cpp:989-1005; the flag is set on the post neuron, src/mapped.cpp:32-40)
        for (size_t k = 0; k < hc.synapses.size(); k++)
        {
            const MappedChip::HostCore::Synapse &hs = hc.synapses[k];
            sanafe::PipelineUnit &sy = *c.units[hs.unit].obj;
            if (!sy.update_every_timestep || !hs.pre_checks_synapses) continue;
            const PipelineResult r = sy.update(static_cast<size_t>(hs.addr), false, static_cast<long int>(t));
            if (r.energy.has_value()) sy.energy += *r.energy;
        }
        for (NeuronRt &n : c.neurons)
        {
            sanafe::PipelineUnit &de = *c.units[n.dend_unit].obj;
            if (!de.update_every_timestep) continue;
            const PipelineResult r = de.update(static_cast<size_t>(n.dend_addr), std::nullopt, std::nullopt, static_cast<long int>(t));
            if (r.energy.has_value()) de.energy += *r.energy;
        }
    }
}

// sim_calculate_core_energy, src/chip.cpp:1207-1261: every in-use unit's energy, binned by the roles it implements
void HostCores::end_step()
{
    for (size_t ci = 0; ci < cores_.size(); ci++)
    {
        Partial &p = partials_[ci];
        for (size_t k = 0; k < cores_[ci].neurons.size(); k++) final_status_[cores_[ci].first + k] = static_cast<uint8_t>(cores_[ci].neurons[k].status);
        for (const UnitRt &u : cores_[ci].units)
        {
            if (!u.used) continue;
            if (u.obj->implements_synapse) p.e_syn += u.obj->energy;
            if (u.obj->implements_dendrite) p.e_dend += u.obj->energy;
            if (u.obj->implements_soma)
            {
                p.e_soma += u.obj->energy;
                p.updated += u.obj->neurons_updated;
                p.fired += u.obj->neurons_fired;
            }
        }
    }
}

void HostCores::reset()
{
    for (CoreRt &c : cores_)
    {
        std::fill(c.buffer.begin(), c.buffer.end(), PipelineResult{});
        for (UnitRt &u : c.units) u.obj->reset();
        for (NeuronRt &n : c.neurons) n.status = sanafe::neuron_state_unset;
    }
}

double HostCores::potential(size_t i) const
{
    for (const CoreRt &c : cores_)
        if (i >= c.first && i < c.first + c.neurons.size())
        {
            const NeuronRt &n = c.neurons[i - c.first];
            return c.units[n.soma_unit].obj->get_potential(n.soma_addr);
        }
    return 0.0;
}

bool HostCores::set_attribute(uint32_t slot, const sanafe::ModelAttribute &a)
{
    for (size_t i = 0; i < slots_.size(); i++)
    {
        if (slots_[i] != slot) continue;
        for (CoreRt &c : cores_)
            if (i >= c.first && i < c.first + c.neurons.size())
            {
                NeuronRt &n = c.neurons[i - c.first];
                if (a.forward_to_dendrite) c.units[n.dend_unit].obj->set_attribute_neuron(n.dend_addr, *a.name, a);
                if (a.forward_to_soma) c.units[n.soma_unit].obj->set_attribute_neuron(n.soma_addr, *a.name, a);
            }
        return true;
    }
    return false;
}
} // namespace sanafe_amd
