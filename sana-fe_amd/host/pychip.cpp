// pychip.cpp -- `SpikingChip` of the PyBind11 module: the reference's Python entry point for the simulation loop
// (class SpikingChip + pysim, src/pymodule.cpp:549-706, 1170-1212) as compiled C++ over libsanafe_host.so.
//
//   sim() runs sanafe_chip_sim in chunks with the GIL RELEASED and polls PyErr_CheckSignals between chunks (Ctrl-C
//   works, src/pymodule.cpp:628-666); after every chunk the recorded steps are moved into the trace objects, so
//   traces stream to their files as the run proceeds (PyTrace, src/pytrace.cpp:76-129: None / True = in memory /
//   filename / object with .write) and memory stays bounded.  Potential and neuron traces come from the device-side
//   state log (no per-step host loop).  mapped_neuron_groups[name][i].set_attributes(...) forwards to
//   sanafe_chip_set_attribute (MappedNeuron::set_attributes, src/mapped.cpp:113-166).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sanafe_host.h"
#include "description.hpp"
#include "pyconv.hpp"

namespace py = pybind11;
using namespace sanafe_amd;

namespace
{
[[noreturn]] void raise_last(const char *fallback = "libsanafe_host call failed")
{
    std::string msg = sanafe_last_error();
    if (msg.empty()) msg = fallback;
    if (msg.rfind("UnsupportedError", 0) == 0)
    {
        PyErr_SetString(PyExc_NotImplementedError, msg.c_str());
        throw py::error_already_set();
    }
    if (msg.rfind("HardwareMappingError", 0) == 0)
    {
        py::object cls = py::module_::import("sanafe_amd.chip").attr("HardwareMappingError");
        PyErr_SetString(cls.ptr(), msg.c_str());
        throw py::error_already_set();
    }
    throw std::runtime_error(msg);
}
void check(int rc)
{
    if (rc != 0) raise_last();
}

// One trace argument of sim(): None, True (in memory), a filename, or an object with .write (src/pytrace.cpp:76-129).
struct Trace
{
    enum Mode { None, Memory, File, Object } mode{None};
    std::ofstream file;
    py::object obj;
    void open(const py::object &arg, bool write_headers)
    {
        if (arg.is_none()) return;
        if (py::isinstance<py::bool_>(arg))
        {
            if (arg.cast<bool>()) mode = Memory;
        }
        else if (py::hasattr(arg, "write"))
        {
            mode = Object;
            obj = arg;
            if (write_headers && py::hasattr(arg, "seek")) arg.attr("seek")(0);
        }
        else if (py::isinstance<py::str>(arg))
        {
            mode = File;
            const std::string name = arg.cast<std::string>();
            file.open(name, write_headers ? (std::ios::out | std::ios::trunc) : (std::ios::out | std::ios::app));
            if (!file.is_open()) throw std::runtime_error("Failed to open trace file: " + name);
        }
        else
        {
            throw std::invalid_argument("trace_obj must be None, True, a filename string, or a file-like object");
        }
    }
    bool on() const { return mode != None; }
    bool to_stream() const { return mode == File || mode == Object; }
    void write(const std::string &text)
    {
        if (mode == File) file << text;
        else if (mode == Object) obj.attr("write")(py::str(text));
    }
    void close()
    {
        if (mode == File) file.close();
    }
};

std::string fmt_g(double v)
{
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%g", v);
    return buf;
}
std::string fmt_e(double v)
{
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%e", v);
    return buf;
}

class Chip;

struct MappedNeuronHandle // chip.mapped_neuron_groups[name][i]
{
    py::object owner; // the chip
    Chip *chip;
    int64_t gid;
};
struct GroupView // a mapped group: sequence of MappedNeuronHandle
{
    py::object owner;
    Chip *chip;
    int64_t base, count;
};

class Chip
{
public:
    py::object arch;
    std::vector<py::object> nets; // networks loaded so far (load(net, overwrite=False) adds)
    py::object keepalive;         // lowered description (and whatever it borrows from)
    uintptr_t desc_address_{0};   // the description the chip was created from (kept alive by `keepalive`)
    bool carry_state{false};      // the next adopt() moves the run-time state of the programmed chip into the new one
    bool whole_attached{false};   // tile-sharded chips: the whole-chip twin exists (sanafe_chip_attach_whole)
    sanafe_chip *h{nullptr};
    int device, n_ranks, rank;
    int64_t n_neurons{0};
    std::vector<std::string> group_names;
    std::vector<int64_t> group_base, group_count;
    std::vector<uint8_t> log_spikes, log_potential;
    std::vector<int64_t> trace_order;   // neuron ids in trace order: groups by name, neurons by offset (std::map)
    std::vector<int32_t> group_of_gid;  // -> index into group_names

    Chip(py::object arch_, int device_, int n_ranks_, int rank_) : arch(std::move(arch_)), device(device_), n_ranks(n_ranks_), rank(rank_) {}
    ~Chip() { free_chip(); }
    void free_chip()
    {
        if (h) sanafe_chip_destroy(h);
        h = nullptr;
    }
    void need_chip() const
    {
        if (!h) throw std::runtime_error("no network loaded");
    }

    // ---- SpikingChip::load, src/chip.cpp:129-138; Python default overwrite=False ----
    void adopt(uintptr_t desc_address, const std::vector<std::tuple<std::string, int64_t, int64_t>> &groups, const py::array_t<uint8_t> &ls,
            const py::array_t<uint8_t> &lp, py::object keep)
    {
        sanafe_chip *out = nullptr;
        // (the flag is consumed here, whatever happens below: a load that fails must not leave the next one carrying state)
        const bool carry = carry_state;
        carry_state = false;
        if (carry && h)
        {
            // load(net, overwrite=False) after timesteps have run: the programmed neurons' state moves into the new lowering
            const int rc = sanafe_chip_create(reinterpret_cast<const sanafe_desc *>(desc_address), device, n_ranks, rank, &out);
            if (rc != 0) raise_last("sanafe_chip_create failed");
            if (sanafe_chip_carry_state(out, h) != 0)
            {
                sanafe_chip_destroy(out);
                raise_last("sanafe_chip_carry_state failed");
            }
            free_chip();
        }
        else
        {
            free_chip();
            const int rc = sanafe_chip_create(reinterpret_cast<const sanafe_desc *>(desc_address), device, n_ranks, rank, &out);
            if (rc != 0) raise_last("sanafe_chip_create failed");
        }
        carry_state = false;
        h = out;
        keepalive = std::move(keep);
        desc_address_ = desc_address;
        whole_attached = false;
        n_neurons = reinterpret_cast<const sanafe_desc *>(desc_address)->n_neurons;
        group_names.clear();
        group_base.clear();
        group_count.clear();
        for (const auto &g : groups)
        {
            group_names.push_back(std::get<0>(g));
            group_base.push_back(std::get<1>(g));
            group_count.push_back(std::get<2>(g));
        }
        log_spikes.assign(ls.data(), ls.data() + ls.size());
        log_potential.assign(lp.data(), lp.data() + lp.size());
        std::vector<size_t> order(group_names.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return group_names[a] < group_names[b]; });
        trace_order.clear();
        group_of_gid.assign(static_cast<size_t>(n_neurons), 0);
        for (size_t gi : order)
            for (int64_t k = 0; k < group_count[gi]; k++)
            {
                trace_order.push_back(group_base[gi] + k);
                group_of_gid[static_cast<size_t>(group_base[gi] + k)] = static_cast<int32_t>(gi);
            }
    }
    void load(py::object net, bool overwrite)
    {
        if (!py::isinstance<SpikingNetwork>(net)) throw std::invalid_argument("load() takes a sanafecpp_amd.Network");
        py::object to_lower = net;
        carry_state = false;
        const bool add = h && !overwrite;
        if (add)
        {
            carry_state = sanafe_chip_total_timesteps(h) > 0; // (src/chip.cpp:129-138: every programmed unit keeps its state)
            // the groups of the new network are mapped after the programmed ones (ids, mapping order and per-core
            // offsets continue): re-lower the combination
            py::object merged = py::cast(std::make_unique<SpikingNetwork>(nets.front().cast<SpikingNetwork &>().name));
            SpikingNetwork &m = merged.cast<SpikingNetwork &>();
            for (const py::object &n : nets) m.absorb(n.cast<SpikingNetwork &>());
            m.absorb(net.cast<SpikingNetwork &>());
            to_lower = merged;
        }
        SpikingNetwork &n = to_lower.cast<SpikingNetwork &>();
        auto built = std::shared_ptr<BuiltDesc>(to_desc(arch.cast<Architecture &>(), n).release());
        std::vector<std::tuple<std::string, int64_t, int64_t>> groups;
        py::array_t<uint8_t> ls(n.neuron_count), lp(n.neuron_count);
        for (const auto &g : n.order)
        {
            groups.emplace_back(g->name, g->base, g->count);
            std::copy(g->log_spikes.begin(), g->log_spikes.end(), ls.mutable_data() + g->base);
            std::copy(g->log_potential.begin(), g->log_potential.end(), lp.mutable_data() + g->base);
        }
        // the description borrows the network's edge arrays: keep both alive with the chip
        py::object keep = py::make_tuple(py::capsule(new std::shared_ptr<BuiltDesc>(built), [](void *p) { delete static_cast<std::shared_ptr<BuiltDesc> *>(p); }),
                to_lower, arch);
        try
        {
            adopt(reinterpret_cast<uintptr_t>(&built->desc), groups, ls, lp, keep);
        }
        catch (...)
        {
            carry_state = false; // a refused or failed load leaves the chip as it was: programmed nets unchanged, no pending carry
            throw;
        }
        // only now is the network part of the chip
        if (add) nets.push_back(net);
        else nets.assign(1, net);
    }

    py::tuple label(int64_t gid) const // NeuronAddress(group_name, neuron_offset)
    {
        const int32_t g = group_of_gid[static_cast<size_t>(gid)];
        return py::make_tuple(group_names[g], gid - group_base[g]);
    }
    std::string label_text(int64_t gid) const
    {
        if (gid < 0 || gid >= n_neurons) return "invalid.0";
        const int32_t g = group_of_gid[static_cast<size_t>(gid)];
        return group_names[g] + "." + std::to_string(gid - group_base[g]);
    }

    // Tile-sharded chips: detailed timing and message traces run on a mapped-only twin of the WHOLE chip
    // (include/sanafe_host.h: sanafe_chip_attach_whole); built on first use from the description load() was given.
    void attach_whole()
    {
        need_chip();
        if (whole_attached || n_ranks <= 1) return;
        int rc = 0;
        {
            py::gil_scoped_release release;
            rc = sanafe_chip_attach_whole(h, reinterpret_cast<const sanafe_desc *>(desc_address_));
        }
        if (rc != 0) raise_last("sanafe_chip_attach_whole failed");
        whole_attached = true;
    }

    std::vector<std::string> perf_columns() const
    {
        const int64_t n = sanafe_chip_perf_columns(h, nullptr, 0);
        std::vector<std::string> out;
        if (n <= 0) return out;
        std::vector<char> buf(static_cast<size_t>(n) * 512);
        sanafe_chip_perf_columns(h, buf.data(), static_cast<int64_t>(buf.size()));
        const char *p = buf.data();
        for (int64_t k = 0; k < n; k++)
        {
            out.emplace_back(p);
            p += out.back().size() + 1;
        }
        return out;
    }

    // ---- pysim, src/pymodule.cpp:549-706 ----
    py::dict sim(long timesteps, const std::string &timing_model_in, int /*processing_threads*/, int scheduler_threads,
            const py::object &spike_arg, const py::object &potential_arg, const py::object &neuron_arg, const py::object &perf_arg,
            const py::object &message_arg, bool write_trace_headers)
    {
        need_chip();
        if (timesteps < 0) throw std::invalid_argument("timesteps must be >= 0");
        int timing = SANAFE_TIMING_DETAILED; // parse_timing_model falls back to detailed, src/chip.cpp:1833-1858
        if (timing_model_in == "simple") timing = SANAFE_TIMING_SIMPLE;
        else if (timing_model_in == "cycle")
        {
            PyErr_SetString(PyExc_NotImplementedError, "UnsupportedError: the cycle-accurate (Booksim2) timing model is out of scope");
            throw py::error_already_set();
        }
        check(sanafe_chip_set_scheduler_threads(h, scheduler_threads));
        Trace spike_t, potential_t, neuron_t, perf_t, message_t;
        spike_t.open(spike_arg, write_trace_headers);
        potential_t.open(potential_arg, write_trace_headers);
        neuron_t.open(neuron_arg, write_trace_headers);
        perf_t.open(perf_arg, write_trace_headers);
        message_t.open(message_arg, write_trace_headers);

        const int64_t start = sanafe_chip_total_timesteps(h) + 1;
        std::vector<int64_t> pot_gids, cur_gids;
        if (potential_t.on())
            for (int64_t g : trace_order)
                if (log_potential[static_cast<size_t>(g)]) pot_gids.push_back(g);
        if (neuron_t.on()) cur_gids = trace_order;
        const bool want_state = !pot_gids.empty() || !cur_gids.empty();
        if (want_state) check(sanafe_chip_set_state_log(h, static_cast<int64_t>(pot_gids.size()), pot_gids.data(), static_cast<int64_t>(cur_gids.size()), cur_gids.data()));
        // (... and for the optional perf columns, which are whole-chip sums over the gathered statuses)
        if (n_ranks > 1 && (timing == SANAFE_TIMING_DETAILED || message_t.on() || (perf_t.on() && sanafe_chip_wants_perf_columns(h) != 0))) attach_whole();
        const bool want_steps = spike_t.on() || perf_t.on() || message_t.on() || want_state;
        const int record = (want_steps ? SANAFE_RECORD_STEPS : 0) | (message_t.on() ? SANAFE_RECORD_MESSAGES : 0) | (want_state ? SANAFE_RECORD_STATE : 0);
        const std::vector<std::string> opt_names = perf_t.on() ? perf_columns() : std::vector<std::string>();

        // ---- headers (src/chip.cpp:1447-1608) ----
        if (write_trace_headers)
        {
            if (spike_t.to_stream()) spike_t.write("neuron,timestep\n");
            if (potential_t.to_stream())
            {
                std::string hdr = "timestep,";
                for (int64_t g : pot_gids) hdr += "neuron " + label_text(g) + ",";
                potential_t.write(hdr + "\n");
            }
            if (neuron_t.to_stream())
            {
                std::string hdr = "timestep,";
                for (int64_t g : cur_gids) hdr += "neuron " + label_text(g) + "/u,";
                neuron_t.write(hdr + "\n");
            }
            if (perf_t.to_stream())
            {
                std::string hdr = "timestep,fired,updated,packets,hops,spikes,sim_time,synapse_energy,dendrite_energy,soma_energy,network_energy,total_energy";
                for (const std::string &n : opt_names) hdr += "," + n;
                perf_t.write(hdr + "\n");
            }
            if (message_t.to_stream())
                message_t.write("timestep,mid,src_neuron,src_hw,dest_hw,hops,spikes,send_timestamp,received_timestamp,processed_timestamp,"
                                "generation_delay,processing_delay,network_delay,blocking_delay,min_hop_delay,messages_along_route\n");
        }

        // ---- in-memory trace objects ----
        py::list spike_mem, potential_mem, message_mem;
        py::list neuron_u_mem;
        std::map<std::string, py::list> perf_mem;
        static const char *perf_keys[] = {"timestep", "fired", "updated", "hops", "spikes", "sim_time", "synapse_energy",
                "dendrite_energy", "soma_energy", "network_energy", "total_energy"}; // timestep_data_to_map, src/pytrace.cpp:55-74
        if (perf_t.mode == Trace::Memory)
        {
            for (const char *k : perf_keys) perf_mem[k] = py::list();
            for (const std::string &n : opt_names) perf_mem[n] = py::list();
        }

        sanafe_hip_totals run{};
        const size_t state_row = pot_gids.size() + cur_gids.size();
        std::vector<sanafe_hip_totals> steps;
        std::vector<uint8_t> fired(static_cast<size_t>(n_neurons));
        std::vector<double> state, optional;
        std::vector<sanafe_message> msgs;
        // chunks: a FIXED schedule (64, 128, 256, ... up to the cap), so the RunData sums -- chunk totals added in chunk
        // order -- do not depend on the wall clock and are reproducible run to run; signals are polled between chunks;
        // recorded runs are also bounded by what a chunk keeps in memory
        long chunk = 64;
        const long chunk_cap = want_steps ? 4096 : 16384;
        struct CloseTraces // file traces are closed on every exit path (an exception, Ctrl-C between chunks)
        {
            std::vector<Trace *> traces;
            ~CloseTraces()
            {
                for (Trace *t : traces)
                {
                    try
                    {
                        t->close();
                    }
                    catch (...)
                    {
                    }
                }
            }
        } close_guard{{&spike_t, &potential_t, &neuron_t, &perf_t, &message_t}};
        for (long done = 0; done < timesteps;)
        {
            const long m = std::min(chunk, timesteps - done);
            sanafe_hip_totals part{};
            int rc = 0;
            {
                py::gil_scoped_release release; // the simulation never touches Python objects
                rc = sanafe_chip_sim(h, m, timing, record, &part);
            }
            if (rc != 0) raise_last("sanafe_chip_sim failed");
            if (chunk < chunk_cap) chunk *= 2;
            run.spikes += part.spikes;
            run.packets_sent += part.packets_sent;
            run.neurons_updated += part.neurons_updated;
            run.neurons_fired += part.neurons_fired;
            run.total_hops += part.total_hops;
            run.total_energy += part.total_energy;
            run.synapse_energy += part.synapse_energy;
            run.dendrite_energy += part.dendrite_energy;
            run.soma_energy += part.soma_energy;
            run.network_energy += part.network_energy;
            run.sim_time += part.sim_time;
            if (want_steps)
            {
                steps.resize(static_cast<size_t>(m));
                check(sanafe_chip_get_step_totals(h, 0, m, steps.data()));
                if (want_state)
                {
                    state.resize(static_cast<size_t>(m) * state_row);
                    check(sanafe_chip_get_step_state(h, 0, m, state.data()));
                }
                if (!opt_names.empty())
                {
                    optional.resize(static_cast<size_t>(m) * opt_names.size());
                    check(sanafe_chip_get_step_optional(h, 0, m, optional.data()));
                }
                for (long s = 0; s < m; s++) record_step(start + done + s, s, steps[s], spike_t, potential_t, neuron_t, perf_t, message_t, fired, pot_gids,
                        cur_gids, state.empty() ? nullptr : state.data() + static_cast<size_t>(s) * state_row, opt_names,
                        optional.empty() ? nullptr : optional.data() + static_cast<size_t>(s) * opt_names.size(), msgs, spike_mem, potential_mem,
                        neuron_u_mem, perf_mem, message_mem);
            }
            done += m;
            if (PyErr_CheckSignals() != 0) throw py::error_already_set(); // Ctrl-C between chunks
        }
        for (Trace *t : {&spike_t, &potential_t, &neuron_t, &perf_t, &message_t}) t->close();

        py::dict result; // src/pymodule.cpp:268-288, 698-703
        result["timestep_start"] = start;
        result["timesteps_executed"] = timesteps;
        py::dict energy;
        energy["total"] = run.total_energy;
        energy["synapse"] = run.synapse_energy;
        energy["dendrite"] = run.dendrite_energy;
        energy["soma"] = run.soma_energy;
        energy["network"] = run.network_energy;
        result["energy"] = energy;
        result["sim_time"] = run.sim_time;
        result["spikes"] = run.spikes;
        result["packets_sent"] = run.packets_sent;
        result["neurons_updated"] = run.neurons_updated;
        result["neurons_fired"] = run.neurons_fired;
        result["spike_trace"] = spike_t.mode == Trace::Memory ? py::object(spike_mem) : py::object(py::none());
        result["potential_trace"] = potential_t.mode == Trace::Memory ? py::object(potential_mem) : py::object(py::none());
        if (neuron_t.mode == Trace::Memory)
        {
            py::dict nt;
            nt["u"] = neuron_u_mem;
            result["neuron_trace"] = nt;
        }
        else
        {
            result["neuron_trace"] = py::none();
        }
        if (perf_t.mode == Trace::Memory)
        {
            py::dict pd;
            for (auto &kv : perf_mem) pd[py::str(kv.first)] = kv.second;
            result["perf_trace"] = pd;
        }
        else
        {
            result["perf_trace"] = py::none();
        }
        result["message_trace"] = message_t.mode == Trace::Memory ? py::object(message_mem) : py::object(py::none());
        return result;
    }

    void record_step(int64_t timestep, long s, const sanafe_hip_totals &ts, Trace &spike_t, Trace &potential_t, Trace &neuron_t, Trace &perf_t,
            Trace &message_t, std::vector<uint8_t> &fired, const std::vector<int64_t> &pot_gids, const std::vector<int64_t> &cur_gids,
            const double *state, const std::vector<std::string> &opt_names, const double *optional, std::vector<sanafe_message> &msgs,
            py::list &spike_mem, py::list &potential_mem, py::list &neuron_u_mem, std::map<std::string, py::list> &perf_mem,
            py::list &message_mem)
    {
        if (spike_t.on()) // sim_trace_record_spikes, src/chip.cpp:1610-1630: fired neurons with log_spikes, in trace order
        {
            check(sanafe_chip_get_step_fired(h, s, fired.data()));
            py::list row;
            std::string text;
            for (int64_t g : trace_order)
                if (fired[static_cast<size_t>(g)] && log_spikes[static_cast<size_t>(g)])
                {
                    if (spike_t.mode == Trace::Memory) row.append(label(g));
                    else text += label_text(g) + "," + std::to_string(timestep) + "\n";
                }
            if (spike_t.mode == Trace::Memory) spike_mem.append(row);
            else spike_t.write(text);
        }
        if (potential_t.on())
        {
            if (potential_t.mode == Trace::Memory)
            {
                py::list row;
                for (size_t k = 0; k < pot_gids.size(); k++) row.append(state[k]);
                potential_mem.append(row);
            }
            else if (!pot_gids.empty()) // default ostream precision: 6 significant digits (src/chip.cpp:1632-1665)
            {
                std::string text = std::to_string(timestep) + ",";
                for (size_t k = 0; k < pot_gids.size(); k++) text += fmt_g(state[k]) + ",";
                potential_t.write(text + "\n");
            }
        }
        if (neuron_t.on())
        {
            const double *u = state + pot_gids.size();
            if (neuron_t.mode == Trace::Memory)
            {
                py::list row;
                for (size_t k = 0; k < cur_gids.size(); k++) row.append(u[k]);
                neuron_u_mem.append(row);
            }
            else
            {
                std::string text = std::to_string(timestep) + ",";
                for (size_t k = 0; k < cur_gids.size(); k++) text += fmt_g(u[k]) + ",";
                neuron_t.write(text + "\n");
            }
        }
        if (perf_t.on())
        {
            if (perf_t.mode == Trace::Memory)
            {
                perf_mem["timestep"].append(timestep);
                perf_mem["fired"].append(ts.neurons_fired);
                perf_mem["updated"].append(ts.neurons_updated);
                perf_mem["hops"].append(ts.total_hops);
                perf_mem["spikes"].append(ts.spikes);
                perf_mem["sim_time"].append(ts.sim_time);
                perf_mem["synapse_energy"].append(ts.synapse_energy);
                perf_mem["dendrite_energy"].append(ts.dendrite_energy);
                perf_mem["soma_energy"].append(ts.soma_energy);
                perf_mem["network_energy"].append(ts.network_energy);
                perf_mem["total_energy"].append(ts.total_energy);
                for (size_t k = 0; k < opt_names.size(); k++) perf_mem[opt_names[k]].append(optional[k]);
            }
            else // sim_trace_record_perf, src/chip.cpp:1704-1729
            {
                std::string text = std::to_string(timestep) + "," + std::to_string(ts.neurons_fired) + "," + std::to_string(ts.neurons_updated) + "," +
                        std::to_string(ts.packets_sent) + "," + std::to_string(ts.total_hops) + "," + std::to_string(ts.spikes) + "," + fmt_e(ts.sim_time) +
                        "," + fmt_e(ts.synapse_energy) + "," + fmt_e(ts.dendrite_energy) + "," + fmt_e(ts.soma_energy) + "," +
                        fmt_e(ts.network_energy) + "," + fmt_e(ts.total_energy);
                for (size_t k = 0; k < opt_names.size(); k++) text += "," + fmt_e(optional[k]);
                perf_t.write(text + "\n");
            }
        }
        if (message_t.on())
        {
            const int64_t n = sanafe_chip_get_step_messages(h, s, nullptr, 0);
            if (n < 0) throw std::runtime_error("messages of the step were not recorded");
            msgs.resize(static_cast<size_t>(n));
            if (n > 0) sanafe_chip_get_step_messages(h, s, msgs.data(), n);
            if (message_t.mode == Trace::Memory)
            {
                // plain mid order: placeholders (-1) first (src/pytrace.hpp:336-339)
                std::stable_sort(msgs.begin(), msgs.end(), [](const sanafe_message &a, const sanafe_message &b) { return a.mid < b.mid; });
                py::list row;
                for (const sanafe_message &m : msgs) row.append(message_dict(m));
                message_mem.append(row);
            }
            else
            {
                // mid order with placeholders LAST (CompareMessagesByID, src/message.cpp:70-91)
                std::stable_sort(msgs.begin(), msgs.end(), [](const sanafe_message &a, const sanafe_message &b) {
                    if ((a.mid < 0) != (b.mid < 0)) return b.mid < 0;
                    return a.mid < b.mid;
                });
                std::string text;
                for (const sanafe_message &m : msgs) // sim_trace_record_message, src/chip.cpp:1731-1764
                {
                    text += std::to_string(m.timestep) + "," + std::to_string(m.mid) + "," + label_text(m.src_neuron) + "," +
                            std::to_string(m.src_tile) + "." + std::to_string(m.src_core_offset) + ",";
                    text += m.placeholder ? std::string("x.x") : std::to_string(m.dest_tile) + "." + std::to_string(m.dest_core_offset);
                    text += "," + std::to_string(m.hops) + "," + std::to_string(m.spikes);
                    for (double v : {m.sent_timestamp, m.received_timestamp, m.processed_timestamp, m.generation_delay, m.processing_delay,
                                 m.network_delay, m.blocking_delay, m.min_hop_delay, m.messages_along_route})
                        text += "," + fmt_g(v);
                    text += "\n";
                }
                message_t.write(text);
            }
        }
    }

    py::dict message_dict(const sanafe_message &m) const // message_to_dict, src/pytrace.cpp:17-53: exactly these 26 keys
    {
        py::dict d;
        d["generation_delay"] = m.generation_delay;
        d["network_delay"] = m.network_delay;
        d["processing_delay"] = m.processing_delay;
        d["blocking_delay"] = m.blocking_delay;
        d["send_timestamp"] = m.sent_timestamp;
        d["received_timestamp"] = m.received_timestamp;
        d["processed_timestamp"] = m.processed_timestamp;
        d["timestep"] = m.timestep;
        d["mid"] = m.mid;
        d["spikes"] = m.spikes;
        d["hops"] = m.hops;
        const bool known = m.src_neuron >= 0 && m.src_neuron < n_neurons;
        const int32_t g = known ? group_of_gid[static_cast<size_t>(m.src_neuron)] : 0;
        d["src_neuron_offset"] = known ? m.src_neuron - group_base[g] : 0;
        d["src_neuron_group_id"] = known ? group_names[g] : std::string("invalid");
        d["src_x"] = m.src_x;
        d["dest_x"] = m.dest_x;
        d["src_y"] = m.src_y;
        d["dest_y"] = m.dest_y;
        d["src_tile_id"] = m.src_tile;
        d["src_core_id"] = m.src_core_id;
        d["src_core_offset"] = m.src_core_offset;
        d["dest_tile_id"] = m.dest_tile;
        d["dest_core_id"] = m.dest_core_id;
        d["dest_core_offset"] = m.dest_core_offset;
        d["dest_axon_hw"] = 0;
        d["dest_axon_id"] = m.dest_axon_id;
        d["placeholder"] = py::bool_(m.placeholder != 0);
        return d;
    }

    // ---- MappedNeuron::set_attributes, src/mapped.cpp:113-166 ----
    void set_attributes(int64_t gid, const py::object &model, const py::object &soma, const py::object &dendrite, const py::object &log_sp)
    {
        need_chip();
        static const char *frozen[] = {"taps", "time_constants", "space_constants"};
        std::map<std::string, py::object> attrs;
        for (const py::object *src : {&model, &dendrite})
            if (!src->is_none())
                for (const auto &kv : src->cast<py::dict>())
                    for (const char *f : frozen)
                        if (kv.first.cast<std::string>() == f)
                        {
                            PyErr_SetString(PyExc_NotImplementedError, "`taps` dendrite attributes cannot change after load() on the MI355X backend");
                            throw py::error_already_set();
                        }
        // every attribute goes to the neuron's soma unit as at load(); the accumulator dendrites have no per-neuron
        // attributes, so dendrite_attributes change nothing there, as in the reference
        for (const py::object *src : {&model, &soma})
            if (!src->is_none())
                for (const auto &kv : src->cast<py::dict>()) attrs[kv.first.cast<std::string>()] = py::reinterpret_borrow<py::object>(kv.second);
        for (const auto &kv : attrs)
        {
            const AttrValue a = py_to_attr(kv.second, true);
            if (a.type == SANAFE_ATTR_LIST) check(sanafe_chip_set_attribute_list(h, gid, kv.first.c_str(), a.list.empty() ? nullptr : a.list.data(), static_cast<int64_t>(a.list.size())));
            else check(sanafe_chip_set_attribute(h, gid, kv.first.c_str(), a.type, a.num, a.type == SANAFE_ATTR_STRING ? a.str.c_str() : nullptr));
        }
        if (!log_sp.is_none()) log_spikes[static_cast<size_t>(gid)] = log_sp.cast<bool>() ? 1 : 0;
    }
};
} // namespace

void bind_spiking_chip(py::module_ &m)
{
    py::class_<MappedNeuronHandle>(m, "MappedNeuron")
            .def(
                    "set_attributes",
                    [](MappedNeuronHandle &n, const py::object &model, const py::object &soma, const py::object &dend, const py::object &ls) {
                        n.chip->set_attributes(n.gid, model, soma, dend, ls);
                    },
                    py::arg("model_attributes") = py::none(), py::arg("soma_attributes") = py::none(), py::arg("dendrite_attributes") = py::none(),
                    py::arg("log_spikes") = py::none())
            .def(
                    "set_model_attributes", // older spelling some scripts use (scripts/computer2026/combined.py:297)
                    [](MappedNeuronHandle &n, const py::object &model, const py::object &soma, const py::object &dend) {
                        n.chip->set_attributes(n.gid, model, soma, dend, py::none());
                    },
                    py::arg("model_attributes") = py::none(), py::arg("soma_attributes") = py::none(), py::arg("dendrite_attributes") = py::none());
    py::class_<GroupView>(m, "MappedNeuronGroup")
            .def("__len__", [](const GroupView &g) { return g.count; })
            .def("__getitem__",
                    [](const GroupView &g, int64_t i) {
                        if (i < 0) i += g.count;
                        if (i < 0 || i >= g.count) throw py::index_error();
                        return MappedNeuronHandle{g.owner, g.chip, g.base + i};
                    })
            .def("__iter__", [](const GroupView &g) {
                py::list out;
                for (int64_t i = 0; i < g.count; i++) out.append(MappedNeuronHandle{g.owner, g.chip, g.base + i});
                return out.attr("__iter__")();
            });
    py::class_<Chip>(m, "SpikingChip")
            .def(py::init<py::object, int, int, int>(), py::arg("arch"), py::arg("device") = 0, py::arg("n_ranks") = 1, py::arg("rank") = 0)
            .def("load", &Chip::load, py::arg("net"), py::arg("overwrite") = false)
            .def(
                    "load_lowered",
                    [](Chip &c, uintptr_t address, const std::vector<std::tuple<std::string, int64_t, int64_t>> &groups,
                            const py::array_t<uint8_t> &ls, const py::array_t<uint8_t> &lp, py::object keep) { c.adopt(address, groups, ls, lp, std::move(keep)); },
                    py::arg("desc_address"), py::arg("groups"), py::arg("log_spikes"), py::arg("log_potential"), py::arg("keepalive"),
                    "Program the chip from an already lowered sanafe_desc (tests: descriptions built by the Python twin).")
            .def("attach_whole", &Chip::attach_whole,
                    "Tile-sharded chips: map the whole chip a second time on the host (no device) -- the tables detailed timing "
                    "and message traces run on; sim() does it on first use")
            .def("sim", &Chip::sim, py::arg("timesteps") = 1, py::arg("timing_model") = "detailed", py::arg("processing_threads") = 0,
                    py::arg("scheduler_threads") = 0, py::arg("spike_trace") = py::none(), py::arg("potential_trace") = py::none(),
                    py::arg("neuron_trace") = py::none(), py::arg("perf_trace") = py::none(), py::arg("message_trace") = py::none(),
                    py::arg("write_trace_headers") = true)
            .def("reset",
                    [](Chip &c) {
                        c.need_chip();
                        check(sanafe_chip_reset(c.h));
                    })
            .def("get_power",
                    [](Chip &c) {
                        c.need_chip();
                        return sanafe_chip_get_power(c.h);
                    })
            .def_property_readonly("mapped_neuron_groups",
                    [](py::object self) {
                        Chip &c = self.cast<Chip &>();
                        py::dict out;
                        for (size_t g = 0; g < c.group_names.size(); g++)
                            out[py::str(c.group_names[g])] = GroupView{self, &c, c.group_base[g], c.group_count[g]};
                        return out;
                    })
            .def_property_readonly("handle", [](const Chip &c) { return reinterpret_cast<uintptr_t>(c.h); })
            .def_property_readonly("total_timesteps", [](const Chip &c) { return c.h ? sanafe_chip_total_timesteps(c.h) : int64_t{0}; })
            .def_readonly("n_neurons", &Chip::n_neurons)
            .def_readonly("arch", &Chip::arch)
            .def("group_table",
                    [](const Chip &c) {
                        py::dict out;
                        for (size_t g = 0; g < c.group_names.size(); g++) out[py::str(c.group_names[g])] = py::make_tuple(c.group_base[g], c.group_count[g]);
                        return out;
                    })
            .def("trace_order", [](const Chip &c) { return py::array_t<int64_t>(static_cast<py::ssize_t>(c.trace_order.size()), c.trace_order.data()); })
            .def("log_flags",
                    [](const Chip &c) {
                        return py::make_tuple(py::array_t<uint8_t>(static_cast<py::ssize_t>(c.log_spikes.size()), c.log_spikes.data()),
                                py::array_t<uint8_t>(static_cast<py::ssize_t>(c.log_potential.size()), c.log_potential.data()));
                    })
            .def("perf_columns", [](const Chip &c) {
                c.need_chip();
                return c.perf_columns();
            });
}
