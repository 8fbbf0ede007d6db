// pymodule.cpp -- PyBind11 module `sanafecpp_amd`: the C++17 description objects of the MI355X
// host exposed with the reference's Python names and signatures (module `sanafecpp`,
// src/pymodule.cpp:850-1213): load_arch, load_net, Architecture, Tile, Core, Network,
// NeuronGroup, Neuron.  Python values are converted with the reference's rules
// (src/pymodule.cpp:118-175: bool/int -> int, float -> float32-narrowed double, str,
// iterable -> list).  `to_desc` lowers to the flat sanafe_desc that SpikingChip.load() hands to
// libsanafe_host.so.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <memory>
#include <string>

#include "description.hpp"
#include "pyconv.hpp"

namespace py = pybind11;
using namespace sanafe_amd;

AttrValue py_to_attr(const py::handle &v, bool narrow)
{
    if (py::isinstance<py::str>(v)) return AttrValue::String(v.cast<std::string>());
    if (py::hasattr(v, "dtype") && py::hasattr(v, "ndim") && v.attr("ndim").cast<int>() == 0)
    {
        const std::string kind = v.attr("dtype").attr("kind").cast<std::string>();
        if (kind == "i" || kind == "u") return AttrValue::Int(v.cast<long>());
        if (kind == "f") return AttrValue::Double(narrow ? static_cast<double>(v.cast<float>()) : v.cast<double>());
        if (kind == "b") return AttrValue::Int(v.cast<bool>() ? 1 : 0);
    }
    if (py::isinstance<py::bool_>(v)) return narrow ? AttrValue::Int(v.cast<bool>() ? 1 : 0) : AttrValue::Bool(v.cast<bool>());
    if (py::isinstance<py::int_>(v)) return AttrValue::Int(v.cast<long>());
    if (py::isinstance<py::float_>(v)) return AttrValue::Double(narrow ? static_cast<double>(v.cast<float>()) : v.cast<double>());
    if (py::isinstance<py::dict>(v)) throw std::invalid_argument("named sub-attribute maps are not supported in neuron/edge attributes");
    if (py::isinstance<py::iterable>(v))
    {
        std::vector<double> l;
        for (const py::handle &e : v)
        {
            const AttrValue a = py_to_attr(e, narrow);
            if (a.type == SANAFE_ATTR_STRING || a.type == SANAFE_ATTR_LIST)
                throw std::invalid_argument("only flat numeric lists are supported as attribute values");
            l.push_back(a.num);
        }
        return AttrValue::List(std::move(l));
    }
    throw std::invalid_argument("Error: dict has unsupported type");
}
namespace
{
std::map<std::string, std::pair<AttrValue, int>> dict_attrs(const py::dict &d, int fwd, bool narrow = true)
{
    std::map<std::string, std::pair<AttrValue, int>> out;
    for (const auto &kv : d) out[kv.first.cast<std::string>()] = {py_to_attr(kv.second, narrow), fwd};
    return out;
}
std::map<std::string, AttrValue> unit_attrs(const py::dict &d)
{
    std::map<std::string, AttrValue> out;
    for (const auto &kv : d) out[kv.first.cast<std::string>()] = py_to_attr(kv.second, false);
    return out;
}
// per-edge attribute lists of the hyper-edge builders: "w"/"weight" and "d"/"delay"
void edge_lists(const py::dict &attributes, std::vector<double> &weight, std::vector<int> &delay)
{
    for (const auto &kv : attributes)
    {
        const std::string k = kv.first.cast<std::string>();
        if (!py::isinstance<py::iterable>(kv.second))
            throw std::invalid_argument("Error: Each attribute must be provided as a 1D list/array of values. Multi-dimensional arrays must be "
                                        "flattened in C-order and storing channels as the last dim");
        if (k == "w" || k == "weight")
        {
            weight.clear();
            for (const py::handle &e : kv.second) weight.push_back(py_to_attr(e, true).num);
        }
        else if (k == "d" || k == "delay")
        {
            delay.clear();
            for (const py::handle &e : kv.second) delay.push_back(static_cast<int>(py_to_attr(e, true).num));
        }
        else if (k == "tap") // include/sanafe_desc.h: the edge's dendrite attribute is a delay, or 64 + a tap index
        {
            delay.clear();
            for (const py::handle &e : kv.second) delay.push_back(64 + static_cast<int>(py_to_attr(e, true).num));
        }
        else
        {
            throw std::invalid_argument("edge attributes other than weight/delay/tap are not supported: " + k);
        }
    }
}

struct NeuronRef // PyNeuronRef: keeps its group (and through it the network) alive
{
    py::object owner;
    NeuronGroup *group;
    int64_t offset;
};
struct DescHandle
{
    std::unique_ptr<BuiltDesc> built;
    py::object arch, net; // keep the sources alive: edges are borrowed from the network
    uintptr_t address() const { return reinterpret_cast<uintptr_t>(&built->desc); }
};
} // namespace

PYBIND11_MODULE(sanafecpp_amd, m)
{
    m.doc() = "MI355X-native SANA-FE host: description objects (C++17)";
    m.attr("buffer_before_dendrite_unit") = static_cast<int>(SANAFE_BUF_BEFORE_DENDRITE);
    m.attr("buffer_inside_dendrite_unit") = static_cast<int>(SANAFE_BUF_INSIDE_DENDRITE);
    m.attr("buffer_before_soma_unit") = static_cast<int>(SANAFE_BUF_BEFORE_SOMA);
    m.attr("buffer_inside_soma_unit") = static_cast<int>(SANAFE_BUF_INSIDE_SOMA);
    m.attr("buffer_before_axon_out_unit") = static_cast<int>(SANAFE_BUF_BEFORE_AXON_OUT);

    py::class_<CoreConfig>(m, "Core")
            .def_readonly("name", &CoreConfig::name)
            .def_readonly("id", &CoreConfig::id)
            .def_readonly("parent_tile_id", &CoreConfig::parent_tile_id)
            .def_readonly("offset_within_tile", &CoreConfig::offset_within_tile)
            .def_readwrite("buffer_position", &CoreConfig::buffer_position)
            .def("create_axon_in", &CoreConfig::create_axon_in, py::arg("name"), py::arg("energy_message_in") = 0.0,
                    py::arg("latency_message_in") = 0.0)
            .def("create_axon_out", &CoreConfig::create_axon_out, py::arg("name"), py::arg("energy_message_out") = 0.0,
                    py::arg("latency_message_out") = 0.0)
            .def(
                    "create_synapse",
                    [](CoreConfig &c, const std::string &name, const std::string &model, const py::dict &attributes, const std::string &plugin,
                            bool le, bool ll, bool ue) { c.create_unit("synapse", name, model, unit_attrs(attributes), plugin, le, ll, ue); },
                    py::arg("name"), py::arg("model") = "current_based", py::arg("attributes") = py::dict(), py::arg("plugin") = "",
                    py::arg("log_energy") = false, py::arg("log_latency") = false, py::arg("update_every_timestep") = false)
            .def(
                    "create_dendrite",
                    [](CoreConfig &c, const std::string &name, const std::string &model, const py::dict &attributes, const std::string &plugin,
                            bool le, bool ll, bool ue) { c.create_unit("dendrite", name, model, unit_attrs(attributes), plugin, le, ll, ue); },
                    py::arg("name"), py::arg("model") = "accumulator", py::arg("attributes") = py::dict(), py::arg("plugin") = "",
                    py::arg("log_energy") = false, py::arg("log_latency") = false, py::arg("update_every_timestep") = false)
            .def(
                    "create_soma",
                    [](CoreConfig &c, const std::string &name, const std::string &model, const py::dict &attributes, const std::string &plugin,
                            bool le, bool ll, bool ue) { c.create_unit("soma", name, model, unit_attrs(attributes), plugin, le, ll, ue); },
                    py::arg("name"), py::arg("model") = "leaky_integrate_fire", py::arg("attributes") = py::dict(), py::arg("plugin") = "",
                    py::arg("log_energy") = false, py::arg("log_latency") = false, py::arg("update_every_timestep") = false);

    py::class_<TileConfig>(m, "Tile").def_readonly("name", &TileConfig::name).def_readonly("id", &TileConfig::id);

    py::class_<Architecture>(m, "Architecture")
            .def(py::init([](const std::string &name, int width, int height, int link_buffer_size, const py::object &sync) {
                std::map<int64_t, double> table{{0, 0.0}};
                if (!sync.is_none()) table = sync.cast<std::map<int64_t, double>>();
                return std::make_unique<Architecture>(name, width, height, link_buffer_size, table);
            }),
                    py::arg("name") = "", py::arg("width") = 1, py::arg("height") = 1, py::arg("link_buffer_size") = 0,
                    py::arg("sync_table") = py::none())
            .def_readonly("name", &Architecture::name)
            .def_readonly("noc_width", &Architecture::noc_width)
            .def_readonly("noc_height", &Architecture::noc_height)
            .def_readonly("noc_buffer_size", &Architecture::noc_buffer_size)
            .def_property_readonly("core_count", &Architecture::core_count)
            .def(
                    "create_tile",
                    [](Architecture &a, const std::string &name, double en, double ln, double ee, double le, double es, double ls, double ew,
                            double lw, bool log) -> TileConfig & { return a.create_tile(name, {en, ee, es, ew}, {ln, le, ls, lw}, log); },
                    py::return_value_policy::reference_internal, py::arg("name"), py::arg("energy_north_hop") = 0.0,
                    py::arg("latency_north_hop") = 0.0, py::arg("energy_east_hop") = 0.0, py::arg("latency_east_hop") = 0.0,
                    py::arg("energy_south_hop") = 0.0, py::arg("latency_south_hop") = 0.0, py::arg("energy_west_hop") = 0.0,
                    py::arg("latency_west_hop") = 0.0, py::arg("log_energy") = false)
            .def(
                    "create_core",
                    [](Architecture &a, const std::string &name, int parent_tile_id, const py::object &buffer_position, bool inside,
                            int64_t max_neurons, bool log, const py::object &share) -> CoreConfig & {
                        int bp = SANAFE_BUF_BEFORE_SOMA;
                        if (py::isinstance<py::str>(buffer_position)) bp = parse_buffer_position(buffer_position.cast<std::string>(), inside);
                        else bp = buffer_position.cast<int>();
                        // extension: replicated cores may share one unit description (every core still gets its own instances)
                        std::shared_ptr<CoreTemplate> tmpl = share.is_none() ? nullptr : share.cast<CoreConfig &>().tmpl;
                        return a.create_core(name, parent_tile_id, bp, max_neurons, log, tmpl);
                    },
                    py::return_value_policy::reference_internal, py::arg("name"), py::arg("parent_tile_id"),
                    py::arg("buffer_position") = static_cast<int>(SANAFE_BUF_BEFORE_SOMA), py::arg("buffer_inside_unit") = false,
                    py::arg("max_neurons_supported") = 1024, py::arg("log_energy") = false, py::arg("share_units_with") = py::none())
            .def_property_readonly("tiles",
                    [](py::object self) {
                        Architecture &a = self.cast<Architecture &>();
                        py::list out;
                        for (TileConfig &t : a.tiles) out.append(py::cast(&t, py::return_value_policy::reference_internal, self));
                        return out;
                    })
            .def(
                    "cores",
                    [](py::object self) {
                        Architecture &a = self.cast<Architecture &>();
                        py::list out;
                        for (CoreConfig &c : a.cores) out.append(py::cast(&c, py::return_value_policy::reference_internal, self));
                        return out;
                    })
            .def(
                    "tile_cores",
                    [](py::object self, int tile) {
                        Architecture &a = self.cast<Architecture &>();
                        py::list out;
                        for (int c : a.tiles.at(tile).cores) out.append(py::cast(&a.cores[c], py::return_value_policy::reference_internal, self));
                        return out;
                    });

    py::class_<NeuronRef>(m, "Neuron")
            .def("get_id", [](const NeuronRef &r) { return r.offset; })
            .def("map_to_core", [](const NeuronRef &r, const CoreConfig &c) { r.group->map_to_core(c, r.offset, r.offset + 1); })
            .def(
                    "set_attributes",
                    [](const NeuronRef &r, const py::object &soma, const py::object &syn, const py::object &dend, const py::object &ls,
                            const py::object &lp, const py::dict &model, const py::dict &soma_attrs, const py::dict &dend_attrs) {
                        auto attrs = dict_attrs(model, 7);
                        for (auto &kv : dict_attrs(dend_attrs, SANAFE_FWD_DENDRITE)) attrs[kv.first] = kv.second;
                        for (auto &kv : dict_attrs(soma_attrs, SANAFE_FWD_SOMA)) attrs[kv.first] = kv.second;
                        auto os = [](const py::object &o) { return o.is_none() ? std::nullopt : std::optional<std::string>(o.cast<std::string>()); };
                        auto ob = [](const py::object &o) { return o.is_none() ? std::nullopt : std::optional<bool>(o.cast<bool>()); };
                        r.group->apply_config(r.offset, r.offset + 1, os(soma), os(syn), os(dend), ob(ls), ob(lp), attrs);
                    },
                    py::arg("soma_hw_name") = py::none(), py::arg("default_synapse_hw_name") = py::none(),
                    py::arg("dendrite_hw_name") = py::none(), py::arg("log_spikes") = py::none(), py::arg("log_potential") = py::none(),
                    py::arg("model_attributes") = py::dict(), py::arg("soma_attributes") = py::dict(),
                    py::arg("dendrite_attributes") = py::dict())
            .def(
                    "connect_to_neuron",
                    [](const NeuronRef &r, const NeuronRef &dest, const py::object &attr) {
                        double w = 0.0;
                        int delay = -1;
                        if (!attr.is_none())
                            for (const auto &kv : attr.cast<py::dict>())
                            {
                                const std::string k = kv.first.cast<std::string>();
                                if (k == "w" || k == "weight") w = py_to_attr(kv.second, true).num;
                                else if (k == "d" || k == "delay") delay = static_cast<int>(py_to_attr(kv.second, true).num);
                                else if (k == "tap") delay = 64 + static_cast<int>(py_to_attr(kv.second, true).num);
                                else throw std::invalid_argument("edge attributes other than weight/delay/tap are not supported: " + k);
                            }
                        SpikingNetwork &net = *r.group->net;
                        net.add_edge(r.group->base + r.offset, dest.group->base + dest.offset, w, delay, dest.group->synapse_hw[dest.offset]);
                        return net.edge_count() - 1;
                    },
                    py::arg("dest"), py::arg("attributes") = py::none());

    py::class_<NeuronGroup>(m, "NeuronGroup")
            .def("get_name", [](const NeuronGroup &g) { return g.name; })
            .def_readonly("name", &NeuronGroup::name)
            .def_readonly("base", &NeuronGroup::base)
            .def("__len__", [](const NeuronGroup &g) { return g.count; })
            .def(
                    "__getitem__",
                    [](py::object self, const py::object &idx) -> py::object {
                        NeuronGroup &g = self.cast<NeuronGroup &>();
                        if (py::isinstance<py::slice>(idx))
                        {
                            size_t start = 0, stop = 0, step = 0, len = 0;
                            if (!idx.cast<py::slice>().compute(g.count, &start, &stop, &step, &len)) throw py::error_already_set();
                            py::list out;
                            for (size_t i = 0; i < len; i++) out.append(NeuronRef{self, &g, static_cast<int64_t>(start + i * step)});
                            return out;
                        }
                        const int64_t i = idx.cast<int64_t>();
                        if (i < 0 || i >= g.count) throw py::index_error();
                        return py::cast(NeuronRef{self, &g, i});
                    })
            .def("__iter__",
                    [](py::object self) {
                        NeuronGroup &g = self.cast<NeuronGroup &>();
                        py::list out;
                        for (int64_t i = 0; i < g.count; i++) out.append(NeuronRef{self, &g, i});
                        return out.attr("__iter__")();
                    })
            .def(
                    "connect_neurons_dense",
                    [](NeuronGroup &g, NeuronGroup &dest, const py::dict &attributes) {
                        std::vector<double> w;
                        std::vector<int> d;
                        edge_lists(attributes, w, d);
                        g.connect_neurons_dense(dest, w, d);
                    },
                    py::arg("dest_group"), py::arg("attributes"))
            .def(
                    "connect_neurons_sparse",
                    [](NeuronGroup &g, NeuronGroup &dest, const py::dict &attributes, const py::iterable &pairs) {
                        std::vector<double> w;
                        std::vector<int> d;
                        edge_lists(attributes, w, d);
                        std::vector<std::pair<int64_t, int64_t>> p;
                        for (const py::handle &e : pairs)
                        {
                            const py::sequence s = e.cast<py::sequence>();
                            p.emplace_back(s[0].cast<int64_t>(), s[1].cast<int64_t>());
                        }
                        g.connect_neurons_sparse(dest, p, w, d);
                    },
                    py::arg("dest_group"), py::arg("attributes"), py::arg("src_dest_id_pairs"))
            .def(
                    "connect_neurons_conv2d",
                    [](NeuronGroup &g, NeuronGroup &dest, const py::dict &attributes, int iw, int ih, int ic, int kw, int kh, int kc, int sw,
                            int sh) {
                        std::vector<double> w;
                        std::vector<int> d;
                        edge_lists(attributes, w, d);
                        g.connect_neurons_conv2d(dest, w, d, iw, ih, ic, kw, kh, kc, sw, sh);
                    },
                    py::arg("dest_group"), py::arg("attributes"), py::arg("input_width"), py::arg("input_height"), py::arg("input_channels"),
                    py::arg("kernel_width"), py::arg("kernel_height"), py::arg("kernel_count") = 1, py::arg("stride_width") = 1,
                    py::arg("stride_height") = 1)
            // bulk extensions (no per-neuron Python objects)
            .def(
                    "map_to_core", [](NeuronGroup &g, const CoreConfig &c, int64_t lo, const py::object &hi) {
                        g.map_to_core(c, lo, hi.is_none() ? g.count : hi.cast<int64_t>());
                    },
                    py::arg("core"), py::arg("lo") = 0, py::arg("hi") = py::none())
            .def(
                    "set_attribute_column",
                    [](NeuronGroup &g, const std::string &key, py::array_t<double, py::array::c_style | py::array::forcecast> values,
                            bool integer) {
                        if (values.size() != g.count) throw std::invalid_argument("one value per neuron expected");
                        g.set_attribute_column(key, values.data(), integer ? SANAFE_ATTR_INT : SANAFE_ATTR_DOUBLE);
                    },
                    py::arg("key"), py::arg("values"), py::arg("integer") = false);

    py::class_<SpikingNetwork>(m, "Network")
            .def(py::init<std::string>(), py::arg("name") = "")
            .def_readonly("name", &SpikingNetwork::name)
            .def_readonly("neuron_count", &SpikingNetwork::neuron_count)
            .def("absorb", &SpikingNetwork::absorb, py::arg("other"),
                    "Append a copy of every group, mapping and edge of `other` (SpikingChip.load(net, overwrite=False)).")
            .def_property_readonly("edge_count", &SpikingNetwork::edge_count)
            .def(
                    "create_neuron_group",
                    [](SpikingNetwork &n, const py::object &name, int64_t count, const py::dict &model_attributes, const std::string &syn,
                            const std::string &dend, bool lp, bool ls, const std::string &soma) -> NeuronGroup & {
                        return n.create_neuron_group(py::str(name).cast<std::string>(), count, dict_attrs(model_attributes, 7), syn, dend, lp, ls,
                                soma);
                    },
                    py::return_value_policy::reference_internal, py::arg("group_name"), py::arg("neuron_count"),
                    py::arg("model_attributes") = py::dict(), py::arg("default_synapse_hw_name") = "",
                    py::arg("default_dendrite_hw_name") = "", py::arg("log_potential") = false, py::arg("log_spikes") = false,
                    py::arg("soma_hw_name") = "")
            .def_property_readonly(
                    "groups",
                    [](py::object self) {
                        SpikingNetwork &n = self.cast<SpikingNetwork &>();
                        py::dict out;
                        for (const auto &kv : n.groups) out[py::str(kv.first)] = py::cast(kv.second, py::return_value_policy::reference_internal, self);
                        return out;
                    })
            .def(
                    "__getitem__",
                    [](SpikingNetwork &n, const py::object &name) -> NeuronGroup & {
                        auto it = n.groups.find(py::str(name).cast<std::string>());
                        if (it == n.groups.end()) throw py::index_error();
                        return *it->second;
                    },
                    py::return_value_policy::reference_internal)
            .def(
                    "add_edges",
                    [](SpikingNetwork &n, py::array_t<int64_t, py::array::c_style | py::array::forcecast> src,
                            py::array_t<int64_t, py::array::c_style | py::array::forcecast> dst,
                            py::array_t<double, py::array::c_style | py::array::forcecast> weight, const std::string &synapse_hw_name) {
                        if (src.size() != dst.size() || src.size() != weight.size()) throw std::invalid_argument("edge arrays differ in length");
                        n.add_edges(src.data(), dst.data(), weight.data(), nullptr, n.intern(synapse_hw_name), src.size());
                    },
                    py::arg("src"), py::arg("dst"), py::arg("weight"), py::arg("synapse_hw_name") = "",
                    "bulk append of edges by global neuron id (group.base + offset)")
            .def("group_table",
                    [](const SpikingNetwork &n) {
                        py::list out;
                        for (const auto &g : n.order) out.append(py::make_tuple(g->name, g->base, g->count));
                        return out;
                    })
            .def("log_flags", [](const SpikingNetwork &n) {
                py::array_t<uint8_t> ls(n.neuron_count), lp(n.neuron_count);
                for (const auto &g : n.order)
                {
                    std::copy(g->log_spikes.begin(), g->log_spikes.end(), ls.mutable_data() + g->base);
                    std::copy(g->log_potential.begin(), g->log_potential.end(), lp.mutable_data() + g->base);
                }
                return py::make_tuple(ls, lp);
            });

    py::class_<DescHandle>(m, "Desc")
            .def_property_readonly("address", &DescHandle::address)
            .def_property_readonly("n_neurons", [](const DescHandle &h) { return h.built->desc.n_neurons; })
            .def_property_readonly("n_edges", [](const DescHandle &h) { return h.built->desc.n_edges; });

    m.def(
            "to_desc",
            [](py::object arch, py::object net) {
                auto h = std::make_unique<DescHandle>();
                h->built = to_desc(arch.cast<Architecture &>(), net.cast<SpikingNetwork &>());
                h->arch = arch;
                h->net = net;
                return h;
            },
            py::arg("arch"), py::arg("net"));
    bind_spiking_chip(m); // SpikingChip, MappedNeuron (pychip.cpp)
    m.def("load_arch", &load_arch, py::arg("path"));
    m.def(
            "load_net",
            [](const std::string &path, Architecture &arch, bool use_netlist_format) {
                if (use_netlist_format) throw std::invalid_argument("legacy netlist (.net) format is out of scope");
                return load_net(path, arch);
            },
            py::arg("path"), py::arg("arch"), py::arg("use_netlist_format") = false);
}
