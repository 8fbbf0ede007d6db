"""YAML front-end: ``load_arch(path)`` and ``load_net(path, arch)``.

Follows the semantics of the reference's RapidYAML parsers
(src/yaml_arch.cpp:188-555, src/yaml_snn.cpp:116-1056,
src/yaml_common.cpp:101-319) on top of PyYAML:

* scalar typing order int -> double -> bool -> string
  (src/yaml_common.cpp:205-262);
* ``name[a..b]`` ranges; tile and core names always get an ``[i]`` suffix,
  unit names only when a range is given (src/yaml_arch.cpp:250-263, 308-314,
  392-395);
* unit sections are visited axon_in, synapse, dendrite, soma, axon_out
  regardless of file order (src/yaml_arch.cpp:260-266);
* neuron attribute maps or lists of maps; ``soma:`` / ``dendrite:`` sub-maps
  restrict forwarding (src/yaml_snn.cpp:331-394);
* edges ``a.i -> b.j`` and hyper-edges ``a -> b`` with ``type`` dense /
  sparse / conv2d (src/yaml_snn.cpp:396-829);
* mappings ``g.i`` / ``g.i..j`` / ``g`` with ``core: t.c`` and optional
  ``soma`` / ``dendrite`` / ``synapse`` unit names (src/yaml_snn.cpp:880-1056).
"""
from __future__ import annotations

import re
import yaml

from . import description as D

SKIP_KEYS = {"soma_hw_name", "default_synapse_hw_name", "dendrite_hw_name", "log_spikes", "log_potential",
             "synapse", "dendrite", "soma"}  # src/yaml_common.cpp:30-36

_INT = re.compile(r"^[+-]?\d+$")


class _Loader(yaml.SafeLoader):
    """SafeLoader that keeps every scalar as its source text (typing is ours)."""


def _str_scalar(loader, node):
    return loader.construct_scalar(node)


for _tag in ("bool", "int", "float", "null", "timestamp"):
    _Loader.add_constructor("tag:yaml.org,2002:" + _tag, _str_scalar)


def _scalar(text):
    """int -> double -> bool -> string (src/yaml_common.cpp:205-262)."""
    t = text.strip()
    if _INT.match(t):
        v = int(t)
        if -2 ** 31 <= v < 2 ** 31:
            return (D.ATTR_INT, float(v), None, None)
    try:
        return (D.ATTR_DOUBLE, float(t), None, None)
    except ValueError:
        pass
    if t in ("true", "True", "TRUE"):
        return (D.ATTR_BOOL, 1.0, None, None)
    if t in ("false", "False", "FALSE"):
        return (D.ATTR_BOOL, 0.0, None, None)
    return (D.ATTR_STRING, 0.0, text, None)


def _attr(node):
    if isinstance(node, list):
        vals = []
        for v in node:
            a = _attr(v)
            if a[0] in (D.ATTR_STRING, D.ATTR_LIST, "nested"):
                return ("nested", 0.0, None, node)
            vals.append(a[1])
        return (D.ATTR_LIST, 0.0, None, vals)
    if isinstance(node, dict):
        return ("nested", 0.0, None, node)
    return _scalar(str(node))


def _model_attributes(node):
    """description_parse_model_attributes_yaml: first definition of a key wins across list entries."""
    out = {}
    if isinstance(node, list):
        for entry in node:
            for k, v in _model_attributes(entry).items():
                out.setdefault(k, v)
    elif isinstance(node, dict):
        for k, v in node.items():
            k = str(k)
            if k not in SKIP_KEYS:
                out[k] = v
    elif node is None:
        pass
    else:
        raise ValueError("Error: Model attributes must be an ordered map or mapping of named attributes.\n")
    return out


def _as_bool(v):
    return str(v).strip() in ("true", "True", "TRUE", "1")


def _parse_range(s):
    m = re.search(r"(\d+)\.\.(\d+)", s)
    if not m:
        raise ValueError("Invalid range format")
    a, b = int(m.group(1)), int(m.group(2))
    if a > b:
        raise ValueError("Invalid range; first > last")
    return a, b


def _seq(node):
    return node if isinstance(node, list) else [node]


# --------------------------------------------------------------------------
def load_arch(path) -> D.Architecture:
    with open(path) as f:
        top = yaml.load(f, Loader=_Loader)
    if "architecture" not in top:
        raise ValueError("No architecture section defined")
    a = top["architecture"]
    name = str(a.get("name", ""))
    if "[" in name:
        raise ValueError("Multiple architectures not supported")
    att = a["attributes"]
    model_type = str(att.get("sync_model", "fixed"))
    table = {}
    if model_type == "fixed":
        table[0] = float(att.get("latency_sync", 0.0))
    elif model_type == "table":
        ls = att.get("latency_sync")
        if ls is None:
            raise ValueError("Attribute 'latency_sync' required when 'table' synchronization model is chosen.")
        if isinstance(ls, list):
            table = {i: float(v) for i, v in enumerate(ls)}
        elif isinstance(ls, dict):
            table = {int(k): float(v) for k, v in ls.items()}
        else:
            table[0] = float(ls)
    else:
        raise ValueError("Unknown sync_model: " + model_type)
    arch = D.Architecture(name, int(att["width"]), int(att["height"]), int(att["link_buffer_size"]), table)
    if "tile" not in a:
        raise ValueError("No tile section defined")
    templates = {}  # one CoreTemplate per description entry, shared by its replicas
    for tile_node in _seq(a["tile"]):
        tname = str(tile_node["name"])
        lo, hi = _parse_range(tname) if ".." in tname else (0, 0)
        ta = tile_node["attributes"]
        for t in range(lo, hi + 1):
            metrics = {k: float(ta[k]) for k in ("energy_north_hop", "latency_north_hop", "energy_east_hop",
                                                  "latency_east_hop", "energy_south_hop", "latency_south_hop",
                                                  "energy_west_hop", "latency_west_hop")}
            tile = arch.create_tile("%s[%d]" % (tname.split("[")[0], t), log_energy=_as_bool(ta.get("log_energy", "false")),
                                    **metrics)
            if "core" not in tile_node:
                raise ValueError("No core section defined")
            for core_node in _seq(tile_node["core"]):
                cname = str(core_node["name"])
                clo, chi = _parse_range(cname) if ".." in cname else (0, 0)
                for c in range(clo, chi + 1):
                    _parse_core(arch, tile.id, core_node, "%s[%d]" % (cname.split("[")[0], c), templates)
    return arch


def _parse_core(arch, tile_id, node, name, templates):
    ca = node["attributes"]
    shared = templates.get(id(node))
    core = arch.create_core(name, tile_id, str(ca["buffer_position"]), _as_bool(ca.get("buffer_inside_unit", "false")),
                            int(ca["max_neurons_supported"]), _as_bool(ca.get("log_energy", "false")), shared)
    if shared is not None:
        return
    templates[id(node)] = core.template
    for section in ("axon_in", "synapse", "dendrite", "soma", "axon_out"):
        if section not in node or node[section] is None:
            raise ValueError("No %s section defined" % section)
        for unit_node in _seq(node[section]):
            uname = str(unit_node["name"])
            lo, hi = _parse_range(uname) if ".." in uname else (0, 0)
            for i in range(lo, hi + 1):
                n = "%s[%d]" % (uname.split("[")[0], i) if ".." in uname else uname
                ua = unit_node.get("attributes") or {}
                if section == "axon_in":
                    core.create_axon_in(n, float(ua["energy_message_in"]), float(ua["latency_message_in"]))
                elif section == "axon_out":
                    core.create_axon_out(n, float(ua["energy_message_out"]), float(ua["latency_message_out"]))
                else:
                    attrs = {}
                    for k, v in _model_attributes(ua).items():
                        t = _attr(v)
                        if t[0] == "nested":
                            continue
                        attrs[k] = t
                    model = str(ua["model"])
                    plugin = str(ua["plugin"]) if "plugin" in ua else None
                    flags = dict(log_energy=_as_bool(ua.get("log_energy", "false")),
                                 log_latency=_as_bool(ua.get("log_latency", "false")),
                                 update_every_timestep=_as_bool(ua.get("update_every_timestep", "false")))
                    getattr(core, "create_" + section)(n, model, attrs, plugin, **flags)


# --------------------------------------------------------------------------
class _NeuronConfig:
    def __init__(self, other=None):
        self.soma = other.soma if other else None
        self.synapse = other.synapse if other else None
        self.dendrite = other.dendrite if other else None
        self.log_spikes = other.log_spikes if other else None
        self.log_potential = other.log_potential if other else None
        self.attrs = dict(other.attrs) if other else {}


def _neuron_attributes(node, template=None):
    """yaml_parse_neuron_attributes (src/yaml_snn.cpp:331-394)."""
    cfg = _NeuronConfig(template)
    if isinstance(node, list):
        for entry in node:
            cfg = _neuron_attributes(entry, cfg)
        return cfg
    if node is None:
        return cfg
    if "log_potential" in node:
        cfg.log_potential = _as_bool(node["log_potential"])
    if "log_spikes" in node:
        cfg.log_spikes = _as_bool(node["log_spikes"])
    if "synapse_hw_name" in node:
        cfg.synapse = str(node["synapse_hw_name"])
    if "dendrite_hw_name" in node:
        cfg.dendrite = str(node["dendrite_hw_name"])
    if "soma_hw_name" in node:
        cfg.soma = str(node["soma_hw_name"])
    for k, v in _model_attributes(node).items():
        cfg.attrs[k] = (_typed(v), D.FWD_ALL)
    if isinstance(node.get("dendrite"), (dict, list)):
        for k, v in _model_attributes(node["dendrite"]).items():
            cfg.attrs[k] = (_typed(v), D.FWD_DENDRITE)
    if isinstance(node.get("soma"), (dict, list)):
        for k, v in _model_attributes(node["soma"]).items():
            cfg.attrs[k] = (_typed(v), D.FWD_SOMA)
    return cfg


def _typed(v):
    t = _attr(v)
    if t[0] == "nested":
        raise NotImplementedError("nested attribute values are not supported")
    return t


def _count_neurons(neurons_node):
    n = 0
    for entry in neurons_node:
        if isinstance(entry, (dict, list)):
            items = entry.items() if isinstance(entry, dict) else [(k, None) for e in entry for k in e]
            for k, _ in items:
                k = str(k)
                if ".." in k:
                    a, b = _parse_range(k)
                    n += b - a + 1
                else:
                    n += 1
        else:
            k = str(entry)
            if ".." in k:
                a, b = _parse_range(k)
                n += b - a + 1
            else:
                n += 1
    return n


def load_net(path, arch: D.Architecture, use_netlist_format=False) -> D.Network:
    if use_netlist_format:
        raise NotImplementedError("legacy netlist (.net) format is out of scope")
    with open(path) as f:
        top = yaml.load(f, Loader=_Loader)
    if "network" not in top:
        raise ValueError("No network section defined")
    nn = top["network"]
    net = D.Network(str(nn.get("name", "")))
    if "groups" not in nn:
        raise ValueError("No neuron groups specified")
    if "edges" not in nn:
        raise ValueError("No edges section specified")
    if not isinstance(nn["groups"], list):
        raise ValueError("Neuron group section does not define a list of groups")
    for g in nn["groups"]:
        gname = str(g["name"])
        if "neurons" not in g:
            raise ValueError("No neurons section defined.")
        count = _count_neurons(g["neurons"])
        default = _neuron_attributes(g.get("attributes"))
        group = net.create_neuron_group(gname, count, None, default.synapse or "", default.dendrite or "",
                                        bool(default.log_potential), bool(default.log_spikes), default.soma or "",
                                        _typed_attrs=default.attrs)
        for entry in g["neurons"]:
            if not isinstance(entry, dict):
                continue
            for k, v in entry.items():
                k = str(k)
                cfg = _neuron_attributes(v, default)
                lo, hi = _parse_range(k) if ".." in k else (int(k), int(k))
                group.apply_config(lo, hi + 1, cfg.soma, cfg.synapse, cfg.dendrite, cfg.log_spikes, cfg.log_potential,
                                   cfg.attrs)
    for entry in (nn["edges"] or []):
        for desc, attrs in entry.items():
            _parse_edge(net, str(desc), attrs)
    if "mappings" in top and top["mappings"] is not None:
        for m in top["mappings"]:
            if not isinstance(m, dict) or len(m) != 1:
                raise ValueError("Should be one entry per mapping")
            for k, v in m.items():
                _parse_mapping(net, arch, str(k), v)
    return net


def _edge_attr_dict(attrs):
    """description_parse_edge_attributes (src/yaml_snn.cpp:831-878): weight / delay only."""
    out = {}
    for k, v in _model_attributes(attrs).items():
        out[k] = v
    for sect in ("synapse", "dendrite"):
        sub = None
        if isinstance(attrs, dict):
            sub = attrs.get(sect)
        elif isinstance(attrs, list):
            for e in attrs:
                if isinstance(e, dict) and sect in e:
                    sub = e[sect]
        if sub is not None:
            for k, v in _model_attributes(sub).items():
                out[k] = v
    return out


def _parse_edge(net, desc, attrs):
    if "->" not in desc:
        raise ValueError("Edge is not formatted correctly: " + desc)
    sp, tp = (x.strip() for x in desc.split("->", 1))
    s_def, t_def = "." in sp, "." in tp
    if s_def != t_def:
        raise ValueError("No target neuron defined in edge:" + desc)
    sg, tg = sp.split(".")[0], tp.split(".")[0]
    for gname, what in ((sg, "source"), (tg, "target")):
        if gname not in net.groups:
            raise ValueError("Invalid %s neuron group:%s" % (what, gname))
    src, dst = net.groups[sg], net.groups[tg]
    ea = _edge_attr_dict(attrs)
    if s_def:
        so, to = int(sp.split(".", 1)[1]), int(tp.split(".", 1)[1])
        if so >= src.count:
            raise ValueError("Invalid source neuron id: %s.%d" % (sg, so))
        if to >= dst.count:
            raise ValueError("Invalid target neuron id: %s.%d" % (tg, to))
        conv = {}
        for k, v in ea.items():
            conv[k] = _typed(v)[1]
        w = conv.get("w", conv.get("weight", 0.0))
        d = conv.get("delay", conv.get("d"))
        if "tap" in conv:
            if d is not None:
                raise NotImplementedError("an edge with both `delay` and `tap`")
            d = 64 + int(conv["tap"])
        import numpy as np
        net._add_edges(np.array([src.base + so]), np.array([dst.base + to]), np.array([float(w)]),
                       None if d is None else np.array([int(d)]), np.array([dst.synapse_hw[to]]))
        return
    etype = str(ea.pop("type", ""))
    if not etype:
        raise ValueError("No hyperedge type specified.")
    lists = {}
    params = {}
    pairs = None
    for k, v in ea.items():
        if etype == "conv2d" and k in ("input_height", "input_width", "input_channels", "kernel_width", "kernel_height",
                                       "kernel_count", "stride_width", "stride_height"):
            params[k] = int(str(v))
        elif etype == "sparse" and k == "source_target_pairs":
            pairs = [(int(str(p[0])), int(str(p[1]))) for p in v]
        else:
            if not isinstance(v, list):
                raise ValueError("Attribute must be a list with an entry for each connection (name: %s)" % k)
            lists[k] = [_scalar(str(x))[1] for x in v]
    import numpy as np
    lists = {k: np.asarray(v, dtype=np.float64) for k, v in lists.items()}
    if etype == "conv2d":
        src.connect_neurons_conv2d(dst, lists, narrow_float=False, **params)
    elif etype == "dense":
        src.connect_neurons_dense(dst, lists, narrow_float=False)
    elif etype == "sparse":
        src.connect_neurons_sparse(dst, lists, pairs or [], narrow_float=False)
    else:
        raise ValueError("Invalid hyperedge type: " + etype)


def _parse_mapping(net, arch, address, info):
    """description_parse_mapping (src/yaml_snn.cpp:923-1056)."""
    gname = address.split(".")[0]
    if gname not in net.groups:
        raise ValueError("While mapping, group not found (%s)" % gname)
    group = net.groups[gname]
    if "." in address:
        ns = address.split(".", 1)[1]
        lo, hi = _parse_range(ns) if ".." in ns else (int(ns), int(ns))
    else:
        lo, hi = 0, group.count - 1
    if hi >= group.count:
        raise ValueError("Invalid neuron id: %s.%d" % (gname, hi))
    fields = {}
    for entry in _seq(info):
        if not isinstance(entry, dict):
            raise ValueError("Expected attributes to be map")
        fields.update({str(k): str(v) for k, v in entry.items()})
    s = net.strings
    if "synapse" in fields:
        group.synapse_hw[lo:hi + 1] = s(fields["synapse"])
    if "dendrite" in fields:
        group.dendrite_hw[lo:hi + 1] = s(fields["dendrite"])
    if "soma" in fields:
        group.soma_hw[lo:hi + 1] = s(fields["soma"])
    core_addr = fields.get("core", "")
    t, c = core_addr.split(".")
    t, c = int(t), int(c)
    if t >= len(arch.tiles):
        raise ValueError("Tile ID >= tile count")
    if c >= len(arch.tiles[t].cores):
        raise ValueError("Core ID >= core count")
    group.map_to_core(arch.tiles[t].cores[c], lo, hi + 1)
