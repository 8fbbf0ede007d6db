"""Architecture / SNN description objects and the flat ``sanafe_desc`` they lower to.

Python mirror of the reference's description API so that scripts written for
``sanafe`` read the same here:

* ``Architecture.create_tile / create_core`` and the ``Core.create_*`` unit
  builders follow src/arch.cpp:41-180 and src/pymodule.cpp (``pycreate_tile``,
  ``pycreate_core``);
* ``Network.create_neuron_group``, ``NeuronGroup.connect_neurons_dense /
  _sparse / _conv2d`` and ``Neuron.map_to_core / set_attributes /
  connect_to_neuron`` follow src/network.cpp:62-605 and
  src/pymodule.cpp:289-470;
* Python values become attributes with the reference's typing rules
  (src/pymodule.cpp:118-175: bool->int, int->int, float->float32-narrowed
  double, str, iterable->list).

Everything is stored columnar (numpy) so a million-neuron / hundred-million
synapse network never materialises per-object Python state; ``to_desc()``
emits the ``sanafe_desc`` struct of include/sanafe_desc.h that both the HIP
host mapper and the CPU oracle consume.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

ATTR_BOOL, ATTR_INT, ATTR_DOUBLE, ATTR_STRING, ATTR_LIST = range(5)
FWD_SYNAPSE, FWD_DENDRITE, FWD_SOMA = 1, 2, 4
FWD_ALL = 7
IMPL_SYNAPSE, IMPL_DENDRITE, IMPL_SOMA = 1, 2, 4
UNIT_LOG_ENERGY, UNIT_LOG_LATENCY, UNIT_UPDATE_EVERY_TIMESTEP = 1, 2, 4
BUF_BEFORE_DENDRITE, BUF_INSIDE_DENDRITE, BUF_BEFORE_SOMA, BUF_INSIDE_SOMA, BUF_BEFORE_AXON_OUT = range(5)

RESERVED_NEURON_ATTRIBUTES = {"soma_hw_name", "default_synapse_hw_name", "dendrite_hw_name",
                              "log_spikes", "log_potential", "log_v"}


from sanafe_amd.chip import HardwareMappingError  # noqa: E402,F401  (the product's exception class)


# --------------------------------------------------------------------------
# ctypes mirror of include/sanafe_desc.h
# --------------------------------------------------------------------------
class AttrTable(C.Structure):
    _fields_ = [("n", C.c_int64), ("key", C.POINTER(C.c_int32)), ("type", C.POINTER(C.c_uint8)),
                ("fwd", C.POINTER(C.c_uint8)), ("num", C.POINTER(C.c_double)), ("str", C.POINTER(C.c_int32)),
                ("list_ptr", C.POINTER(C.c_int64)), ("list_num", C.POINTER(C.c_double))]


class Desc(C.Structure):
    _fields_ = [
        ("n_strings", C.c_int32), ("strings", C.POINTER(C.c_char_p)),
        ("noc_width", C.c_int32), ("noc_height", C.c_int32), ("noc_buffer_size", C.c_int32),
        ("n_sync", C.c_int32), ("sync_key", C.POINTER(C.c_int64)), ("sync_val", C.POINTER(C.c_double)),
        ("n_tiles", C.c_int32), ("tile_name", C.POINTER(C.c_int32)),
        ("tile_hop_energy", C.POINTER(C.c_double)), ("tile_hop_latency", C.POINTER(C.c_double)),
        ("tile_log_energy", C.POINTER(C.c_uint8)),
        ("n_cores", C.c_int32), ("core_name", C.POINTER(C.c_int32)), ("core_tile", C.POINTER(C.c_int32)),
        ("core_buffer_pos", C.POINTER(C.c_int32)), ("core_max_neurons", C.POINTER(C.c_int64)),
        ("core_log_energy", C.POINTER(C.c_uint8)),
        ("core_template", C.POINTER(C.c_int32)),
        ("n_templates", C.c_int32),
        ("tmpl_axon_in_ptr", C.POINTER(C.c_int32)), ("axon_in_energy", C.POINTER(C.c_double)),
        ("axon_in_latency", C.POINTER(C.c_double)),
        ("tmpl_axon_out_ptr", C.POINTER(C.c_int32)), ("axon_out_energy", C.POINTER(C.c_double)),
        ("axon_out_latency", C.POINTER(C.c_double)),
        ("tmpl_unit_ptr", C.POINTER(C.c_int32)),
        ("n_units", C.c_int32), ("unit_name", C.POINTER(C.c_int32)), ("unit_model", C.POINTER(C.c_int32)),
        ("unit_plugin", C.POINTER(C.c_int32)), ("unit_implements", C.POINTER(C.c_uint8)),
        ("unit_flags", C.POINTER(C.c_uint8)), ("unit_attr_ptr", C.POINTER(C.c_int64)), ("unit_attrs", AttrTable),
        ("n_groups", C.c_int32), ("group_name", C.POINTER(C.c_int32)), ("group_ptr", C.POINTER(C.c_int64)),
        ("n_neurons", C.c_int64), ("neuron_core", C.POINTER(C.c_int32)), ("neuron_map_order", C.POINTER(C.c_int64)),
        ("neuron_soma_hw", C.POINTER(C.c_int32)), ("neuron_dendrite_hw", C.POINTER(C.c_int32)),
        ("neuron_synapse_hw", C.POINTER(C.c_int32)), ("neuron_log_spikes", C.POINTER(C.c_uint8)),
        ("neuron_log_potential", C.POINTER(C.c_uint8)), ("neuron_attr_ptr", C.POINTER(C.c_int64)),
        ("neuron_attrs", AttrTable),
        ("n_edges", C.c_int64), ("edge_src", C.POINTER(C.c_int64)), ("edge_dst", C.POINTER(C.c_int64)),
        ("edge_synapse_hw", C.POINTER(C.c_int32)), ("edge_weight", C.POINTER(C.c_double)),
        ("edge_delay", C.POINTER(C.c_int8)), ("edge_attr_ptr", C.POINTER(C.c_int64)), ("edge_attrs", AttrTable),
    ]


_CT = {np.dtype(np.int32): C.c_int32, np.dtype(np.int64): C.c_int64, np.dtype(np.uint8): C.c_uint8,
       np.dtype(np.int8): C.c_int8, np.dtype(np.float64): C.c_double}


def _ptr(a):
    return a.ctypes.data_as(C.POINTER(_CT[a.dtype]))


# --------------------------------------------------------------------------
# attribute values
# --------------------------------------------------------------------------
def py_to_attr(value, narrow_float=True):
    """Python value -> (type, num, str, list).  src/pymodule.cpp:118-175."""
    if isinstance(value, str):
        return (ATTR_STRING, 0.0, value, None)
    if hasattr(value, "dtype") and getattr(value, "ndim", 1) == 0:
        kind = value.dtype.kind
        if kind in "iu":
            return (ATTR_INT, float(int(value)), None, None)
        if kind == "f":
            return (ATTR_DOUBLE, float(np.float32(value)) if narrow_float else float(value), None, None)
        if kind == "b":
            return (ATTR_INT, float(bool(value)), None, None)
    if isinstance(value, (bool, np.bool_)):
        # pybind11 sees a Python bool as an int_ (bool subclasses int)
        return (ATTR_INT if narrow_float else ATTR_BOOL, float(bool(value)), None, None)
    if isinstance(value, (int, np.integer)):
        return (ATTR_INT, float(int(value)), None, None)
    if isinstance(value, (float, np.floating)):
        return (ATTR_DOUBLE, float(np.float32(value)) if narrow_float else float(value), None, None)
    if isinstance(value, dict):
        raise ValueError("named sub-attribute maps are not supported in neuron/edge attributes")
    if hasattr(value, "__iter__"):
        out = []
        for v in value:
            t, num, s, lst = py_to_attr(v, narrow_float)
            if t in (ATTR_STRING, ATTR_LIST):
                raise ValueError("only flat numeric lists are supported as attribute values")
            out.append(num)
        return (ATTR_LIST, 0.0, None, out)
    raise ValueError("Error: dict has unsupported type")


class _Strings:
    def __init__(self):
        self.ids = {}
        self.items = []

    def __call__(self, s):
        if s is None or s == "":
            return -1
        s = str(s)
        i = self.ids.get(s)
        if i is None:
            i = len(self.items)
            self.ids[s] = i
            self.items.append(s)
        return i


class _AttrRows:
    """Accumulates attribute records and emits an AttrTable + row pointers."""

    def __init__(self, strings):
        self.s = strings
        self.key, self.type, self.fwd, self.num, self.str, self.lists = [], [], [], [], [], []

    def add(self, key, attr, fwd=FWD_ALL):
        t, num, sval, lst = attr
        self.key.append(self.s(key))
        self.type.append(t)
        self.fwd.append(fwd)
        self.num.append(num)
        self.str.append(self.s(sval) if t == ATTR_STRING else -1)
        self.lists.append(lst if t == ATTR_LIST else None)

    def __len__(self):
        return len(self.key)


def _emit_attr_table(keys, types, fwds, nums, strs, lists, keep):
    n = len(keys)
    key = np.ascontiguousarray(keys, dtype=np.int32)
    typ = np.ascontiguousarray(types, dtype=np.uint8)
    fwd = np.ascontiguousarray(fwds, dtype=np.uint8)
    num = np.ascontiguousarray(nums, dtype=np.float64)
    sv = np.ascontiguousarray(strs, dtype=np.int32)
    lptr = np.zeros(n + 1, dtype=np.int64)
    payload = []
    if lists is not None:
        for i, l in lists.items():
            lptr[i + 1] = len(l)
            payload.append((i, l))
    np.cumsum(lptr, out=lptr)
    lnum = np.zeros(max(1, int(lptr[-1])), dtype=np.float64)
    for i, l in payload:
        lnum[lptr[i]:lptr[i + 1]] = l
    keep += [key, typ, fwd, num, sv, lptr, lnum]
    t = AttrTable()
    t.n = n
    t.key, t.type, t.fwd, t.num, t.str, t.list_ptr, t.list_num = (_ptr(key), _ptr(typ), _ptr(fwd), _ptr(num),
                                                                    _ptr(sv), _ptr(lptr), _ptr(lnum))
    return t


# --------------------------------------------------------------------------
# Architecture
# --------------------------------------------------------------------------
class PipelineUnit:
    def __init__(self, name, model, attributes, implements, plugin=None, log_energy=False, log_latency=False,
                 update_every_timestep=False):
        self.name, self.model, self.attributes = name, model, dict(attributes)
        self.implements, self.plugin = implements, plugin
        self.log_energy, self.log_latency, self.update_every_timestep = log_energy, log_latency, update_every_timestep


class CoreTemplate:
    """The axon units and pipeline units of a core.  Cores replicated from one
    description entry share one template object (each core still gets its own
    unit instances when a chip is built)."""

    def __init__(self):
        self.axon_in, self.units, self.axon_out = [], [], []


class Core:
    """CoreConfiguration (src/arch.hpp:154-169)."""

    def __init__(self, name, parent_tile_id, offset_within_tile, core_id, buffer_position=BUF_BEFORE_SOMA,
                 max_neurons_supported=1024, log_energy=False, template=None):
        self.name, self.parent_tile_id, self.offset_within_tile, self.id = name, parent_tile_id, offset_within_tile, core_id
        self.buffer_position, self.max_neurons_supported, self.log_energy = buffer_position, max_neurons_supported, log_energy
        self.template = template if template is not None else CoreTemplate()

    @property
    def axon_in(self):
        return self.template.axon_in

    @property
    def units(self):
        return self.template.units

    @property
    def axon_out(self):
        return self.template.axon_out

    def create_axon_in(self, name, energy_message_in=0.0, latency_message_in=0.0):
        self.axon_in.append((name, float(energy_message_in), float(latency_message_in)))

    def create_axon_out(self, name, energy_message_out=0.0, latency_message_out=0.0):
        self.axon_out.append((name, float(energy_message_out), float(latency_message_out)))

    def _merge_or_create(self, name, section, model, attributes, plugin, flags):
        """yaml_merge_or_create_hardware_unit (src/yaml_arch.cpp:149-186)."""
        bit = {"synapse": IMPL_SYNAPSE, "dendrite": IMPL_DENDRITE, "soma": IMPL_SOMA}[section]
        for u in self.units:
            if u.name == name:
                u.implements |= bit
                for k, v in attributes.items():
                    u.attributes.setdefault(k, v)  # std::map::merge keeps existing keys
                if plugin is not None:
                    u.plugin = plugin
                return u
        u = PipelineUnit(name, model, attributes, bit, plugin, **flags)
        self.units.append(u)
        return u

    def create_synapse(self, name, model="current_based", attributes=None, plugin=None, **flags):
        return self._merge_or_create(name, "synapse", model, _unit_attrs(model, attributes, plugin), plugin, flags)

    def create_dendrite(self, name, model="accumulator", attributes=None, plugin=None, **flags):
        return self._merge_or_create(name, "dendrite", model, _unit_attrs(model, attributes, plugin), plugin, flags)

    def create_soma(self, name, model="leaky_integrate_fire", attributes=None, plugin=None, **flags):
        return self._merge_or_create(name, "soma", model, _unit_attrs(model, attributes, plugin), plugin, flags)


def _unit_attrs(model, attributes, plugin):
    out = {}
    for k, v in (attributes or {}).items():
        out[k] = v if isinstance(v, tuple) else py_to_attr(v, narrow_float=False)
    out.setdefault("model", (ATTR_STRING, 0.0, model, None))
    if plugin is not None:
        out.setdefault("plugin", (ATTR_STRING, 0.0, str(plugin), None))
    return out


class Tile:
    def __init__(self, name, tile_id, energy_north_hop=0.0, latency_north_hop=0.0, energy_east_hop=0.0,
                 latency_east_hop=0.0, energy_south_hop=0.0, latency_south_hop=0.0, energy_west_hop=0.0,
                 latency_west_hop=0.0, log_energy=False):
        self.name, self.id, self.log_energy = name, tile_id, log_energy
        self.hop_energy = [energy_north_hop, energy_east_hop, energy_south_hop, energy_west_hop]
        self.hop_latency = [latency_north_hop, latency_east_hop, latency_south_hop, latency_west_hop]
        self.cores = []


class Architecture:
    """Architecture (src/arch.hpp:70-101, src/arch.cpp:41-117)."""

    def __init__(self, name="", width=1, height=1, link_buffer_size=0, sync_table=None):
        self.name, self.noc_width, self.noc_height, self.noc_buffer_size = name, int(width), int(height), int(link_buffer_size)
        self.sync_table = dict(sync_table) if sync_table else {0: 0.0}
        self.tiles = []
        self._cores = []

    def create_tile(self, name, **metrics):
        t = Tile(name, len(self.tiles), **metrics)
        self.tiles.append(t)
        return t

    def create_core(self, name, parent_tile_id, buffer_position=BUF_BEFORE_SOMA, buffer_inside_unit=False,
                    max_neurons_supported=1024, log_energy=False, template=None, share_units_with=None):
        if share_units_with is not None:
            template = share_units_with.template
        if isinstance(buffer_position, str):
            buffer_position = parse_buffer_position(buffer_position, buffer_inside_unit)
        tile = self.tiles[parent_tile_id]
        core = Core(name, parent_tile_id, len(tile.cores), len(self._cores), buffer_position, max_neurons_supported,
                    log_energy, template)
        tile.cores.append(core)
        self._cores.append(core)
        return core

    def cores(self):
        return list(self._cores)

    def tile_cores(self, tile):
        return list(self.tiles[tile].cores)

    @property
    def core_count(self):
        return len(self._cores)


def parse_buffer_position(s, inside):
    """pipeline_parse_buffer_pos_str (src/pipeline.cpp:268-310)."""
    if s == "dendrite":
        return BUF_INSIDE_DENDRITE if inside else BUF_BEFORE_DENDRITE
    if s == "soma":
        return BUF_INSIDE_SOMA if inside else BUF_BEFORE_SOMA
    if s == "axon_out":
        return BUF_BEFORE_AXON_OUT
    raise ValueError("Error: Buffer position not supported")


# --------------------------------------------------------------------------
# Network
# --------------------------------------------------------------------------
class Neuron:
    """Lightweight reference to one neuron of a group (PyNeuronRef)."""
    __slots__ = ("group", "offset")

    def __init__(self, group, offset):
        self.group, self.offset = group, offset

    def get_id(self):
        return self.offset

    def map_to_core(self, core):
        self.group.map_to_core(core, self.offset, self.offset + 1)

    def set_attributes(self, soma_hw_name=None, default_synapse_hw_name=None, dendrite_hw_name=None, log_spikes=None,
                       log_potential=None, model_attributes=None, soma_attributes=None, dendrite_attributes=None):
        self.group.set_neuron_attributes(self.offset, soma_hw_name, default_synapse_hw_name, dendrite_hw_name,
                                         log_spikes, log_potential, model_attributes, soma_attributes, dendrite_attributes)

    def connect_to_neuron(self, dest, attributes=None):
        attributes = attributes or {}
        w = attributes.get("w", attributes.get("weight", 0.0))
        d = attributes.get("delay", attributes.get("d", None))
        if "tap" in attributes:
            if d is not None:
                raise NotImplementedError("an edge with both `delay` and `tap`")
            d = 64 + int(attributes["tap"])  # include/sanafe_desc.h: edge_delay carries 64 + tap index
        net = self.group.net
        net._add_edges(np.array([self.group.base + self.offset]), np.array([dest.group.base + dest.offset]),
                       np.array([float(np.float32(w)) if isinstance(w, float) else float(w)]),
                       None if d is None else np.array([int(d)]))


class NeuronGroup:
    def __init__(self, net, name, count, base):
        self.net, self.name, self.count, self.base = net, str(name), int(count), int(base)
        n = self.count
        self.core = np.full(n, -1, dtype=np.int32)
        self.map_order = np.zeros(n, dtype=np.int64)
        self.soma_hw = np.full(n, -1, dtype=np.int32)
        self.dendrite_hw = np.full(n, -1, dtype=np.int32)
        self.synapse_hw = np.full(n, -1, dtype=np.int32)
        self.log_spikes = np.zeros(n, dtype=np.uint8)
        self.log_potential = np.zeros(n, dtype=np.uint8)
        # attribute columns: key -> dict(mask, type, fwd, num, str) ; list values kept sparse
        self.cols = {}
        self.list_vals = {}  # (key, offset) -> list

    # -- reference API -------------------------------------------------
    def get_name(self):
        return self.name

    def __len__(self):
        return self.count

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [Neuron(self, j) for j in range(*i.indices(self.count))]
        if i < 0 or i >= self.count:
            raise IndexError
        return Neuron(self, int(i))

    def __iter__(self):
        for i in range(self.count):
            yield Neuron(self, i)

    @property
    def neurons(self):
        return self

    # -- columnar setters ------------------------------------------------
    def _col(self, key):
        c = self.cols.get(key)
        if c is None:
            n = self.count
            c = dict(mask=np.zeros(n, dtype=bool), type=np.zeros(n, dtype=np.uint8), fwd=np.zeros(n, dtype=np.uint8),
                     num=np.zeros(n, dtype=np.float64), str=np.full(n, -1, dtype=np.int32))
            self.cols[key] = c
        return c

    def set_attribute(self, key, attr, fwd=FWD_ALL, lo=0, hi=None):
        """Set one attribute (already typed tuple) on neurons [lo, hi)."""
        if key in RESERVED_NEURON_ATTRIBUTES:
            raise ValueError("Reserved neuron attribute '%s' cannot be used as a model attribute. "
                             "Pass it as a direct argument instead." % key)
        hi = self.count if hi is None else hi
        t, num, sval, lst = attr
        c = self._col(key)
        c["mask"][lo:hi] = True
        c["type"][lo:hi] = t
        c["fwd"][lo:hi] = fwd
        c["num"][lo:hi] = num
        c["str"][lo:hi] = self.net.strings(sval) if t == ATTR_STRING else -1
        if t == ATTR_LIST:
            for i in range(lo, hi):
                self.list_vals[(key, i)] = list(lst)

    def set_attribute_column(self, key, values, attr_type=None, fwd=FWD_ALL):
        """Vectorised per-neuron numeric attribute (one value per neuron)."""
        values = np.asarray(values)
        if attr_type is None:
            attr_type = ATTR_INT if values.dtype.kind in "iub" else ATTR_DOUBLE
        c = self._col(key)
        c["mask"][:] = True
        c["type"][:] = attr_type
        c["fwd"][:] = fwd
        c["num"][:] = values.astype(np.float64)
        c["str"][:] = -1

    def apply_config(self, lo, hi, soma_hw_name=None, default_synapse_hw_name=None, dendrite_hw_name=None,
                     log_spikes=None, log_potential=None, attrs=None):
        """Neuron::set_attributes (src/network.cpp:94-128) on a range."""
        s = self.net.strings
        if default_synapse_hw_name is not None:
            self.synapse_hw[lo:hi] = s(default_synapse_hw_name)
        if dendrite_hw_name is not None:
            self.dendrite_hw[lo:hi] = s(dendrite_hw_name)
        if soma_hw_name is not None:
            self.soma_hw[lo:hi] = s(soma_hw_name)
        if log_spikes is not None:
            self.log_spikes[lo:hi] = bool(log_spikes)
        if log_potential is not None:
            self.log_potential[lo:hi] = bool(log_potential)
        for key, (attr, fwd) in (attrs or {}).items():
            self.set_attribute(key, attr, fwd, lo, hi)

    def set_neuron_attributes(self, offset, soma_hw_name=None, default_synapse_hw_name=None, dendrite_hw_name=None,
                              log_spikes=None, log_potential=None, model_attributes=None, soma_attributes=None,
                              dendrite_attributes=None):
        attrs = {}
        for k, v in (model_attributes or {}).items():
            attrs[k] = (py_to_attr(v), FWD_ALL)
        for k, v in (dendrite_attributes or {}).items():
            attrs[k] = (py_to_attr(v), FWD_DENDRITE)
        for k, v in (soma_attributes or {}).items():
            attrs[k] = (py_to_attr(v), FWD_SOMA)
        self.apply_config(offset, offset + 1, soma_hw_name, default_synapse_hw_name, dendrite_hw_name, log_spikes,
                          log_potential, attrs)

    def map_to_core(self, core, lo=0, hi=None):
        """Neuron::map_to_core (src/network.cpp:85-92) for neurons [lo, hi) in offset order."""
        hi = self.count if hi is None else hi
        self.core[lo:hi] = core.id
        k = hi - lo
        self.map_order[lo:hi] = self.net._mapping_count + 1 + np.arange(k, dtype=np.int64)
        self.net._mapping_count += k

    # -- connectivity (src/network.cpp:229-605) --------------------------
    @staticmethod
    def _weights(attributes, count):
        attributes = attributes or {}
        w = attributes.get("w", attributes.get("weight"))
        d = attributes.get("delay", attributes.get("d"))
        if "tap" in attributes:
            if d is not None:
                raise NotImplementedError("edges with both `delay` and `tap`")
            d = 64 + np.asarray(attributes["tap"], dtype=np.int64)  # include/sanafe_desc.h: 64 + tap index
        extra = set(attributes) - {"w", "weight", "delay", "d", "tap"}
        if extra:
            raise NotImplementedError("edge attributes other than weight/delay/tap: %s" % sorted(extra))
        return w, d

    @staticmethod
    def _narrow(w, narrow):
        w = np.asarray(w)
        if w.dtype.kind == "f" and narrow:
            return w.astype(np.float32).astype(np.float64)
        return w.astype(np.float64)

    def connect_neurons_sparse(self, dest_group, attributes, src_dest_id_pairs, narrow_float=True):
        pairs = np.asarray(src_dest_id_pairs, dtype=np.int64).reshape(-1, 2)
        w, d = self._weights(attributes, len(pairs))
        if (pairs[:, 0] >= self.count).any() or (pairs[:, 0] < 0).any():
            raise ValueError("Error: src id is out of range.")
        if (pairs[:, 1] >= dest_group.count).any() or (pairs[:, 1] < 0).any():
            raise ValueError("Error: dest nid is out of range.")
        for v in (w, d):
            if v is not None and len(v) != len(pairs):
                raise ValueError("Error: Length of attribute list != number of defined edges.")
        wv = np.zeros(len(pairs)) if w is None else self._narrow(w, narrow_float)
        dv = None if d is None else np.asarray(d, dtype=np.int64)
        self.net._add_edges(self.base + pairs[:, 0], dest_group.base + pairs[:, 1], wv, dv,
                            dest_group.synapse_hw[pairs[:, 1]])

    def connect_neurons_dense(self, dest_group, attributes, narrow_float=True):
        ns, nd = self.count, dest_group.count
        w, d = self._weights(attributes, ns * nd)
        for v in (w, d):
            if v is not None and len(v) < ns * nd:
                raise ValueError("Not enough entries defined for attribute")
        src = np.repeat(np.arange(ns, dtype=np.int64), nd)
        dst = np.tile(np.arange(nd, dtype=np.int64), ns)
        wv = np.zeros(ns * nd) if w is None else self._narrow(w, narrow_float)[: ns * nd]
        dv = None if d is None else np.asarray(d, dtype=np.int64)[: ns * nd]
        self.net._add_edges(self.base + src, dest_group.base + dst, wv, dv, dest_group.synapse_hw[dst])

    def connect_neurons_conv2d(self, dest_group, attributes, input_width, input_height, input_channels, kernel_width,
                               kernel_height, kernel_count=1, stride_width=1, stride_height=1, narrow_float=True):
        for nm, v in (("input_width", input_width), ("input_height", input_height), ("input_channels", input_channels),
                      ("kernel_width", kernel_width), ("kernel_height", kernel_height), ("kernel_count", kernel_count),
                      ("stride_width", stride_width), ("stride_height", stride_height)):
            if v <= 0:
                raise ValueError("Error: Conv2D parameter '%s' must be > 0 (got %d)." % (nm, v))
        if kernel_width > input_width or kernel_height > input_height:
            raise ValueError("Error: Conv2D kernel larger than input with zero padding.")
        ow = (input_width - kernel_width) // stride_width + 1
        oh = (input_height - kernel_height) // stride_height + 1
        oc = kernel_count
        if input_channels * input_width * input_height != self.count:
            raise ValueError("Expected %d neurons in source group for convolution but there are %d neurons.\n"
                             % (input_channels * input_width * input_height, self.count))
        if oc * ow * oh != dest_group.count:
            raise ValueError("Expected %d neurons in dest group for convolution but there are %d neurons.\n"
                             % (oc * ow * oh, dest_group.count))
        w, d = self._weights(attributes, 0)
        # creation order: c_out, y_out, x_out, c_in, y_filter, x_filter  (src/network.cpp:310-374)
        co, yo, xo, ci, yf, xf = np.meshgrid(np.arange(oc), np.arange(oh), np.arange(ow), np.arange(input_channels),
                                             np.arange(kernel_height), np.arange(kernel_width), indexing="ij")
        co, yo, xo, ci, yf, xf = (a.ravel().astype(np.int64) for a in (co, yo, xo, ci, yf, xf))
        ypos = yo * stride_height + yf
        xpos = xo * stride_width + xf
        ok = (ypos < input_height) & (xpos < input_width)
        co, yo, xo, ci, yf, xf, ypos, xpos = (a[ok] for a in (co, yo, xo, ci, yf, xf, ypos, xpos))
        dst = co * ow * oh + yo * ow + xo
        src = ci * input_width * input_height + ypos * input_width + xpos
        fidx = yf * kernel_width * input_channels * kernel_count + xf * input_channels * kernel_count + ci * kernel_count + co
        for v in (w, d):
            if v is not None and len(v) <= fidx.max():
                raise ValueError("Not enough entries defined for attribute")
        wv = np.zeros(len(src)) if w is None else self._narrow(w, narrow_float)[fidx]
        dv = None if d is None else np.asarray(d, dtype=np.int64)[fidx]
        self.net._add_edges(self.base + src, dest_group.base + dst, wv, dv, dest_group.synapse_hw[dst])


class Network:
    """SpikingNetwork (src/network.hpp:148-176)."""

    def __init__(self, name=""):
        self.name = name
        self.strings = _Strings()
        self.groups = {}
        self._order = []
        self._n = 0
        self._mapping_count = 0
        self._edge_blocks = []
        self._n_edges = 0

    def lower_for_chip(self, arch):
        """What SpikingChip.load() needs from a front-end other than sanafecpp_amd: (address of the sanafe_desc,
        [(group, first neuron id, count)], log_spikes, log_potential, object that keeps the buffers alive)."""
        built = to_desc(arch, self)
        cat = lambda name: (np.concatenate([getattr(g, name) for g in self._order]).astype(np.uint8)  # noqa: E731
                            if self._order else np.zeros(0, np.uint8))
        return (C.addressof(built.desc), [(g.name, int(g.base), int(g.count)) for g in self._order], cat("log_spikes"),
                cat("log_potential"), built)

    def __getitem__(self, name):
        return self.groups[str(name)]

    def create_neuron_group(self, group_name, neuron_count, model_attributes=None, default_synapse_hw_name="",
                            default_dendrite_hw_name="", log_potential=False, log_spikes=False, soma_hw_name="",
                            _typed_attrs=None):
        group_name = str(group_name)
        if group_name in self.groups:
            raise ValueError("Group: %s already exists in SNN." % group_name)
        g = NeuronGroup(self, group_name, neuron_count, self._n)
        self._n += g.count
        self.groups[group_name] = g
        self._order.append(g)
        attrs = dict(_typed_attrs or {})
        for k, v in (model_attributes or {}).items():
            attrs[k] = (py_to_attr(v), FWD_ALL)
        g.apply_config(0, g.count, soma_hw_name or None, default_synapse_hw_name or None,
                       default_dendrite_hw_name or None, log_spikes, log_potential, attrs)
        return g

    @property
    def neuron_count(self):
        return self._n

    @property
    def edge_count(self):
        return self._n_edges

    def _add_edges(self, src, dst, w, delay=None, syn_hw=None):
        """Edges in creation order.  ``syn_hw`` is the post-neuron's default synapse
        unit at creation time (Neuron::connect_to_neuron, src/network.cpp:175-192)."""
        n = len(src)
        if syn_hw is None:
            syn_hw = self._neuron_col("synapse_hw", dst)
        self._edge_blocks.append((np.ascontiguousarray(src, dtype=np.int64), np.ascontiguousarray(dst, dtype=np.int64),
                                  np.ascontiguousarray(w, dtype=np.float64),
                                  None if delay is None else np.ascontiguousarray(delay, dtype=np.int8),
                                  np.ascontiguousarray(syn_hw, dtype=np.int32)))
        self._n_edges += n

    def _neuron_col(self, name, gids):
        gids = np.asarray(gids, dtype=np.int64)
        bases = np.array([g.base for g in self._order] + [self._n], dtype=np.int64)
        gi = np.searchsorted(bases, gids, side="right") - 1
        out = np.empty(len(gids), dtype=np.int32)
        for k in np.unique(gi):
            sel = gi == k
            g = self._order[k]
            out[sel] = getattr(g, name)[gids[sel] - g.base]
        return out


# --------------------------------------------------------------------------
# lowering to sanafe_desc
# --------------------------------------------------------------------------
class BuiltDesc:
    """Owns the numpy buffers behind a ``Desc`` struct."""

    def __init__(self, desc, keep, arch, net):
        self.desc, self._keep, self.arch, self.net = desc, keep, arch, net

    @property
    def ptr(self):
        return C.byref(self.desc)


def _emit_group_attrs(net, keep):
    """Per-neuron attribute rows, key-sorted per neuron (std::map order)."""
    n_total = net._n
    counts = np.zeros(n_total, dtype=np.int64)
    per_group = []
    for g in net._order:
        keys = sorted(g.cols)
        per_group.append(keys)
        for k in keys:
            counts[g.base:g.base + g.count] += g.cols[k]["mask"]
    ptr = np.zeros(n_total + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    total = int(ptr[-1])
    key = np.zeros(total, dtype=np.int32)
    typ = np.zeros(total, dtype=np.uint8)
    fwd = np.zeros(total, dtype=np.uint8)
    num = np.zeros(total, dtype=np.float64)
    sv = np.full(total, -1, dtype=np.int32)
    lists = {}
    for g, keys in zip(net._order, per_group):
        cursor = ptr[g.base:g.base + g.count].copy()
        for k in keys:
            c = g.cols[k]
            m = c["mask"]
            pos = cursor[m]
            key[pos] = net.strings(k)
            typ[pos] = c["type"][m]
            fwd[pos] = c["fwd"][m]
            num[pos] = c["num"][m]
            sv[pos] = c["str"][m]
            if g.list_vals:
                idx = np.nonzero(m & (c["type"] == ATTR_LIST))[0]
                for i in idx:
                    lists[int(cursor[i])] = g.list_vals[(k, int(i))]
            cursor[m] += 1
    keep.append(ptr)
    return ptr, _emit_attr_table(key, typ, fwd, num, sv, lists, keep)


def to_desc(arch: Architecture, net: Network) -> BuiltDesc:
    s = net.strings
    keep = []
    d = Desc()

    def arr(values, dtype):
        a = values if (isinstance(values, np.ndarray) and values.dtype == dtype and values.flags.c_contiguous) \
            else np.ascontiguousarray(values, dtype=dtype)
        if a.size == 0:
            a = np.zeros(1, dtype=dtype)
        keep.append(a)
        return a

    # ---- NoC / tiles / cores / units
    d.noc_width, d.noc_height, d.noc_buffer_size = arch.noc_width, arch.noc_height, arch.noc_buffer_size
    sk = sorted(arch.sync_table)
    d.n_sync = len(sk)
    d.sync_key = _ptr(arr(sk, np.int64))
    d.sync_val = _ptr(arr([arch.sync_table[k] for k in sk], np.float64))
    tiles = arch.tiles
    d.n_tiles = len(tiles)
    d.tile_name = _ptr(arr([s(t.name) for t in tiles], np.int32))
    d.tile_hop_energy = _ptr(arr([e for t in tiles for e in t.hop_energy], np.float64))
    d.tile_hop_latency = _ptr(arr([e for t in tiles for e in t.hop_latency], np.float64))
    d.tile_log_energy = _ptr(arr([t.log_energy for t in tiles], np.uint8))
    cores = arch.cores()
    d.n_cores = len(cores)
    d.core_name = _ptr(arr([s(c.name) for c in cores], np.int32))
    d.core_tile = _ptr(arr([c.parent_tile_id for c in cores], np.int32))
    d.core_buffer_pos = _ptr(arr([c.buffer_position for c in cores], np.int32))
    d.core_max_neurons = _ptr(arr([c.max_neurons_supported for c in cores], np.int64))
    d.core_log_energy = _ptr(arr([c.log_energy for c in cores], np.uint8))
    ain_ptr, ain_e, ain_l, aout_ptr, aout_e, aout_l, unit_ptr = [0], [], [], [0], [], [], [0]
    u_name, u_model, u_plugin, u_impl, u_flags, u_attr_ptr = [], [], [], [], [], [0]
    rows = _AttrRows(s)
    tmpl_index = {}
    core_tmpl = []
    for c in cores:
        tid = tmpl_index.get(id(c.template))
        if tid is None:
            tid = len(tmpl_index)
            tmpl_index[id(c.template)] = tid
            for (_, e, l) in c.axon_in:
                ain_e.append(e)
                ain_l.append(l)
            ain_ptr.append(len(ain_e))
            for (_, e, l) in c.axon_out:
                aout_e.append(e)
                aout_l.append(l)
            aout_ptr.append(len(aout_e))
            for u in c.units:
                u_name.append(s(u.name))
                u_model.append(s(u.model))
                u_plugin.append(s(u.plugin) if u.plugin else -1)
                u_impl.append(u.implements)
                u_flags.append((UNIT_LOG_ENERGY if u.log_energy else 0) | (UNIT_LOG_LATENCY if u.log_latency else 0)
                               | (UNIT_UPDATE_EVERY_TIMESTEP if u.update_every_timestep else 0))
                for k in sorted(u.attributes):
                    rows.add(k, u.attributes[k])
                u_attr_ptr.append(len(rows))
            unit_ptr.append(len(u_name))
        core_tmpl.append(tid)
    d.core_template = _ptr(arr(core_tmpl, np.int32))
    d.n_templates = len(tmpl_index)
    d.tmpl_axon_in_ptr = _ptr(arr(ain_ptr, np.int32))
    d.axon_in_energy = _ptr(arr(ain_e, np.float64))
    d.axon_in_latency = _ptr(arr(ain_l, np.float64))
    d.tmpl_axon_out_ptr = _ptr(arr(aout_ptr, np.int32))
    d.axon_out_energy = _ptr(arr(aout_e, np.float64))
    d.axon_out_latency = _ptr(arr(aout_l, np.float64))
    d.tmpl_unit_ptr = _ptr(arr(unit_ptr, np.int32))
    d.n_units = len(u_name)
    d.unit_name = _ptr(arr(u_name, np.int32))
    d.unit_model = _ptr(arr(u_model, np.int32))
    d.unit_plugin = _ptr(arr(u_plugin, np.int32))
    d.unit_implements = _ptr(arr(u_impl, np.uint8))
    d.unit_flags = _ptr(arr(u_flags, np.uint8))
    d.unit_attr_ptr = _ptr(arr(u_attr_ptr, np.int64))
    lists = {i: l for i, l in enumerate(rows.lists) if l is not None}
    d.unit_attrs = _emit_attr_table(rows.key, rows.type, rows.fwd, rows.num, rows.str, lists, keep)

    # ---- groups / neurons
    groups = net._order
    d.n_groups = len(groups)
    d.group_name = _ptr(arr([s(g.name) for g in groups], np.int32))
    d.group_ptr = _ptr(arr([g.base for g in groups] + [net._n], np.int64))
    d.n_neurons = net._n

    def cat(name, dtype):
        return arr(np.concatenate([getattr(g, name) for g in groups]) if groups else [], dtype)

    d.neuron_core = _ptr(cat("core", np.int32))
    d.neuron_map_order = _ptr(cat("map_order", np.int64))
    d.neuron_soma_hw = _ptr(cat("soma_hw", np.int32))
    d.neuron_dendrite_hw = _ptr(cat("dendrite_hw", np.int32))
    d.neuron_synapse_hw = _ptr(cat("synapse_hw", np.int32))
    d.neuron_log_spikes = _ptr(cat("log_spikes", np.uint8))
    d.neuron_log_potential = _ptr(cat("log_potential", np.uint8))
    nptr, d.neuron_attrs = _emit_group_attrs(net, keep)
    d.neuron_attr_ptr = _ptr(nptr)

    # ---- edges
    blocks = net._edge_blocks
    d.n_edges = net._n_edges
    if len(blocks) == 1:
        src, dst, w, _, hw = blocks[0]  # no copy: synthetic networks are one multi-GB block
        dl = blocks[0][3]
    elif blocks:
        src = np.concatenate([b[0] for b in blocks])
        dst = np.concatenate([b[1] for b in blocks])
        w = np.concatenate([b[2] for b in blocks])
        hw = np.concatenate([b[4] for b in blocks])
        if any(b[3] is not None for b in blocks):
            dl = np.concatenate([b[3] if b[3] is not None else np.full(len(b[0]), -1, dtype=np.int8) for b in blocks])
        else:
            dl = None
    else:
        src = dst = np.zeros(0, dtype=np.int64)
        w = np.zeros(0)
        hw = np.zeros(0, dtype=np.int32)
        dl = None
    d.edge_src = _ptr(arr(src, np.int64))
    d.edge_dst = _ptr(arr(dst, np.int64))
    d.edge_weight = _ptr(arr(w, np.float64))
    d.edge_synapse_hw = _ptr(arr(hw, np.int32))
    d.edge_delay = _ptr(arr(dl, np.int8)) if dl is not None else None
    d.edge_attr_ptr = None
    d.edge_attrs = AttrTable()

    # ---- strings last (everything above interns)
    enc = [x.encode() for x in s.items]
    sarr = (C.c_char_p * max(1, len(enc)))(*enc)
    keep += [enc, sarr]
    d.n_strings = len(enc)
    d.strings = C.cast(sarr, C.POINTER(C.c_char_p))
    return BuiltDesc(d, keep, arch, net)


def describe(built: BuiltDesc):
    """Expand a built desc into plain Python data with names resolved (for comparisons in tests)."""
    d = built.desc
    strings = [d.strings[i].decode() for i in range(d.n_strings)]

    def name(i):
        return "" if i < 0 else strings[i]

    def attrs(table, lo, hi):
        out = {}
        for i in range(lo, hi):
            t = table.type[i]
            if t == ATTR_STRING:
                v = name(table.str[i])
            elif t == ATTR_LIST:
                v = [table.list_num[j] for j in range(table.list_ptr[i], table.list_ptr[i + 1])]
            else:
                v = table.num[i]
            out[name(table.key[i])] = (int(t), v, int(table.fwd[i]))
        return out

    tiles = [dict(name=name(d.tile_name[t]), e=[d.tile_hop_energy[4 * t + k] for k in range(4)],
                  l=[d.tile_hop_latency[4 * t + k] for k in range(4)], log=int(d.tile_log_energy[t]))
             for t in range(d.n_tiles)]
    cores = []
    for c in range(d.n_cores):
        units = []
        tm = d.core_template[c]
        for u in range(d.tmpl_unit_ptr[tm], d.tmpl_unit_ptr[tm + 1]):
            units.append(dict(name=name(d.unit_name[u]), model=name(d.unit_model[u]), plugin=name(d.unit_plugin[u]),
                              impl=int(d.unit_implements[u]), flags=int(d.unit_flags[u]),
                              attrs=attrs(d.unit_attrs, d.unit_attr_ptr[u], d.unit_attr_ptr[u + 1])))
        cores.append(dict(name=name(d.core_name[c]), tile=int(d.core_tile[c]), buf=int(d.core_buffer_pos[c]),
                          max_neurons=int(d.core_max_neurons[c]), log=int(d.core_log_energy[c]),
                          axon_in=[(d.axon_in_energy[i], d.axon_in_latency[i])
                                   for i in range(d.tmpl_axon_in_ptr[tm], d.tmpl_axon_in_ptr[tm + 1])],
                          axon_out=[(d.axon_out_energy[i], d.axon_out_latency[i])
                                    for i in range(d.tmpl_axon_out_ptr[tm], d.tmpl_axon_out_ptr[tm + 1])],
                          units=units))
    groups = []
    for g in range(d.n_groups):
        ns = []
        for n in range(d.group_ptr[g], d.group_ptr[g + 1]):
            ns.append(dict(core=int(d.neuron_core[n]), order=int(d.neuron_map_order[n]), soma=name(d.neuron_soma_hw[n]),
                           dendrite=name(d.neuron_dendrite_hw[n]), synapse=name(d.neuron_synapse_hw[n]),
                           log_spikes=int(d.neuron_log_spikes[n]), log_potential=int(d.neuron_log_potential[n]),
                           attrs=attrs(d.neuron_attrs, d.neuron_attr_ptr[n], d.neuron_attr_ptr[n + 1])))
        groups.append(dict(name=name(d.group_name[g]), neurons=ns))
    edges = [(int(d.edge_src[e]), int(d.edge_dst[e]), name(d.edge_synapse_hw[e]), d.edge_weight[e],
              int(d.edge_delay[e]) if d.edge_delay else -1) for e in range(d.n_edges)]
    return dict(noc=(d.noc_width, d.noc_height, d.noc_buffer_size),
                sync={int(d.sync_key[i]): d.sync_val[i] for i in range(d.n_sync)},
                tiles=tiles, cores=cores, groups=groups, edges=edges)
