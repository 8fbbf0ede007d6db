"""``SpikingChip``: the reference's Python entry point for the simulation loop
(``sanafe.SpikingChip``, src/pymodule.cpp:1170-1212) on top of the MI355X host
library (include/sanafe_host.h).

There is no CPU fallback: constructing a chip without the built HIP libraries
or without a gfx950 device raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import description as D

HERE = os.path.dirname(os.path.abspath(__file__))

try:  # C++17 / PyBind11 description objects (the product front-end); built by `make -C sana-fe_amd`
    from . import sanafecpp_amd as cpp
except ImportError:  # pragma: no cover - the Python twin in description.py still works for tests
    cpp = None


class _Lowered:
    """A lowered description + what SpikingChip needs to label traces, from either front-end."""

    def __init__(self, arch, net):
        if cpp is not None and isinstance(net, cpp.Network):
            if not isinstance(arch, cpp.Architecture):
                raise TypeError("a sanafecpp_amd.Network needs a sanafecpp_amd.Architecture")
            self.handle = cpp.to_desc(arch, net)
            self.address = self.handle.address
            self.n_neurons = int(self.handle.n_neurons)
            self.groups = {name: (int(base), int(count)) for name, base, count in net.group_table()}
            ls, lp = net.log_flags()
            self.log_spikes, self.log_potential = np.asarray(ls).astype(bool), np.asarray(lp).astype(bool)
        else:
            self.handle = D.to_desc(arch, net)
            self.address = C.addressof(self.handle.desc)
            self.n_neurons = int(self.handle.desc.n_neurons)
            self.groups = {g.name: (g.base, g.count) for g in net._order}
            cat = lambda name: (np.concatenate([getattr(g, name) for g in net._order]).astype(bool)  # noqa: E731
                                if net._order else np.zeros(0, bool))
            self.log_spikes, self.log_potential = cat("log_spikes"), cat("log_potential")


class Totals(C.Structure):
    """sanafe_hip_totals (include/sanafe_hip.h)."""
    _fields_ = [(n, C.c_int64) for n in ("timesteps", "spikes", "packets_sent", "neurons_updated", "neurons_fired",
                                         "total_hops")] + \
               [(n, C.c_double) for n in ("total_energy", "synapse_energy", "dendrite_energy", "soma_energy",
                                          "network_energy", "sim_time")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


TOTALS_DTYPE = np.dtype([(n, np.int64) for n in ("timesteps", "spikes", "packets_sent", "neurons_updated",
                                                 "neurons_fired", "total_hops")] +
                        [(n, np.float64) for n in ("total_energy", "synapse_energy", "dendrite_energy", "soma_energy",
                                                   "network_energy", "sim_time")])


class ChipInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_cores", "n_local_cores", "n_slots", "n_global_slots", "ring_slots",
                                          "n_slices")] + \
               [(n, C.c_uint64) for n in ("n_neurons", "n_axons", "n_synapses", "mapped_tiles", "mapped_cores")] + \
               [("n_soma_classes", C.c_uint32), ("n_cost_classes", C.c_uint32), ("sync_delay", C.c_double),
                ("image_bytes", C.c_uint64)]


MSG_INT = ("timestep", "mid", "src_neuron", "src_tile", "src_core_offset", "src_core_id", "dest_tile",
           "dest_core_offset", "dest_core_id", "dest_axon_id", "hops", "spikes", "placeholder",
           "src_x", "src_y", "dest_x", "dest_y")
MSG_DBL = ("generation_delay", "processing_delay", "network_delay", "blocking_delay", "min_hop_delay",
           "sent_timestamp", "received_timestamp", "processed_timestamp", "messages_along_route")
MSG_DTYPE = np.dtype([(n, np.int64) for n in MSG_INT] + [(n, np.float64) for n in MSG_DBL])

_lib = None
COMM_ID_BYTES = 128
# int (*sanafe_allgather_fn)(void *ctx, const void *send, uint64_t bytes, void *recv)  (include/sanafe_host.h)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class BackendMissingError(RuntimeError):
    pass


def lib():
    """Loads host/libsanafe_host.so (which loads csrc/libsanafe_hip.so); never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(HERE, "host", "libsanafe_host.so")
    hip = os.path.join(HERE, "csrc", "libsanafe_hip.so")
    for p in (hip, path):
        if not os.path.exists(p):
            raise BackendMissingError("%s is missing: build it with `make -C sana-fe_amd` "
                                      "(or __graft_entry__.build()); there is no CPU fallback" % p)
    C.CDLL(hip, mode=C.RTLD_GLOBAL)
    L = C.CDLL(path)
    L.sanafe_last_error.restype = C.c_char_p
    L.sanafe_chip_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.sanafe_chip_destroy.argtypes = [C.c_void_p]
    L.sanafe_chip_get_info.argtypes = [C.c_void_p, C.POINTER(ChipInfo)]
    L.sanafe_chip_device.argtypes = [C.c_void_p]
    L.sanafe_chip_device.restype = C.c_void_p
    L.sanafe_chip_sim.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(Totals)]
    L.sanafe_chip_set_scheduler_threads.argtypes = [C.c_void_p, C.c_int]
    L.sanafe_chip_reset.argtypes = [C.c_void_p]
    L.sanafe_chip_get_power.argtypes = [C.c_void_p]
    L.sanafe_chip_get_power.restype = C.c_double
    for n in ("sanafe_chip_get_status", "sanafe_chip_get_potentials", "sanafe_chip_get_input_current"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
    L.sanafe_chip_get_step_totals.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    L.sanafe_chip_get_step_fired.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.sanafe_chip_get_step_messages.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    L.sanafe_chip_get_step_messages.restype = C.c_int64
    L.sanafe_chip_set_bias.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.sanafe_chip_step_neurons.argtypes = [C.c_void_p]
    L.sanafe_chip_step_deliver.argtypes = [C.c_void_p, C.c_int]
    L.sanafe_chip_spike_buffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sanafe_chip_synchronize.argtypes = [C.c_void_p]
    L.sanafe_chip_read_totals.argtypes = [C.c_void_p, C.POINTER(Totals)]
    L.sanafe_chip_perf_columns.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.sanafe_chip_perf_columns.restype = C.c_int64
    L.sanafe_chip_get_step_optional.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    L.sanafe_chip_set_state_log.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    L.sanafe_chip_get_step_state.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    L.sanafe_chip_total_timesteps.argtypes = [C.c_void_p]
    L.sanafe_chip_total_timesteps.restype = C.c_int64
    L.sanafe_comm_unique_id.argtypes = [C.c_void_p]
    L.sanafe_chip_comm_init_rccl.argtypes = [C.c_void_p, C.c_void_p]
    L.sanafe_chip_comm_init_callback.argtypes = [C.c_void_p, ALLGATHER_FN, C.c_void_p]
    L.sanafe_generate_random_edges.argtypes = [C.c_int64, C.c_int64, C.c_uint64, C.c_int, C.c_int64, C.c_int64,
                                               C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def hip_lib():
    """The device C ABI (include/sanafe_hip.h), for callers that drive it directly."""
    lib()
    L = C.CDLL(os.path.join(HERE, "csrc", "libsanafe_hip.so"))
    L.sanafe_hip_last_error.restype = C.c_char_p
    L.sanafe_hip_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.sanafe_hip_read_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.sanafe_hip_step.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]
    L.sanafe_hip_synchronize.argtypes = [C.c_void_p]
    L.sanafe_hip_stream.argtypes = [C.c_void_p]
    L.sanafe_hip_stream.restype = C.c_void_p
    L.sanafe_hip_read_core_delays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.sanafe_hip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.sanafe_hip_export_spikes.argtypes = [C.c_void_p, C.c_void_p]
    L.sanafe_hip_import_spikes.argtypes = [C.c_void_p, C.c_void_p]
    return L


TIMING = {"simple": 0, "detailed": 1, "cycle": 2}


class MappedNeuronRef:
    """``chip.mapped_neuron_groups[name][i]`` (src/pymodule.cpp:1174-1192)."""

    def __init__(self, chip, gid):
        self._chip, self._gid = chip, gid

    def set_model_attributes(self, model_attributes=None, soma_attributes=None, dendrite_attributes=None):
        """Older spelling some scripts use (scripts/computer2026/combined.py:297)."""
        return self.set_attributes(model_attributes, soma_attributes, dendrite_attributes)

    def set_attributes(self, model_attributes=None, soma_attributes=None, dendrite_attributes=None, log_spikes=None):
        """MappedNeuron::set_attributes (src/mapped.cpp:113-166): every attribute goes to the neuron's soma unit as at
        load().  The accumulator dendrites have no per-neuron attributes, so ``dendrite_attributes`` and the dendrite
        copy of ``model_attributes`` change nothing there, as in the reference; the constants of a `taps` dendrite
        cannot be changed after load() on this backend."""
        frozen = {"taps", "time_constants", "space_constants"}
        hit = frozen & (set(model_attributes or {}) | set(dendrite_attributes or {}))
        if hit:
            raise NotImplementedError("`taps` dendrite attributes cannot change after load() on the MI355X backend: %s"
                                      % sorted(hit))
        attrs = dict(model_attributes or {})
        attrs.update(soma_attributes or {})
        for key, value in attrs.items():
            t, num, sval, lst = D.py_to_attr(value)
            if t == D.ATTR_LIST:
                self._chip._set_attribute_list(self._gid, key, lst)
            else:
                self._chip._set_attribute(self._gid, key, t, num, sval)
        if log_spikes is not None:
            self._chip._log_spikes[self._gid] = bool(log_spikes)


class SpikingChip:
    """Drop-in for ``sanafe.SpikingChip`` on one MI355X (or one rank of a tile-sharded run)."""

    def __init__(self, arch: D.Architecture, device=0, n_ranks=1, rank=0):
        self._L = lib()
        self.arch = arch
        self._device, self._n_ranks, self._rank = device, n_ranks, rank
        self._h = None
        self._net = None
        self._nets = []
        self._built = None
        self.total_timesteps = 0

    # -- SpikingChip::load (src/chip.cpp:129-138); Python default overwrite=False ---------------
    def load(self, net: D.Network, overwrite=False):
        """SpikingChip::load (src/chip.cpp:129-138).  ``overwrite=False`` on a programmed chip ADDS the network: its
        groups are mapped after the ones already there (neuron ids, mapping order and per-core offsets continue).
        The chip is re-lowered from the combined description, so this is supported until the first timestep has
        been simulated (the reference would also carry the running state of the first network over)."""
        if self._h is not None and not overwrite:
            if self.total_timesteps > 0:
                raise NotImplementedError("UnsupportedError: load(net, overwrite=False) after timesteps have been simulated "
                                          "(the state of the programmed network cannot be carried into the re-lowered chip)")
            if cpp is None or not isinstance(net, cpp.Network) or not all(isinstance(n, cpp.Network) for n in self._nets):
                raise NotImplementedError("UnsupportedError: adding a network to a programmed chip needs sanafecpp_amd networks")
            merged = cpp.Network(self._nets[0].name)
            for n in self._nets + [net]:
                merged.absorb(n)
            self._nets.append(net)
            net = merged
        else:
            self._nets = [net]
        self._free()
        self._net = net
        self._built = _Lowered(self.arch, net)
        h = C.c_void_p()
        rc = self._L.sanafe_chip_create(self._built.address, self._device, self._n_ranks, self._rank, C.byref(h))
        if rc != 0:
            msg = self._L.sanafe_last_error().decode()
            if msg.startswith("HardwareMappingError"):
                raise D.HardwareMappingError(msg)
            if msg.startswith("UnsupportedError"):
                raise NotImplementedError(msg)
            raise RuntimeError(msg)
        self._h = h
        self.n_neurons = self._built.n_neurons
        self._log_spikes = self._built.log_spikes.copy()
        self._log_potential = self._built.log_potential.copy()
        # trace order: groups lexicographically by name, neurons by offset (std::map, src/chip.cpp:1616-1629)
        order = []
        for name in sorted(self._built.groups):
            base, count = self._built.groups[name]
            order.append(np.arange(base, base + count))
        self._trace_order = np.concatenate(order) if order else np.zeros(0, np.int64)
        self._gid_label = {}
        self.total_timesteps = 0

    @property
    def mapped_neuron_groups(self):
        return {name: [MappedNeuronRef(self, base + i) for i in range(count)]
                for name, (base, count) in self._built.groups.items()}

    def info(self):
        i = ChipInfo()
        self._check(self._L.sanafe_chip_get_info(self._h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in i._fields_}

    def _check(self, rc):
        if rc != 0:
            msg = self._L.sanafe_last_error().decode()
            if msg.startswith("UnsupportedError"):
                raise NotImplementedError(msg)
            raise RuntimeError(msg)

    def _free(self):
        if self._h is not None:
            self._L.sanafe_chip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def _set_bias(self, gids, values):
        g = np.ascontiguousarray(gids, dtype=np.int64)
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self._L.sanafe_chip_set_bias(self._h, len(g), g.ctypes.data, v.ctypes.data))

    def set_bias(self, group, values):
        """Vectorised MappedNeuron.set_attributes(model_attributes={'bias': b}) for a whole group (DVS frames)."""
        base, count = self._built.groups[str(group)]
        self._set_bias(np.arange(base, base + count), np.asarray(values, dtype=np.float64))

    # -- raw accessors (desc order) ------------------------------------------------------------
    def status(self):
        out = np.zeros(self.n_neurons, dtype=np.uint8)
        self._check(self._L.sanafe_chip_get_status(self._h, out.ctypes.data))
        return out

    def potentials(self):
        out = np.zeros(self.n_neurons, dtype=np.float64)
        self._check(self._L.sanafe_chip_get_potentials(self._h, out.ctypes.data))
        return out

    def input_currents(self):
        out = np.zeros(self.n_neurons, dtype=np.float64)
        self._check(self._L.sanafe_chip_get_input_current(self._h, out.ctypes.data))
        return out

    def step_totals(self, first, count):
        out = np.zeros(count, dtype=TOTALS_DTYPE)
        self._check(self._L.sanafe_chip_get_step_totals(self._h, first, count, out.ctypes.data))
        return out

    def perf_columns(self):
        """Names of the optional perf-trace columns (tiles / cores / units with log_energy / log_latency)."""
        n = self._L.sanafe_chip_perf_columns(self._h, None, 0)
        if n <= 0:
            return []
        buf = C.create_string_buffer(512 * n)
        self._L.sanafe_chip_perf_columns(self._h, buf, len(buf))
        return [b.decode() for b in buf.raw.split(b"\0")[:n]]

    def step_optional(self, first, count):
        """[count, len(perf_columns())] values of the optional perf-trace columns of recorded steps."""
        out = np.zeros((count, len(self.perf_columns())), dtype=np.float64)
        if out.size:
            self._check(self._L.sanafe_chip_get_step_optional(self._h, first, count, out.ctypes.data))
        return out

    def step_fired(self, index):
        out = np.zeros(self.n_neurons, dtype=np.uint8)
        self._check(self._L.sanafe_chip_get_step_fired(self._h, index, out.ctypes.data))
        return out

    def step_messages(self, index):
        n = self._L.sanafe_chip_get_step_messages(self._h, index, None, 0)
        if n < 0:
            raise RuntimeError("messages of step %d were not recorded (detailed timing + traces only)" % index)
        out = np.zeros(n, dtype=MSG_DTYPE)
        if n:
            self._L.sanafe_chip_get_step_messages(self._h, index, out.ctypes.data, n)
        return out

    RECORD_STEPS, RECORD_MESSAGES, RECORD_STATE = 1, 4, 8  # include/sanafe_host.h

    def set_state_log(self, potential_gids, current_gids=()):
        """Neurons whose potential / LIF input current every recorded step keeps (sampled on the device)."""
        pv = np.ascontiguousarray(potential_gids, dtype=np.int64)
        pu = np.ascontiguousarray(current_gids, dtype=np.int64)
        self._check(self._L.sanafe_chip_set_state_log(self._h, len(pv), pv.ctypes.data if len(pv) else None, len(pu),
                                                      pu.ctypes.data if len(pu) else None))
        self._state_row = len(pv) + len(pu)

    def step_state(self, first, count):
        out = np.zeros((count, self._state_row), dtype=np.float64)
        if out.size:
            self._check(self._L.sanafe_chip_get_step_state(self._h, first, count, out.ctypes.data))
        return out

    def run(self, timesteps, timing_model="simple", record=False, messages=False, state=False):
        """One sanafe_chip_sim call; returns the raw totals dict.  ``record`` keeps the per-step totals and spike
        lists, ``messages`` also every step's messages (step_messages), ``state`` the potentials / currents of the
        neurons given to set_state_log (step_state)."""
        t = Totals()
        flags = (self.RECORD_STEPS if record or messages or state else 0) | (self.RECORD_MESSAGES if messages else 0) | \
                (self.RECORD_STATE if state else 0)
        try:
            self._check(self._L.sanafe_chip_sim(self._h, int(timesteps), TIMING[timing_model], flags, C.byref(t)))
        finally:
            self.total_timesteps = int(self._L.sanafe_chip_total_timesteps(self._h))  # the C side is the one counter
        return t.as_dict()

    # -- SpikingChip.sim (src/pymodule.cpp:549-706, 1198-1208) ---------------------------------
    def sim(self, timesteps=1, timing_model="detailed", processing_threads=0, scheduler_threads=0, spike_trace=None,
            potential_trace=None, neuron_trace=None, perf_trace=None, message_trace=None, write_trace_headers=True):
        if self._h is None:
            raise RuntimeError("no network loaded")
        if timing_model not in TIMING:
            timing_model = "detailed"  # parse_timing_model falls back to detailed, src/chip.cpp:1833-1858
        self._check(self._L.sanafe_chip_set_scheduler_threads(self._h, int(scheduler_threads)))
        start = self.total_timesteps + 1
        want_steps = any(t is not None and t is not False for t in (spike_trace, perf_trace, message_trace))
        want_state = any(t is not None and t is not False for t in (potential_trace, neuron_trace))
        spikes, pots, ntraces, perf, msgs = [], [], {}, None, []
        opt_names = self.perf_columns() if perf_trace else []
        opt_rows = []
        if timing_model == "cycle":
            raise NotImplementedError("UnsupportedError: the cycle-accurate (Booksim2) timing model is out of scope")
        # One sanafe_chip_sim call whatever is traced: potentials and model traces are sampled on the device right after
        # each neuron update (src/pytrace.cpp:190-222 samples them per step) and fetched in bulk afterwards.
        pot_gids = self._trace_order[self._log_potential[self._trace_order]] if potential_trace else np.zeros(0, np.int64)
        cur_gids = self._trace_order if neuron_trace else np.zeros(0, np.int64)
        want_state = want_state and (len(pot_gids) + len(cur_gids)) > 0
        if want_state:
            self.set_state_log(pot_gids, cur_gids)
        tot = self.run(timesteps, timing_model, record=want_steps or want_state, messages=bool(message_trace), state=want_state)
        steps = self.step_totals(0, timesteps) if (want_steps or want_state) and timesteps > 0 else None
        if opt_names and timesteps > 0:
            opt_rows = list(self.step_optional(0, timesteps))
        fired = [self.step_fired(i) for i in range(timesteps)] if spike_trace else []
        if message_trace:
            msgs = [self.step_messages(i) for i in range(timesteps)]
        if potential_trace or neuron_trace:
            state = self.step_state(0, timesteps) if want_state else np.zeros((timesteps, 0))
            if potential_trace:
                pots = state[:, :len(pot_gids)].tolist()
            if neuron_trace:
                ntraces["u"] = state[:, len(pot_gids):].tolist()
        result = {
            "timestep_start": start, "timesteps_executed": timesteps,
            "energy": {"total": tot["total_energy"], "synapse": tot["synapse_energy"], "dendrite": tot["dendrite_energy"],
                       "soma": tot["soma_energy"], "network": tot["network_energy"]},
            "sim_time": tot["sim_time"], "spikes": tot["spikes"], "packets_sent": tot["packets_sent"],
            "neurons_updated": tot["neurons_updated"], "neurons_fired": tot["neurons_fired"],
        }
        if spike_trace:
            names = self._labels()
            for f in fired:
                sel = self._trace_order[(f[self._trace_order] != 0) & self._log_spikes[self._trace_order]]
                spikes.append([names[int(g)] for g in sel])
        if perf_trace and steps is not None:
            perf = {"timestep": [int(start + i) for i in range(timesteps)],
                    "fired": steps["neurons_fired"].tolist(), "updated": steps["neurons_updated"].tolist(),
                    "packets": steps["packets_sent"].tolist(), "hops": steps["total_hops"].tolist(),
                    "spikes": steps["spikes"].tolist(), "sim_time": steps["sim_time"].tolist(),
                    "synapse_energy": steps["synapse_energy"].tolist(), "dendrite_energy": steps["dendrite_energy"].tolist(),
                    "soma_energy": steps["soma_energy"].tolist(), "network_energy": steps["network_energy"].tolist(),
                    "total_energy": steps["total_energy"].tolist()}
            # optional columns, in the reference's std::map order (src/chip.cpp:1541-1579, src/pytrace.hpp:249-258)
            for k, name in enumerate(opt_names):
                perf[name] = [float(row[k]) for row in opt_rows]
        result["spike_trace"] = spikes if spike_trace else None
        result["potential_trace"] = pots if potential_trace else None
        result["neuron_trace"] = ntraces if neuron_trace else None
        result["perf_trace"] = perf
        result["message_trace"] = None
        for tr, key in ((spike_trace, "spike_trace"), (potential_trace, "potential_trace"), (perf_trace, "perf_trace"),
                        (message_trace, "message_trace")):
            if isinstance(tr, str) or hasattr(tr, "write"):
                self._write_trace(tr, key, msgs if key == "message_trace" else result[key], start, write_trace_headers)
                result[key] = None
            elif key == "message_trace" and tr:
                result[key] = [self._message_dicts(m) for m in msgs]
            elif key == "perf_trace" and result[key] is not None:
                # the in-memory map has no "packets" entry (timestep_data_to_map, src/pytrace.cpp:55-74); the CSV has
                result[key] = {k: v for k, v in result[key].items() if k != "packets"}
        return result

    def _labels(self):
        if not self._gid_label:
            for name, (base, count) in self._built.groups.items():
                for i in range(count):
                    self._gid_label[base + i] = (name, i)  # NeuronAddress(group_name, neuron_offset)
        return self._gid_label

    def _message_dicts(self, arr):
        """In-memory message trace rows: exactly the 26 keys of message_to_dict (src/pytrace.cpp:17-53), sorted by
        plain mid, so placeholders (mid -1) come first (src/pytrace.hpp:336-339)."""
        labels = self._labels()
        rows = []
        for m in np.sort(arr, order="mid", kind="stable"):
            g, o = labels.get(int(m["src_neuron"]), ("invalid", 0))
            rows.append({
                "generation_delay": float(m["generation_delay"]), "network_delay": float(m["network_delay"]),
                "processing_delay": float(m["processing_delay"]), "blocking_delay": float(m["blocking_delay"]),
                "send_timestamp": float(m["sent_timestamp"]), "received_timestamp": float(m["received_timestamp"]),
                "processed_timestamp": float(m["processed_timestamp"]),
                "timestep": int(m["timestep"]), "mid": int(m["mid"]), "spikes": int(m["spikes"]), "hops": int(m["hops"]),
                "src_neuron_offset": int(o), "src_neuron_group_id": g,
                "src_x": int(m["src_x"]), "dest_x": int(m["dest_x"]), "src_y": int(m["src_y"]), "dest_y": int(m["dest_y"]),
                "src_tile_id": int(m["src_tile"]), "src_core_id": int(m["src_core_id"]),
                "src_core_offset": int(m["src_core_offset"]), "dest_tile_id": int(m["dest_tile"]),
                "dest_core_id": int(m["dest_core_id"]), "dest_core_offset": int(m["dest_core_offset"]),
                "dest_axon_hw": 0, "dest_axon_id": int(m["dest_axon_id"]), "placeholder": bool(m["placeholder"]),
            })
        return rows

    def _write_trace(self, target, key, data, start, headers):
        """CSV traces with the reference's column layout (src/chip.cpp:1447-1764)."""
        close = False
        f = target
        if isinstance(target, str):
            f = open(target, "w" if headers else "a")
            close = True
        try:
            names = self._labels()
            if key == "spike_trace":
                if headers:
                    f.write("neuron,timestep\n")
                for i, row in enumerate(data or []):
                    for (g, o) in row:
                        f.write("%s.%d,%d\n" % (g, o, start + i))
            elif key == "potential_trace":
                sel = self._trace_order[self._log_potential[self._trace_order]]
                if headers:
                    f.write("timestep," + "".join("neuron %s.%d," % names[int(g)] for g in sel) + "\n")
                for i, row in enumerate(data or []):
                    if row:
                        f.write("%d," % (start + i) + "".join("%g," % v for v in row) + "\n")
            elif key == "perf_trace":
                cols = ["timestep", "fired", "updated", "packets", "hops", "spikes", "sim_time", "synapse_energy",
                        "dendrite_energy", "soma_energy", "network_energy", "total_energy"] + self.perf_columns()
                if headers:
                    f.write(",".join(cols) + "\n")
                for i in range(len(data["timestep"]) if data else 0):
                    f.write(",".join(("%d" % data[c][i]) if c in cols[:6] else ("%e" % data[c][i]) for c in cols) + "\n")
            elif key == "message_trace":
                cols = ["timestep", "mid", "src_neuron", "src_hw", "dest_hw", "hops", "spikes", "send_timestamp",
                        "received_timestamp", "processed_timestamp", "generation_delay", "processing_delay",
                        "network_delay", "blocking_delay", "min_hop_delay", "messages_along_route"]
                if headers:
                    f.write(",".join(cols) + "\n")
                for step in data or []:  # raw records (MSG_DTYPE) of one timestep
                    # the CSV writer sorts by mid with placeholders LAST (src/message.cpp:70-91)
                    order = sorted(range(len(step)), key=lambda i: (step["mid"][i] < 0, step["mid"][i]))
                    for i in order:
                        m = step[i]
                        g, o = names.get(int(m["src_neuron"]), ("invalid", 0))
                        csv = {"timestep": int(m["timestep"]), "mid": int(m["mid"]), "src_neuron": "%s.%d" % (g, o),
                               "src_hw": "%d.%d" % (m["src_tile"], m["src_core_offset"]),
                               "dest_hw": "x.x" if m["placeholder"] else "%d.%d" % (m["dest_tile"], m["dest_core_offset"]),
                               "hops": int(m["hops"]), "spikes": int(m["spikes"]), "send_timestamp": float(m["sent_timestamp"])}
                        for k in MSG_DBL:
                            csv.setdefault(k, float(m[k]))
                        f.write(",".join(("%g" % csv[c]) if isinstance(csv[c], float) else str(csv[c]) for c in cols) + "\n")
        finally:
            if close:
                f.close()

    def reset(self):
        self._check(self._L.sanafe_chip_reset(self._h))

    def get_power(self):
        return self._L.sanafe_chip_get_power(self._h)

    # -- tile-sharded chips: the per-step spike exchange lives in the host library (host/comm.cpp) ----------
    @staticmethod
    def comm_unique_id():
        """Rank 0: the 128-byte RCCL id every rank passes to ``comm_init_rccl`` (ncclGetUniqueId)."""
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        if lib().sanafe_comm_unique_id(buf) != 0:
            raise RuntimeError(lib().sanafe_last_error().decode())
        return bytes(buf)

    def comm_init_rccl(self, comm_id):
        """Collective over all ranks (ncclCommInitRank): afterwards ``sim()`` exchanges spikes over RCCL."""
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        self._check(self._L.sanafe_chip_comm_init_rccl(self._h, buf))

    def comm_init_callback(self, allgather):
        """``allgather(send: bytes-like numpy uint8 array) -> numpy uint8 array [n_ranks, len(send)]``: a blocking
        all-gather over host memory (tests on one GPU, MPI-style bindings)."""
        n_ranks = self._n_ranks

        def _cb(_ctx, send, nbytes, recv):
            try:
                src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                out = np.ascontiguousarray(allgather(src), dtype=np.uint8).reshape(n_ranks, nbytes)
                C.memmove(recv, out.ctypes.data, n_ranks * nbytes)
                return 0
            except Exception as e:  # the C side reports a failed exchange
                self._callback_error = e
                return -1

        self._allgather_cb = ALLGATHER_FN(_cb)  # keep the thunk alive as long as the chip
        self._check(self._L.sanafe_chip_comm_init_callback(self._h, self._allgather_cb, None))

    def comm_init_torch(self, dist, backend="rccl"):
        """Sets the exchange up from an initialised ``torch.distributed`` process group: ``rccl`` ships the unique
        id through the group's store and creates the library's own RCCL communicator; ``host`` gathers through
        the group (gloo) in host memory."""
        if backend == "rccl":
            box = [self.comm_unique_id() if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(box, src=0)
            self.comm_init_rccl(box[0])
            return
        import torch

        def allgather(send):
            t = torch.from_numpy(np.array(send, copy=True))
            out = torch.empty(dist.get_world_size() * t.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(out, t)
            return out.numpy()

        self.comm_init_callback(allgather)

    # -- split step for callers that drive the exchange themselves -------------------------------
    def _set_attribute(self, gid, key, attr_type, num, sval):
        L = self._L
        L.sanafe_chip_set_attribute.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_int, C.c_double, C.c_char_p]
        self._check(L.sanafe_chip_set_attribute(self._h, int(gid), str(key).encode(), int(attr_type), float(num),
                                                None if sval is None else str(sval).encode()))

    def _set_attribute_list(self, gid, key, values):
        L = self._L
        arr = np.ascontiguousarray(np.asarray(values, dtype=np.float64))
        L.sanafe_chip_set_attribute_list.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_void_p, C.c_int64]
        self._check(L.sanafe_chip_set_attribute_list(self._h, int(gid), str(key).encode(), arr.ctypes.data if len(arr) else None,
                                                     len(arr)))

    def device_layout(self):
        """(synapse format, compact axon slices) the device image was packed with (sanafe_hip_get_layout)."""
        H = hip_lib()
        fmt, n = C.c_int(), C.c_uint32()
        H.sanafe_hip_get_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
        if H.sanafe_hip_get_layout(self.device_handle(), C.byref(fmt), C.byref(n)) != 0:
            raise RuntimeError(H.sanafe_hip_last_error().decode())
        return {"syn_format": fmt.value, "n_compact_slices": n.value}

    def step_neurons(self):
        self._check(self._L.sanafe_chip_step_neurons(self._h))

    def step_deliver(self, timing_model="simple"):
        self._check(self._L.sanafe_chip_step_deliver(self._h, TIMING[timing_model]))
        self.total_timesteps += 1

    def spike_buffers(self):
        lp, gp = C.c_void_p(), C.c_void_p()
        lb, gb, off = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._L.sanafe_chip_spike_buffers(self._h, C.byref(lp), C.byref(lb), C.byref(gp), C.byref(gb),
                                                      C.byref(off)))
        return dict(local_ptr=lp.value, local_bytes=lb.value, global_ptr=gp.value, global_bytes=gb.value,
                    local_offset_bytes=off.value)

    def synchronize(self):
        self._check(self._L.sanafe_chip_synchronize(self._h))

    def read_totals(self):
        t = Totals()
        self._check(self._L.sanafe_chip_read_totals(self._h, C.byref(t)))
        return t.as_dict()

    def device_handle(self):
        return self._L.sanafe_chip_device(self._h)


# --------------------------------------------------------------------------------------------
# The lowered device image (include/sanafe_hip.h), for inspection and host-side checks
# --------------------------------------------------------------------------------------------
class SomaClass(C.Structure):
    _fields_ = [("threshold", C.c_double), ("reverse_threshold", C.c_double), ("reset", C.c_double),
                ("reverse_reset", C.c_double), ("leak_decay", C.c_double), ("input_decay", C.c_double),
                ("refractory_delay", C.c_int32), ("reset_mode", C.c_uint8), ("reverse_reset_mode", C.c_uint8),
                ("force_update", C.c_uint8), ("leak_towards_zero", C.c_uint8)]


class CostClass(C.Structure):
    _fields_ = [("soma_energy", C.c_double * 3), ("soma_latency", C.c_double * 3), ("dendrite_energy", C.c_double),
                ("dendrite_latency", C.c_double)]


def _p(t):
    return C.POINTER(t)


class HipImage(C.Structure):
    _fields_ = [
        ("n_cores", C.c_uint32), ("n_slots", C.c_uint32), ("n_soma_classes", C.c_uint32), ("n_cost_classes", C.c_uint32),
        ("ring_slots", C.c_uint32), ("n_slices", C.c_uint32), ("n_axons", C.c_uint64), ("n_synapses", C.c_uint64),
        ("n_input", C.c_uint32), ("n_train_words", C.c_uint64), ("slot_offset", C.c_uint32),
        ("n_global_slots", C.c_uint32), ("sync_delay", C.c_double),
        ("core_nbase", _p(C.c_uint32)), ("core_ncount", _p(C.c_uint32)), ("core_axon_out_latency", _p(C.c_double)),
        ("soma_classes", _p(SomaClass)), ("cost_classes", _p(CostClass)),
        ("slot_cls", _p(C.c_uint32)), ("slot_bias", _p(C.c_double)), ("slot_v0", _p(C.c_double)),
        ("slot_aux", _p(C.c_uint32)), ("slot_packets", _p(C.c_uint32)), ("slot_hops", _p(C.c_uint32)),
        ("slot_events", _p(C.c_uint32)), ("slot_e_net", _p(C.c_double)), ("slot_e_syn", _p(C.c_double)),
        ("slot_e_dend", _p(C.c_double)),
        ("in_train_beg", _p(C.c_uint32)), ("in_train_len", _p(C.c_uint32)), ("in_rate_period", _p(C.c_int64)),
        ("in_train_bits", _p(C.c_uint32)),
        ("n_taps", C.c_uint32), ("tap_slot", _p(C.c_uint32)), ("tap_count", _p(C.c_uint32)), ("tap_tc", _p(C.c_double)),
        ("tap_sc", _p(C.c_double)),
        ("n_ext", C.c_uint32), ("slot_ext", _p(C.c_uint32)),
        ("slice_core", _p(C.c_uint32)), ("slice_axon_beg", _p(C.c_uint64)), ("slice_axon_end", _p(C.c_uint64)),
        ("core_syn_base", _p(C.c_uint64)), ("core_axon_in_latency", _p(C.c_double)),
        ("ax_pre", _p(C.c_uint32)), ("ax_syn_beg", _p(C.c_uint32)), ("ax_nsyn", _p(C.c_uint32)),
        ("ax_proc_delay", _p(C.c_double)), ("ax_lat_class", _p(C.c_uint8)), ("lat_class_per_event", _p(C.c_double)),
        ("syn_meta", _p(C.c_uint32)), ("syn_weight", _p(C.c_double)),
    ]


def map_only(arch, net, n_ranks=1, rank=0, ext_steps=0):
    """Maps and lowers without touching a device; returns (image dict of numpy arrays, slot_of_neuron).
    The arrays are copies, so they outlive the temporary chip.  ``ext_steps`` > 0 also generates that many rows
    of the external value streams (``ext_rows``: [ext_steps, n_ext] int32)."""
    L = lib()
    L.sanafe_chip_get_image.argtypes = [C.c_void_p, C.POINTER(HipImage)]
    L.sanafe_chip_get_slot_map.argtypes = [C.c_void_p, C.c_void_p]
    built = _Lowered(arch, net)
    h = C.c_void_p()
    if L.sanafe_chip_create(built.address, -1, n_ranks, rank, C.byref(h)) != 0:
        msg = L.sanafe_last_error().decode()
        if msg.startswith("HardwareMappingError"):
            raise D.HardwareMappingError(msg)
        if msg.startswith("UnsupportedError"):
            raise NotImplementedError(msg)
        raise RuntimeError(msg)
    try:
        im = HipImage()
        if L.sanafe_chip_get_image(h, C.byref(im)) != 0:
            raise RuntimeError(L.sanafe_last_error().decode())
        counts = {"core_nbase": im.n_cores, "core_ncount": im.n_cores, "core_axon_out_latency": im.n_cores,
                  "core_syn_base": im.n_cores, "core_axon_in_latency": im.n_cores,
                  "slice_core": im.n_slices, "slice_axon_beg": im.n_slices, "slice_axon_end": im.n_slices,
                  "ax_pre": im.n_axons, "ax_syn_beg": im.n_axons, "ax_nsyn": im.n_axons, "ax_proc_delay": im.n_axons, "ax_lat_class": im.n_axons, "lat_class_per_event": 255,
                  "syn_meta": im.n_synapses, "syn_weight": im.n_synapses,
                  "in_train_beg": im.n_input, "in_train_len": im.n_input, "in_rate_period": im.n_input,
                  "in_train_bits": im.n_train_words}
        for n in ("slot_cls", "slot_bias", "slot_v0", "slot_aux", "slot_packets", "slot_hops", "slot_events",
                  "slot_e_net", "slot_e_syn", "slot_e_dend"):
            counts[n] = im.n_slots
        counts["slot_ext"] = im.n_slots if im.n_ext else 0
        counts.update(tap_slot=im.n_taps, tap_count=im.n_taps, tap_tc=im.n_taps * 8, tap_sc=im.n_taps * 8)
        out = {}
        for n, _t in HipImage._fields_:
            v = getattr(im, n)
            if n in counts:
                out[n] = np.ctypeslib.as_array(v, shape=(counts[n],)).copy() if counts[n] else np.zeros(0)
            elif n == "soma_classes":
                out[n] = [{f: getattr(v[i], f) for f, _ in SomaClass._fields_} for i in range(im.n_soma_classes)]
            elif n == "cost_classes":
                out[n] = [dict(soma_energy=list(v[i].soma_energy), soma_latency=list(v[i].soma_latency),
                               dendrite_energy=v[i].dendrite_energy, dendrite_latency=v[i].dendrite_latency)
                          for i in range(im.n_cost_classes)]
            else:
                out[n] = v
        slot_of = np.zeros(built.n_neurons, dtype=np.uint32)
        L.sanafe_chip_get_slot_map(h, slot_of.ctypes.data)
        rows = np.zeros((int(ext_steps), int(im.n_ext)), dtype=np.int32)
        if rows.size:
            L.sanafe_chip_generate_ext.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
            if L.sanafe_chip_generate_ext(h, int(ext_steps), rows.ctypes.data) != 0:
                raise RuntimeError(L.sanafe_last_error().decode())
        out["ext_rows"] = rows
        return out, slot_of
    finally:
        L.sanafe_chip_destroy(h)


def generate_random_edges(n_neurons, out_degree, seed=1, n_threads=None, src_base=0, dst_base=0, shard=None):
    """(src, dst, weight) of the synthetic random SNN of the benchmark configs.
    ``shard=(lo, hi)`` keeps only the edges with source or destination in [lo, hi)."""
    n_threads = n_threads or min(32, os.cpu_count() or 1)
    if shard is not None:
        L = lib()
        L.sanafe_generate_random_edges_sharded.argtypes = [C.c_int64, C.c_int64, C.c_uint64, C.c_int, C.c_int64, C.c_int64,
                                                           C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.sanafe_edge_set_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sanafe_edge_set_free.argtypes = [C.c_void_p]
        h, cnt = C.c_void_p(), C.c_int64()
        if L.sanafe_generate_random_edges_sharded(n_neurons, out_degree, seed, n_threads, shard[0], shard[1],
                                                  C.byref(h), C.byref(cnt)) != 0:
            raise RuntimeError(L.sanafe_last_error().decode())
        src = np.empty(cnt.value, dtype=np.int64)
        dst = np.empty(cnt.value, dtype=np.int64)
        w = np.empty(cnt.value, dtype=np.float64)
        L.sanafe_edge_set_copy(h, src.ctypes.data, dst.ctypes.data, w.ctypes.data)
        L.sanafe_edge_set_free(h)
        return src, dst, w
    e = int(n_neurons) * int(out_degree)
    src = np.empty(e, dtype=np.int64)
    dst = np.empty(e, dtype=np.int64)
    w = np.empty(e, dtype=np.float64)
    if lib().sanafe_generate_random_edges(n_neurons, out_degree, seed, n_threads, src_base, dst_base, src.ctypes.data,
                                          dst.ctypes.data, w.ctypes.data) != 0:
        raise RuntimeError(lib().sanafe_last_error().decode())
    return src, dst, w
