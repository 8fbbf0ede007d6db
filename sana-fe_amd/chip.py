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

HERE = os.path.dirname(os.path.abspath(__file__))


class HardwareMappingError(RuntimeError):
    """The reference's exception for networks that do not fit the hardware (src/mapped.hpp:30-38)."""


try:  # C++17 / PyBind11 module: description objects + the compiled SpikingChip; built by `make -C sana-fe_amd`
    from . import sanafecpp_amd as cpp
except ImportError:  # pragma: no cover
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
            # any other front-end (the tests' pure-Python twin) lowers itself: lower_for_chip(arch) ->
            # (address of a sanafe_desc, [(group, base, count)], log_spikes, log_potential, keep-alive object)
            self.address, groups, ls, lp, self.handle = net.lower_for_chip(arch)
            self.n_neurons = int(len(ls))
            self.groups = {name: (int(base), int(count)) for name, base, count in groups}
            self.log_spikes, self.log_potential = np.asarray(ls).astype(bool), np.asarray(lp).astype(bool)


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
    L.sanafe_chip_wants_perf_columns.argtypes = [C.c_void_p]
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


class _GroupsView(dict):
    """{group name: (first neuron id, count)} of the programmed chip (kept for tests and helpers)."""


_Base = cpp.SpikingChip if cpp is not None else object


class SpikingChip(_Base):
    """Drop-in for ``sanafe.SpikingChip`` on one MI355X (or one rank of a tile-sharded run).

    The class itself is compiled C++ (PyBind11, host/pychip.cpp): ``load``, ``sim`` (GIL released, Ctrl-C polled
    between chunks, traces streamed), ``reset``, ``get_power``, ``mapped_neuron_groups[...][i].set_attributes``.
    This Python subclass only adds the diagnostics the tests and bench.py use (raw step records, state dumps, the
    exchange set-up) through the C API of libsanafe_host.so on the same chip handle, and the loader for networks
    built with the pure-Python twin of the description layer."""

    def __init__(self, arch, device=0, n_ranks=1, rank=0):
        if cpp is None:
            raise BackendMissingError("sanafecpp_amd is missing: build it with `make -C sana-fe_amd`")
        self._L = lib()
        super().__init__(arch, device, n_ranks, rank)
        self._device, self._n_ranks, self._rank = device, n_ranks, rank
        self._state_row = 0

    @property
    def _h(self):
        return C.c_void_p(self.handle) if self.handle else None

    # -- SpikingChip::load (src/chip.cpp:129-138); Python default overwrite=False ---------------
    def load(self, net, overwrite=False):
        if isinstance(net, cpp.Network):
            return super().load(net, overwrite)
        # a network from another front-end (the tests' pure-Python twin) lowers itself and hands the flat description over
        if self.handle and not overwrite:
            raise NotImplementedError("UnsupportedError: adding a network to a programmed chip needs sanafecpp_amd networks")
        address, groups, ls, lp, keep = net.lower_for_chip(self.arch)
        self.load_lowered(address, [(str(n), int(b), int(c)) for n, b, c in groups], np.asarray(ls, dtype=np.uint8),
                          np.asarray(lp, dtype=np.uint8), (keep, net, self.arch))

    # -- views the tests use -----------------------------------------------------------------------
    @property
    def _built(self):
        class _B:
            pass
        b = _B()
        b.groups = self.group_table()
        return b

    @property
    def _trace_order(self):
        return np.asarray(self.trace_order())

    @property
    def _log_spikes(self):
        return np.asarray(self.log_flags()[0]).astype(bool)

    @property
    def _log_potential(self):
        return np.asarray(self.log_flags()[1]).astype(bool)

    def _labels(self):
        return {base + i: (name, i) for name, (base, count) in self.group_table().items() for i in range(count)}

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

    def _set_bias(self, gids, values):
        g = np.ascontiguousarray(gids, dtype=np.int64)
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self._L.sanafe_chip_set_bias(self._h, len(g), g.ctypes.data, v.ctypes.data))

    def set_bias(self, group, values):
        """Vectorised MappedNeuron.set_attributes(model_attributes={'bias': b}) for a whole group (DVS frames)."""
        base, count = self.group_table()[str(group)]
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
        if self._n_ranks > 1 and self._L.sanafe_chip_wants_perf_columns(self._h):
            self.attach_whole()  # a rank only knows that columns are wanted: the whole-chip twin holds their plan
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
        if self._n_ranks > 1 and (timing_model == "detailed" or messages or (record and self._L.sanafe_chip_wants_perf_columns(self._h))):
            self.attach_whole()  # whole-chip host tables for the NoC schedule / message trace / optional perf columns of a sharded chip
        flags = (self.RECORD_STEPS if record or messages or state else 0) | (self.RECORD_MESSAGES if messages else 0) | \
                (self.RECORD_STATE if state else 0)
        self._check(self._L.sanafe_chip_sim(self._h, int(timesteps), TIMING[timing_model], flags, C.byref(t)))
        return t.as_dict()

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
        H.sanafe_hip_get_acc_shift.argtypes = [C.c_void_p]
        en, pushed = C.c_uint32(), C.c_uint32()
        H.sanafe_hip_get_push_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        H.sanafe_hip_get_push_info(self.device_handle(), C.byref(en), C.byref(pushed))
        H.sanafe_hip_get_bitmap_slices.argtypes = [C.c_void_p]
        H.sanafe_hip_get_sub_accumulators.argtypes = [C.c_void_p]
        ev = (C.c_uint64 * 11)()
        H.sanafe_hip_get_event_info.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        H.sanafe_hip_get_event_info(self.device_handle(), ev, 11)
        event = {"groups": int(ev[0]), "segments": int(ev[1]), "units": int(ev[2]), "words_per_block": ev[3] / 1000.0,
                 "lanes_per_block": int(ev[4]), "code_bits": int(ev[5]), "shift": int(ev[6]), "always": bool(ev[7]),
                 "max_events": int(ev[8]), "sparse_max_events": int(ev[9]), "sparse_steps": int(ev[10])} if ev[0] else None
        return {"syn_format": fmt.value, "n_compact_slices": n.value, "acc_shift": int(H.sanafe_hip_get_acc_shift(self.device_handle())),
                "event_layout": event, "msg_cores_on_device": int(self._msg_cores(H)),
                "push_enabled": bool(en.value), "push_only": en.value == 2, "pushed_steps": int(pushed.value),
                "n_bitmap_slices": int(H.sanafe_hip_get_bitmap_slices(self.device_handle())),
                "sub_accumulators": bool(H.sanafe_hip_get_sub_accumulators(self.device_handle()))}

    def _msg_cores(self, H):
        H.sanafe_hip_get_msg_cores.argtypes = [C.c_void_p]
        return H.sanafe_hip_get_msg_cores(self.device_handle())

    def step_neurons(self):
        self._check(self._L.sanafe_chip_step_neurons(self._h))

    def step_deliver(self, timing_model="simple"):
        self._check(self._L.sanafe_chip_step_deliver(self._h, TIMING[timing_model]))

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
        ("n_msg_cores", C.c_uint32), ("msg_core", _p(C.c_uint32)), ("msg_ax_beg", _p(C.c_uint32)), ("msg_ax_pre", _p(C.c_uint32)),
        ("msg_ax_nsyn", _p(C.c_uint32)), ("msg_syn_beg", _p(C.c_uint32)), ("msg_syn_post", _p(C.c_uint32)),
        ("msg_syn_weight", _p(C.c_double)), ("msg_costs", C.c_void_p),
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
            raise HardwareMappingError(msg)
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
        n_msg_ax = int(im.msg_ax_beg[im.n_msg_cores]) if im.n_msg_cores else 0
        n_msg_syn = int(im.msg_syn_beg[im.n_msg_cores]) if im.n_msg_cores else 0
        counts.update(msg_core=im.n_msg_cores, msg_ax_beg=im.n_msg_cores + 1 if im.n_msg_cores else 0, msg_ax_pre=n_msg_ax,
                      msg_ax_nsyn=n_msg_ax, msg_syn_beg=im.n_msg_cores + 1 if im.n_msg_cores else 0, msg_syn_post=n_msg_syn,
                      msg_syn_weight=n_msg_syn)
        out = {}
        for n, _t in HipImage._fields_:
            v = getattr(im, n)
            if n in counts:
                out[n] = np.ctypeslib.as_array(v, shape=(counts[n],)).copy() if counts[n] else np.zeros(0)
            elif n == "soma_classes":
                out[n] = [{f: getattr(v[i], f) for f, _ in SomaClass._fields_} for i in range(im.n_soma_classes)]
            elif n == "msg_costs":
                # sanafe_hip_msg_core_costs: axon_in_latency, synapse e / l, dendrite e / l, soma_energy[3], soma_latency[3]
                out[n] = (np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_double)), shape=(int(im.n_msg_cores), 11)).copy()
                          if im.n_msg_cores else np.zeros((0, 11)))
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


def generate_random_edges(n_neurons, out_degree, seed=1, n_threads=None, src_base=0, dst_base=0, shard=None, window=None):
    """(src, dst, weight) of the synthetic random SNN of the benchmark configs.
    ``shard=(lo, hi)`` keeps only the edges with source or destination in [lo, hi); ``window`` draws every neuron's
    targets from the `window` neurons centred on it (ids wrap) instead of from all of them."""
    n_threads = n_threads or min(32, os.cpu_count() or 1)
    if window is not None:
        L = lib()
        L.sanafe_generate_random_edges_windowed.argtypes = [C.c_int64, C.c_int64, C.c_uint64, C.c_int, C.c_int64, C.c_int64,
                                                            C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.sanafe_edge_set_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sanafe_edge_set_free.argtypes = [C.c_void_p]
        lo, hi = shard if shard is not None else (0, n_neurons)
        h, cnt = C.c_void_p(), C.c_int64()
        if L.sanafe_generate_random_edges_windowed(n_neurons, out_degree, seed, n_threads, int(window), lo, hi,
                                                   C.byref(h), C.byref(cnt)) != 0:
            raise RuntimeError(L.sanafe_last_error().decode())
        src = np.empty(cnt.value, dtype=np.int64)
        dst = np.empty(cnt.value, dtype=np.int64)
        w = np.empty(cnt.value, dtype=np.float64)
        L.sanafe_edge_set_copy(h, src.ctypes.data, dst.ctypes.data, w.ctypes.data)
        L.sanafe_edge_set_free(h)
        return src, dst, w
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
