"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class OracleTs(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("timestep", "spike_count", "total_hops", "packets_sent", "neurons_updated",
                                         "neurons_fired", "n_messages")] + \
               [(n, C.c_double) for n in ("total_energy", "synapse_energy", "dendrite_energy", "soma_energy",
                                          "network_energy", "sim_time")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


MSG_INT = ("timestep", "mid", "src_neuron", "src_tile", "src_core_offset", "src_core_id", "dest_tile",
           "dest_core_offset", "dest_core_id", "dest_axon_id", "hops", "spikes", "placeholder",
           "src_x", "src_y", "dest_x", "dest_y")
MSG_DBL = ("generation_delay", "processing_delay", "network_delay", "blocking_delay", "min_hop_delay",
           "sent_timestamp", "received_timestamp", "processed_timestamp", "messages_along_route")
MSG_DTYPE = np.dtype([(n, np.int64) for n in MSG_INT] + [(n, np.float64) for n in MSG_DBL])

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_step.argtypes = [C.c_void_p, C.c_int, C.POINTER(OracleTs), C.c_char_p, C.c_int]
        L.oracle_get_status.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_get_potentials.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_get_trace.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        L.oracle_get_messages.restype = C.c_int64
        L.oracle_get_messages.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.oracle_reset.argtypes = [C.c_void_p]
        L.oracle_set_neuron_attr.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_int, C.c_double, C.c_char_p,
                                             C.c_void_p, C.c_int64, C.c_int, C.c_char_p, C.c_int]
        L.oracle_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.oracle_mapped_tiles.restype = C.c_int64
        L.oracle_mapped_tiles.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class OracleChip:
    SIMPLE, DETAILED = 0, 1

    def __init__(self, built_desc):
        self._built = built_desc  # keeps the buffers alive: the oracle borrows them
        if hasattr(built_desc, "address"):  # sanafecpp_amd.Desc (C++ front-end)
            address, self.n = built_desc.address, int(built_desc.n_neurons)
        else:  # description.BuiltDesc (Python twin)
            address, self.n = C.addressof(built_desc.desc), int(built_desc.desc.n_neurons)
        err = C.create_string_buffer(1024)
        self._h = lib().oracle_create(address, err, 1024)
        if not self._h:
            raise RuntimeError(err.value.decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_destroy(self._h)
            self._h = None

    def set_threads(self, n):
        """n > 1: OpenMP over cores like the reference (src/chip.cpp:629-632, 675-678); 1 = serial (the parity mode)."""
        lib().oracle_set_threads(self._h, int(n))

    def step(self, timing="simple"):
        ts = OracleTs()
        err = C.create_string_buffer(1024)
        tm = {"simple": 0, "detailed": 1}[timing]
        if lib().oracle_step(self._h, tm, C.byref(ts), err, 1024) != 0:
            raise RuntimeError(err.value.decode())
        return ts.as_dict()

    def status(self):
        out = np.zeros(self.n, dtype=np.uint8)
        lib().oracle_get_status(self._h, out.ctypes.data)
        return out

    def potentials(self):
        out = np.zeros(self.n, dtype=np.float64)
        lib().oracle_get_potentials(self._h, out.ctypes.data)
        return out

    def trace(self, name):
        out = np.zeros(self.n, dtype=np.float64)
        lib().oracle_get_trace(self._h, name.encode(), out.ctypes.data)
        return out

    def messages(self):
        n = lib().oracle_get_messages(self._h, None, 0)
        out = np.zeros(n, dtype=MSG_DTYPE)
        if n:
            lib().oracle_get_messages(self._h, out.ctypes.data, n)
        return out

    def optional_traces(self):
        """{column name: value} of the optional perf-trace columns of the last step."""
        L = lib()
        L.oracle_optional_traces.restype = C.c_int64
        L.oracle_optional_traces.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]
        n = L.oracle_optional_traces(self._h, None, 0, None, 0)
        names = C.create_string_buffer(256 * max(1, n))
        vals = np.zeros(max(1, n), dtype=np.float64)
        L.oracle_optional_traces(self._h, names, len(names), vals.ctypes.data, n)
        keys = names.raw.split(b"\0")[:n]
        return {k.decode(): float(v) for k, v in zip(keys, vals[:n])}

    def reset(self):
        lib().oracle_reset(self._h)

    def set_neuron_attr(self, gid, key, attr, fwd=7):
        t, num, s, lst = attr
        err = C.create_string_buffer(1024)
        arr = np.asarray(lst, dtype=np.float64) if lst is not None else None
        rc = lib().oracle_set_neuron_attr(self._h, gid, key.encode(), t, num, s.encode() if s else None,
                                          arr.ctypes.data if arr is not None else None,
                                          len(arr) if arr is not None else 0, fwd, err, 1024)
        if rc != 0:
            raise RuntimeError(err.value.decode())

    def mapped_tiles(self):
        return lib().oracle_mapped_tiles(self._h)
