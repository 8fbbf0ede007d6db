"""Drive one pipeline-unit model through either the reference build (oracle/_ref,
prefix ``refm_``) or the oracle's restatement (prefix ``oracle_unit_``)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Result(C.Structure):
    _fields_ = [("has_current", C.c_int), ("current", C.c_double), ("status", C.c_int), ("has_energy", C.c_int),
                ("energy", C.c_double), ("has_latency", C.c_int), ("latency", C.c_double)]

    def key(self):
        cur = np.float64(self.current).tobytes().hex() if self.has_current else None
        return (cur, self.status, self.has_energy, self.has_latency)


class UnitLib:
    def __init__(self, path, prefix):
        self.L = C.CDLL(path)
        self.p = prefix
        f = lambda n: getattr(self.L, prefix + n)  # noqa: E731
        if prefix == "refm_":
            f("create").argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        else:
            f("create").argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        f("create").restype = C.c_void_p
        f("destroy").argtypes = [C.c_void_p]
        f("set_attr_hw").argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_double, C.c_char_p, C.c_void_p, C.c_long,
                                     C.c_char_p, C.c_int]
        for n in ("set_attr_neuron", "set_attr_edge"):
            f(n).argtypes = [C.c_void_p, C.c_long, C.c_char_p, C.c_int, C.c_double, C.c_char_p, C.c_void_p, C.c_long,
                             C.c_char_p, C.c_int]
        f("update_syn").argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_long, C.POINTER(Result), C.c_char_p, C.c_int]
        f("update_dend").argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_double, C.c_int, C.c_long, C.c_long,
                                     C.POINTER(Result), C.c_char_p, C.c_int]
        f("update_soma").argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_double, C.c_long, C.POINTER(Result),
                                     C.c_char_p, C.c_int]
        f("get_potential").argtypes = [C.c_void_p, C.c_long]
        f("get_potential").restype = C.c_double
        f("get_trace").argtypes = [C.c_void_p, C.c_long, C.c_char_p, C.POINTER(C.c_double)]
        f("reset").argtypes = [C.c_void_p]

    def fn(self, n):
        return getattr(self.L, self.p + n)


class Unit:
    def __init__(self, lib: UnitLib, model, plugin=None):
        self.lib = lib
        err = C.create_string_buffer(512)
        if lib.p == "refm_":
            self.h = lib.fn("create")(model.encode(), plugin.encode() if plugin else None, err, 512)
        else:
            self.h = lib.fn("create")(model.encode(), err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.fn("destroy")(self.h)
            self.h = None

    @staticmethod
    def _conv(value):
        if isinstance(value, bool):
            return 0, float(value), None, None
        if isinstance(value, int):
            return 1, float(value), None, None
        if isinstance(value, float):
            return 2, value, None, None
        if isinstance(value, str):
            return 3, 0.0, value.encode(), None
        return 4, 0.0, None, np.asarray(value, dtype=np.float64)

    def _set(self, fname, addr, key, value):
        t, num, s, lst = self._conv(value)
        err = C.create_string_buffer(512)
        args = [self.h] + ([] if addr is None else [addr]) + [key.encode(), t, num, s,
                                                               lst.ctypes.data if lst is not None else None,
                                                               len(lst) if lst is not None else 0, err, 512]
        if self.lib.fn(fname)(*args) != 0:
            raise RuntimeError(err.value.decode())

    def set_hw(self, key, value):
        self._set("set_attr_hw", None, key, value)

    def set_neuron(self, addr, key, value):
        self._set("set_attr_neuron", addr, key, value)

    def set_edge(self, addr, key, value):
        self._set("set_attr_edge", addr, key, value)

    def _call(self, fname, *args):
        r = Result()
        err = C.create_string_buffer(512)
        if self.lib.fn(fname)(self.h, *args, C.byref(r), err, 512) != 0:
            raise RuntimeError(err.value.decode())
        return r

    def syn(self, addr, read, t):
        return self._call("update_syn", addr, int(read), t)

    def dend(self, naddr, cur, syn, t):
        return self._call("update_dend", naddr, cur is not None, 0.0 if cur is None else cur, syn is not None,
                          0 if syn is None else syn, t)

    def soma(self, naddr, cur, t):
        return self._call("update_soma", naddr, cur is not None, 0.0 if cur is None else cur, t)

    def potential(self, addr):
        return self.lib.fn("get_potential")(self.h, addr)

    def trace(self, addr, name):
        out = C.c_double()
        return out.value if self.lib.fn("get_trace")(self.h, addr, name.encode(), C.byref(out)) else None

    def reset(self):
        self.lib.fn("reset")(self.h)


def oracle_lib():
    return UnitLib(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_unit_")


def reference_lib():
    return UnitLib(os.path.join(ROOT, "oracle", "_ref", "libsanafe_ref_models.so"), "refm_")
