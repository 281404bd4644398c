"""Host side of the C ABI's op-table replay (include/radar_depth_hip.h, rd_optable_*).

An execution plan (engine.LateFusionPlan, main.HipTrainStep) is a flat list of (name, C-ABI function, ctypes arguments).  The
step body of the reference's main.py:416-445 replays that list every iteration; OpTable marshals it ONCE into the library's
table form and a step becomes one rd_optable_run call (per graph-capturable piece) instead of ~500 ctypes calls from a Python
loop.  Arguments that are plan streams (the SAME c_void_p objects the plan rebinds in set_stream) become slots resolved at
run time.
"""
import ctypes as C
import struct

from ._lib import check

_BYREF = type(C.byref(C.c_int()))
_M64 = (1 << 64) - 1


def _word(a):
    """ctypes argument -> 64-bit word exactly as the C prototype receives it."""
    if a is None:
        return 0
    if isinstance(a, bool):
        return int(a)
    if isinstance(a, int):
        return a & _M64
    if isinstance(a, C.c_float):
        return struct.unpack("<I", struct.pack("<f", a.value))[0]
    if isinstance(a, C.c_double):
        return struct.unpack("<Q", struct.pack("<d", a.value))[0]
    if isinstance(a, C.c_void_p):
        return (a.value or 0) & _M64
    if isinstance(a, (C.c_int64, C.c_int32, C.c_uint32, C.c_uint64, C.c_int16, C.c_int8)):
        return a.value & _M64
    if isinstance(a, _BYREF):
        return C.addressof(a._obj)
    if isinstance(a, (C.Array, C.Structure)):
        return C.addressof(a)
    raise TypeError("op-table: cannot marshal an argument of type %s" % type(a).__name__)


class OpTable:
    """ops: [(name, ctypes function of the library, argument tuple)]; stream_objs: the c_void_p objects whose CURRENT value
    is to be used wherever an argument `is` one of them (identity, not equality: unbound streams are all 0)."""

    def __init__(self, L, ops, stream_objs):
        self.L = L
        self.names = [name for name, _, _ in ops]
        self.stream_objs = list(stream_objs)
        self._keep = [args for _, _, args in ops]        # byref targets / arrays referenced by address
        self.h = C.c_void_p(0)
        check(L.rd_optable_create(C.byref(self.h)), "rd_optable_create")
        slot_of = {id(s): k for k, s in enumerate(self.stream_objs)}
        for name, fn, args in ops:
            n = len(args)
            words = (C.c_uint64 * max(n, 1))()
            slots = (C.c_int32 * max(n, 1))()
            stream_vals = {s.value for s in self.stream_objs if s.value}
            for i, a in enumerate(args):
                k = slot_of.get(id(a), -1) if isinstance(a, C.c_void_p) else -1
                slots[i] = k
                words[i] = 0 if k >= 0 else _word(a)
                # a stream handed over as a FRESH c_void_p (equal value, different object) would be baked in and silently survive a
                # later set_stream(): the last argument of every launching entry point is its stream -- it must be a slot
                if k < 0 and i == n - 1 and isinstance(a, C.c_void_p) and a.value and a.value in stream_vals:
                    raise ValueError("op-table: %s passes a stream by value (a new c_void_p), not the plan's stream object" % name)
            rc = L.rd_optable_add(self.h, fn.__name__.encode(), n, words, slots)
            if rc < 0:
                check(rc, "rd_optable_add(%s -> %s)" % (name, fn.__name__))
        self._vals = (C.c_void_p * max(len(self.stream_objs), 1))()
        self._failed = C.c_int32(-1)

    def __len__(self):
        return len(self.names)

    def run(self, begin=0, end=None, lanes=1):
        """lanes > 1: issue from that many host threads, one per stream slot modulo lanes (rd_optable_run_mt; never under capture)."""
        end = len(self.names) if end is None else end
        for k, s in enumerate(self.stream_objs):
            self._vals[k] = s.value
        if lanes > 1:
            rc = self.L.rd_optable_run_mt(self.h, begin, end, self._vals, len(self.stream_objs), lanes, C.byref(self._failed))
        else:
            rc = self.L.rd_optable_run(self.h, begin, end, self._vals, len(self.stream_objs), C.byref(self._failed))
        if rc != 0:
            k = self._failed.value
            check(rc, self.names[k] if 0 <= k < len(self.names) else "rd_optable_run")

    def close(self):
        h, self.h = self.h, C.c_void_p(0)
        if h and h.value:
            self.L.rd_optable_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
