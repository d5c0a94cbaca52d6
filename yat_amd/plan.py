"""Launch plans: a recorded training-step half (forward or backward) compiled into segments the C side replays.

``ops.Recorder`` collects, in launch order, every C-ABI call made through ``ops._lib()`` plus the stream / event operations
and host callbacks the model code routes through ``FlatParamModule``'s helpers.  ``LaunchPlan`` turns that list into
``yat_plan_entry`` arrays (include/yat_hip.h, "launch plans") -- one array per run of entries between host callbacks (the
data-parallel hook stays Python) -- and ``replay`` walks them with one ``yat_plan_replay`` call each: ~1 us per entry
instead of a ctypes call (arguments converted every time) or, before recording at all, ~35 us of tensor slicing, stride
arithmetic and struct building per launch.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _l

EV_RECORD, WAIT_EVENT, WAIT_STREAM, PY = "ev_record", "wait_event", "wait_stream", "py"
_FLOATS = (C.c_float, C.c_double)


def _addr(x):
    return C.addressof(x)


class LaunchPlan:
    def __init__(self, entries, dynamic, result=None, saved=None):
        self.result, self.saved = result, saved
        self.keep = []                      # structs, arrays, events the C arrays point into
        self.segments = []                  # ("c", array, n) | ("py", callable)
        self.dynamic = {}                   # name -> [(array, position, argument)]
        lib = _l.load()
        where = {}                          # recorded entry index -> (pending list position)
        pending = []

        def flush():
            if not pending:
                return
            arr = (_l.PlanEntry * len(pending))()
            for k, (op, vals) in enumerate(pending):
                arr[k].op, arr[k].nargs = op, len(vals)
                for j, (is_float, v) in enumerate(vals):
                    if is_float:
                        arr[k].a[j].d = v
                    else:
                        arr[k].a[j].i = v
            for idx, pos in list(where.items()):
                if isinstance(pos, int):
                    where[idx] = (arr, pos)
            self.segments.append(("c", arr, len(pending)))
            pending.clear()

        for idx, e in enumerate(entries):
            kind = e[0]
            if kind == EV_RECORD:
                _, ev, stream = e
                self.keep.append(ev)
                pending.append((-1, [(False, ev.cuda_event), (False, stream.cuda_stream)]))
            elif kind == WAIT_EVENT:
                _, stream, ev = e
                self.keep.append(ev)
                pending.append((-2, [(False, stream.cuda_stream), (False, ev.cuda_event)]))
            elif kind == WAIT_STREAM:
                _, waiter, waited = e
                ev = torch.cuda.Event()
                ev.record(waited)           # creates the handle; an extra record is harmless
                self.keep.append(ev)
                pending.append((-1, [(False, ev.cuda_event), (False, waited.cuda_stream)]))
                pending.append((-2, [(False, waiter.cuda_stream), (False, ev.cuda_event)]))
            elif kind == PY:
                flush()
                self.segments.append(("py", e[1]))
            else:                           # [ctypes function, args]
                fn, args = e
                name = fn.__name__
                op = lib.yat_plan_op_id(name.encode())
                if op < 0 or len(args) > 30:
                    flush()
                    self.segments.append(("py", lambda fn=fn, args=args, name=name: _l.check(fn(*args), name)))
                    self.keep.append(args)
                    continue
                argtypes = _l.SIGNATURES[name][1]
                vals = []
                for a, t in zip(args, argtypes):
                    if t in _FLOATS:
                        vals.append((True, float(a)))
                    elif a is None:
                        vals.append((False, 0))
                    elif isinstance(a, int):
                        vals.append((False, a))
                    elif isinstance(a, C.c_void_p):
                        vals.append((False, a.value or 0))
                    elif hasattr(a, "_obj"):                    # C.byref(struct)
                        self.keep.append(a._obj)
                        vals.append((False, _addr(a._obj)))
                    elif isinstance(a, (C.Structure, C.Array)):
                        self.keep.append(a)
                        vals.append((False, _addr(a)))
                    elif hasattr(a, "contents"):                # C.pointer(struct)
                        self.keep.append(a)
                        vals.append((False, _addr(a.contents)))
                    else:
                        raise TypeError(f"launch plan: cannot convert argument {a!r} of {name}")
                where[idx] = len(pending)
                pending.append((op, vals))
        flush()
        for name, slots in dynamic.items():
            self.dynamic[name] = [(where[e][0], where[e][1], a, mul, add) for e, a, mul, add in slots]
        self.n_entries = len(entries)
        self._fail = C.c_int(0)

    def replay(self, dynamic_values):
        for name, slots in self.dynamic.items():
            v = int(dynamic_values[name])
            for arr, pos, a, mul, add in slots:
                arr[pos].a[a].i = v * mul + add
        lib = _l.load()
        for seg in self.segments:
            if seg[0] == "c":
                rc = lib.yat_plan_replay(seg[1], seg[2], C.byref(self._fail))
                if rc:
                    raise _l.YatLibraryError(f"launch plan replay: entry {self._fail.value} failed with status {rc}")
            else:
                seg[1]()
        return self.result
