"""Timing helper: the host-buffer entry point mcx_alias_match (buffers in ordinary host memory,
everything included) for a few batch sizes against the 30000-pattern table: through the binding
(arrays checked and results allocated per call) and as the bare C call on prepared arrays."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, synth  # noqa: E402
from montecarlo_amd.binding import _ptr  # noqa: E402

A, a_off, a_mod = synth.alias_table(2, 30000)
ctx = Context(0)
ctx.alias_table_load(A, a_off, a_mod)
for N in [int(x) for x in os.environ.get("HOST_N", "1,100,1000,3000,10000,25000,60000").split(",")]:
    S, s_off = synth.modalias_strings(2, 30000, N)
    for _ in range(50):
        ctx.alias_match(S, s_off)
    t0 = time.perf_counter()
    for _ in range(300):
        ctx.alias_match(S, s_off)
    bound = (time.perf_counter() - t0) / 300 * 1e6
    first, count = np.empty(N, np.uint32), np.empty(N, np.uint32)
    args = (ctx.h, _ptr(S), _ptr(s_off), N, _ptr(first), _ptr(count))
    fn = ctx.lib.mcx_alias_match
    t0 = time.perf_counter()
    for _ in range(300):
        fn(*args)
    bare = (time.perf_counter() - t0) / 300 * 1e6
    print("N=%d host-buffer call: binding %.1f us, bare C call %.1f us" % (N, bound, bare), flush=True)

# the K2 host-buffer calls on a graph the size of one host's module table
import ctypes  # noqa: E402

for V in [int(x) for x in os.environ.get("HOST_V", "6000,100000").split(",")]:
    rowptr, col = synth.dep_graph(3, V, 10)
    flags = synth.dep_flags(3, rowptr, col)
    ctx.dep_graph_load(rowptr, col)
    seeds, got = synth.seed_bitmap(3, flags, 4)
    out, ref = np.empty(V, np.uint8), np.empty(V, np.uint32)
    clo = np.zeros((V + 31) // 32, np.uint32)
    edges, levels = ctypes.c_uint64(), ctypes.c_uint32()
    calls = {
        "dep_flags": lambda: ctx.lib.mcx_dep_flags(ctx.h, _ptr(flags), _ptr(out), _ptr(ref)),
        "dep_flags (no refcnt)": lambda: ctx.lib.mcx_dep_flags(ctx.h, _ptr(flags), _ptr(out), None),
        "dep_closure holders": lambda: ctx.lib.mcx_dep_closure(ctx.h, _ptr(flags), _ptr(seeds), _ptr(clo), 0,
                                                               ctypes.byref(edges), ctypes.byref(levels)),
        "dep_closure deps": lambda: ctx.lib.mcx_dep_closure(ctx.h, _ptr(flags), _ptr(seeds), _ptr(clo), 1,
                                                            ctypes.byref(edges), ctypes.byref(levels)),
    }
    for name, fn in calls.items():
        for _ in range(20):
            assert fn() == 0
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        print("V=%d %s: host-buffer call %.1f us" % (V, name, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
