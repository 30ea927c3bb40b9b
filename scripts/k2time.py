"""Timing helper: K2 flags and closure on configs[2] (1M modules, ~10M edges)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, synth  # noqa: E402

V = int(os.environ.get("K2_V", 1000000))
rowptr, col = synth.dep_graph(3, V, 10)
flags = synth.dep_flags(3, rowptr, col)
ctx = Context(0)
ctx.dep_graph_load(rowptr, col)
W = (V + 31) // 32
d_in = ctx.to_device(flags)
d_out, d_ref, d_clo = ctx.alloc(V), ctx.alloc(4 * V), ctx.alloc(4 * W)
for _ in range(5):
    ctx.dep_flags_dev(d_in.ptr, d_out.ptr, d_ref.ptr)
ctx.sync()
ctx.timer_begin()
for _ in range(50):
    ctx.dep_flags_dev(d_in.ptr, d_out.ptr, d_ref.ptr)
print("dep_flags us/launch %.1f  (E=%d)" % (ctx.timer_end() * 1000 / 50, len(col)))
for nseed in (1, 16, 1024, 100000):
    seeds, got = synth.seed_bitmap(3, flags, nseed)
    d_seed = ctx.to_device(seeds)
    for direction in (0, 1):
        ctx.dep_closure_dev(d_in.ptr, d_seed.ptr, d_clo.ptr, direction)
        ctx.sync()
        us = 1e30
        for _ in range(4):     # best of 4 batches of 10 calls
            t0 = time.perf_counter()
            for _ in range(10):
                edges, levels = ctx.dep_closure_dev(d_in.ptr, d_seed.ptr, d_clo.ptr, direction)
            ctx.sync()
            us = min(us, (time.perf_counter() - t0) * 1e5)
        print("closure dir=%d seeds=%d edges=%d levels=%d us/call %.1f  (%.2f Gedges/s)" %
              (direction, got, edges, levels, us, edges / us / 1e3))
    d_seed.free()
