"""Timing helper: dependency flags alone on configs[2], device time per call (both kernels)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, synth  # noqa: E402

V = int(os.environ.get("K2_V", 1000000))
rowptr, col = synth.dep_graph(3, V, 10)
flags = synth.dep_flags(3, rowptr, col)
ctx = Context(0)
ctx.dep_graph_load(rowptr, col)
d_in = ctx.to_device(flags)
d_out, d_ref = ctx.alloc(V), ctx.alloc(4 * V)
best = 1e30
for _ in range(5):
    for _ in range(10):
        ctx.dep_flags_dev(d_in.ptr, d_out.ptr, d_ref.ptr)
    ctx.sync()
    ctx.timer_begin()
    for _ in range(100):
        ctx.dep_flags_dev(d_in.ptr, d_out.ptr, d_ref.ptr)
    best = min(best, ctx.timer_end() * 1000 / 100)
print("%s dep_flags us/call %.2f  (E=%d)" % (os.environ.get("MONTECARLO_LIB", "product"), best, len(col)))
