"""Timing helper: one closure configuration (env K2_SEEDS, K2_DIR) on configs[2]."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, synth  # noqa: E402

V = int(os.environ.get("K2_V", 1000000))
nseed, direction = int(os.environ.get("K2_SEEDS", 100000)), int(os.environ.get("K2_DIR", 0))
rowptr, col = synth.dep_graph(3, V, 10)
flags = synth.dep_flags(3, rowptr, col)
ctx = Context(0)
ctx.dep_graph_load(rowptr, col)
d_in = ctx.to_device(flags)
d_clo = ctx.alloc(4 * ((V + 31) // 32))
seeds, got = synth.seed_bitmap(3, flags, nseed)
d_seed = ctx.to_device(seeds)
for _ in range(3):
    edges, levels = ctx.dep_closure_dev(d_in.ptr, d_seed.ptr, d_clo.ptr, direction)
ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
    edges, levels = ctx.dep_closure_dev(d_in.ptr, d_seed.ptr, d_clo.ptr, direction)
ctx.sync()
print("closure dir=%d seeds=%d edges=%d levels=%d us/call %.1f" % (direction, got, edges, levels,
                                                                  (time.perf_counter() - t0) / 5 * 1e6))
