"""Timing helper: the closure on configs[2] for a few seed counts, both directions; the
library under test from MONTECARLO_LIB (default: the product build).  K2_SEEDS_LIST=1,16."""
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
d_clo = ctx.alloc(4 * W)
out = []
for nseed in [int(x) for x in os.environ.get("K2_SEEDS_LIST", "1,16,1024,100000").split(",")]:
    seeds, got = synth.seed_bitmap(3, flags, nseed)
    d_seed = ctx.to_device(seeds)
    for direction in (0, 1):
        for _ in range(3):
            ctx.dep_closure_dev(d_in.ptr, d_seed.ptr, d_clo.ptr, direction)
        ctx.sync()
        us = []
        for _ in range(7):     # median of 7 batches of 10 calls
            t0 = time.perf_counter()
            for _ in range(10):
                edges, levels = ctx.dep_closure_dev(d_in.ptr, d_seed.ptr, d_clo.ptr, direction)
            ctx.sync()
            us.append((time.perf_counter() - t0) * 1e5)
        us.sort()
        out.append("%d/%d: %.1f (min %.1f) e=%d l=%d" % (got, direction, us[3], us[0], edges, levels))
    d_seed.free()
print(os.environ.get("MONTECARLO_LIB", "product").split("/")[-1], " | ".join(out))
