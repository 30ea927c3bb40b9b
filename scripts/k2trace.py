"""Diagnostic: a few closure calls of one configuration (env K2_V, K2_SEEDS, K2_DIR), to be run under
rocprofv3 --kernel-trace; prints nothing but the call's results."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, synth  # noqa: E402

V = int(os.environ.get("K2_V", 1000000))
nseed, direction = int(os.environ.get("K2_SEEDS", 16)), int(os.environ.get("K2_DIR", 0))
rowptr, col = synth.dep_graph(3, V, 10)
flags = synth.dep_flags(3, rowptr, col)
ctx = Context(0)
ctx.dep_graph_load(rowptr, col)
d_in = ctx.to_device(flags)
d_clo = ctx.alloc(4 * ((V + 31) // 32))
seeds, got = synth.seed_bitmap(3, flags, nseed)
d_seed = ctx.to_device(seeds)
for _ in range(4):
    edges, levels = ctx.dep_closure_dev(d_in.ptr, d_seed.ptr, d_clo.ptr, direction)
ctx.sync()
print(edges, levels)
