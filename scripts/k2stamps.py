"""Diagnostic (needs a library built with -DK2B_STAMPS, MONTECARLO_LIB): per-level time stamps of
k2b_expand's block 0 / wave 0 in the 16-seed closure; prints microseconds between the stamps."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, synth  # noqa: E402

V = 1000000
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
out = np.zeros((64, 16), dtype=np.uint64)
assert ctx.lib.mcx_debug_k2b_stamps(ctypes.c_void_p(out.ctypes.data)) == 0
names = ["entry", "cnt", "entries", "ids", "look", "hash", "claim", "bufd", "sync", "resv", "end"]
print("edges", edges, "levels", levels, "(100 MHz clock: 0.01 us a tick)")
for lv in range(20):
    row = out[lv]
    if row[0] == 0:
        continue
    t = [(int(row[k]) - int(row[0])) / 100.0 if row[k] else None for k in range(11)]
    prev = out[lv - 1] if lv else None
    gap = (int(row[0]) - int(max(prev[1:11]))) / 100.0 if prev is not None and prev[0] and row[0] > max(prev[1:11]) else None
    print("level %2d: " % lv + ("(%5.2f after block 0 of the level before) " % gap if gap is not None else "")
          + "  ".join("%s %5.2f" % (names[k], t[k]) for k in range(1, 11) if t[k] is not None))

small = np.zeros((32, 8), dtype=np.uint64)
assert ctx.lib.mcx_debug_k2s_stamps(ctypes.c_void_p(small.ctypes.data)) == 0
print("the last one-workgroup launch that ran levels (us from its level's start):")
sn = ["scan", "ids", "look", "claim", "end"]
for lv in range(32):
    row = small[lv]
    if row[0] == 0:
        continue
    t = [(int(row[k]) - int(row[0])) / 100.0 if row[k] else None for k in range(1, 6)]
    gap = (int(small[lv + 1][0]) - int(row[0])) / 100.0 if lv + 1 < 32 and small[lv + 1][0] > row[0] else None
    print("level %2d: n=%d adj=%d  " % (lv, row[6], row[7]) + "  ".join("%s %5.2f" % (sn[k], t[k]) for k in range(5) if t[k] is not None)
          + ("  next level at %5.2f" % gap if gap else ""))
