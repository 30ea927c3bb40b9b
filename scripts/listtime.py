"""Timing helper: match lists against plain first/count on 70 000 x 30 000 (device buffers)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, synth  # noqa: E402

N, P = int(os.environ.get("K1_N", 70000)), int(os.environ.get("K1_P", 30000))
A, a_off, a_mod = synth.alias_table(2, P)
S, s_off = synth.modalias_strings(2, P, N)
ctx = Context(0)
ctx.alias_table_load(A, a_off, a_mod)
dS, dO = ctx.to_device(S), ctx.to_device(s_off)
dF, dC, dL = ctx.alloc(4 * N), ctx.alloc(4 * N), ctx.alloc(4 * (N + 1))
total = ctx.alias_match_list_dev(dS.ptr, dO.ptr, N, 0, dL.ptr, None, 0, size_only=True)
dI = ctx.alloc(4 * max(total, 1))
for name, fn in (("first/count", lambda: ctx.alias_match_dev(dS.ptr, dO.ptr, N, dF.ptr, dC.ptr)),
                 ("lists K=0", lambda: ctx.alias_match_list_dev(dS.ptr, dO.ptr, N, 0, dL.ptr, dI.ptr, total)),
                 ("lists K=3", lambda: ctx.alias_match_list_dev(dS.ptr, dO.ptr, N, 3, dL.ptr, dI.ptr, total))):
    for _ in range(3):
        fn()
    ctx.sync()
    ctx.timer_begin()
    for _ in range(20):
        fn()
    print(f"N={N} P={P} {name}: {ctx.timer_end() / 20 * 1000:.1f} us/call (total {total} entries)")
