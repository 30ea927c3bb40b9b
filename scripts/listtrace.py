"""Diagnostic: a few match-list calls (70 000 x 30 000), to be run under rocprofv3 --kernel-trace."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, synth  # noqa: E402

N, P = int(os.environ.get("K1_N", 70000)), int(os.environ.get("K1_P", 30000))
A, a_off, a_mod = synth.alias_table(2, P)
S, s_off = synth.modalias_strings(2, P, N)
ctx = Context(0)
ctx.alias_table_load(A, a_off, a_mod)
dS, dO = ctx.to_device(S), ctx.to_device(s_off)
dL = ctx.alloc(4 * (N + 1))
total = ctx.alias_match_list_dev(dS.ptr, dO.ptr, N, 0, dL.ptr, None, 0, size_only=True)
dI = ctx.alloc(4 * max(total, 1))
for _ in range(4):
    ctx.alias_match_list_dev(dS.ptr, dO.ptr, N, 0, dL.ptr, dI.ptr, total)
ctx.sync()
print(total)
