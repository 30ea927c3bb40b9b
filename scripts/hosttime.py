"""Timing helper: the host-buffer entry point mcx_alias_match (buffers in ordinary host memory,
everything included) for a few batch sizes against the 30000-pattern table."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, synth  # noqa: E402

A, a_off, a_mod = synth.alias_table(2, 30000)
ctx = Context(0)
ctx.alias_table_load(A, a_off, a_mod)
for N in (1, 100, 1000, 3000, 10000, 25000, 60000):
    S, s_off = synth.modalias_strings(2, 30000, N)
    for _ in range(50):
        ctx.alias_match(S, s_off)
    t0 = time.perf_counter()
    for _ in range(300):
        ctx.alias_match(S, s_off)
    print("N=%d host-buffer call %.1f us" % (N, (time.perf_counter() - t0) / 300 * 1e6), flush=True)
