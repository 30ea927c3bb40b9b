"""Timing helper: K1 device time per launch on configs[1] ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, synth  # noqa: E402

N, P = int(os.environ.get("K1_N", 1000)), int(os.environ.get("K1_P", 30000))
A, a_off, a_mod = synth.alias_table(2, P)
S, s_off = synth.modalias_strings(2, P, N)
if os.environ.get("K1_EXTRA_LENS"):
    # literal patterns of every length 1..33: the table then has all 33 key lengths
    import numpy as np
    extra = [b"zz" + b"q" * k for k in range(0, 33)]
    blob = b"".join(extra)
    A = np.concatenate([A, np.frombuffer(blob, dtype=np.uint8)])
    lens = np.cumsum([len(e) for e in extra]).astype(np.uint32) + a_off[-1]
    a_off = np.concatenate([a_off, lens]).astype(np.uint32)
    a_mod = np.concatenate([a_mod, np.zeros(len(extra), dtype=np.uint32)]).astype(np.uint32)
ctx = Context(0)
ctx.alias_table_load(A, a_off, a_mod)
dS, dO = ctx.to_device(S), ctx.to_device(s_off)
dF, dC = ctx.alloc(4 * N), ctx.alloc(4 * N)
for reps in [int(x) for x in os.environ.get("K1_REPS", "200").split(",")]:
    for _ in range(20):
        ctx.alias_match_dev(dS.ptr, dO.ptr, N, dF.ptr, dC.ptr)
    ctx.sync()
    ctx.timer_begin()
    for _ in range(reps):
        ctx.alias_match_dev(dS.ptr, dO.ptr, N, dF.ptr, dC.ptr)
    print("N=%d P=%d reps=%d us/launch %.2f" % (N, P, reps,
                                                       ctx.timer_end() * 1000 / reps))
