"""Timing helper: the tile kernel on 10^6 strings against a table of K1_P patterns (default 3 x 10^6)
for a list of group sizes from which a group gets a second index (SUBS; 0 = the library's default, 128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, synth
N, P = 1000000, int(os.environ.get("K1_P", 3000000))
A, a_off, a_mod = synth.alias_table(4, P)
S, s_off = synth.modalias_strings(4, P, N)
ctx = Context(0)
dS, dO = ctx.to_device(S), ctx.to_device(s_off)
dF, dC = ctx.alloc(4 * N), ctx.alloc(4 * N)
for sm in [int(x) for x in os.environ.get("SUBS", "0,32,64,256,512,1024").split(",")]:
    ctx.tune(sub_min=sm)
    ctx.alias_table_load(A, a_off, a_mod)
    for _ in range(3):
        ctx.alias_match_dev(dS.ptr, dO.ptr, N, dF.ptr, dC.ptr)
    ctx.sync()
    ctx.timer_begin()
    for _ in range(10):
        ctx.alias_match_dev(dS.ptr, dO.ptr, N, dF.ptr, dC.ptr)
    print("P=%d sub_min=%d us/launch %.1f" % (P, sm, ctx.timer_end() * 1000 / 10), flush=True)
