"""A/B timing of the large-batch alias kernels (MCX_K1_TILES=0/1 chosen per table load) and a
check of both against the CPU twin on a sample."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from montecarlo_amd import Context, synth  # noqa: E402

P = int(os.environ.get("K1_P", 30000))
sizes = [int(x) for x in os.environ.get("K1_NS", "20000,100000,1000000").split(",")]
A, a_off, a_mod = synth.alias_table(2, P)
ctx = Context(0)
for N in sizes:
    S, s_off = synth.modalias_strings(2, P, N)
    dS, dO = ctx.to_device(S), ctx.to_device(s_off)
    dF, dC = ctx.alloc(4 * N), ctx.alloc(4 * N)
    res = {}
    for tiles in ("0", "1"):
        os.environ["MCX_K1_TILES"] = tiles
        ctx.alias_table_load(A, a_off, a_mod)
        for _ in range(2):
            ctx.alias_match_dev(dS.ptr, dO.ptr, N, dF.ptr, dC.ptr)
        ctx.sync()
        reps = 5
        ctx.timer_begin()
        for _ in range(reps):
            ctx.alias_match_dev(dS.ptr, dO.ptr, N, dF.ptr, dC.ptr)
        ms = ctx.timer_end() / reps
        res[tiles] = (dF.download(np.uint32, N), dC.download(np.uint32, N))
        print(f"N={N} P={P} tiles={tiles}: {ms:.3f} ms  ({N * P / ms / 1e9:.2f}e12 pairs/s)", flush=True)
    same = np.array_equal(res["0"][0], res["1"][0]) and np.array_equal(res["0"][1], res["1"][1])
    n_chk = min(N, 3000)
    tf, tc = oracle.alias_match(A, a_off, S, s_off, i0=0, i1=n_chk, use_twin=True)
    ok = np.array_equal(res["1"][0][:n_chk], tf) and np.array_equal(res["1"][1][:n_chk], tc)
    print(f"   tiles == older kernel: {same}; tiles == twin on {n_chk} strings: {ok}", flush=True)
    for b in (dS, dO, dF, dC):
        b.free()
