"""Diagnostic: per-phase shader-clock cycles of the large-batch K1 kernel (needs `make -C montecarlo_amd/csrc diag`)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MONTECARLO_LIB"] = os.path.join(ROOT, "scripts", "_bin", "libmontecarlo_diag.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from montecarlo_amd import Context, synth, binding  # noqa: E402

N, P = int(os.environ.get("K1_N", 1000000)), int(os.environ.get("K1_P", 30000))
A, a_off, a_mod = synth.alias_table(2, P)
S, s_off = synth.modalias_strings(2, P, N)
ctx = Context(0)
ctx.alias_table_load(A, a_off, a_mod)
dS, dO = ctx.to_device(S), ctx.to_device(s_off)
dF, dC = ctx.alloc(4 * N), ctx.alloc(4 * N)
for _ in range(5):
    ctx.alias_match_dev(dS.ptr, dO.ptr, N, dF.ptr, dC.ptr)
ctx.sync()
lib = binding.load_library()
nb = (N + 7) // 8
buf = np.zeros((nb, 16), dtype=np.uint64)
n = lib.mcx_debug_k1_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb)
st = buf[:n].astype(np.int64)
cols = [0, 1, 7, 2, 4, 5, 6]
names = ["stage", "grams+probe", "second index", "rounds", "leftover runs", "store"]
d = np.diff(st[:, cols], axis=1)
print("blocks", n, "kernel span (cycles)", int(st[:, 6].max() - st[:, 0].min()))
for k, nm in enumerate(names):
    print("%-14s median %7d  mean %7d  p90 %7d  max %7d" % (nm, np.median(d[:, k]), d[:, k].mean(), np.percentile(d[:, k], 90), d[:, k].max()))
tot = st[:, 6] - st[:, 0]
print("block total    median %7d  mean %7d  p90 %7d  max %7d" % (np.median(tot), tot.mean(), np.percentile(tot, 90), tot.max()))
