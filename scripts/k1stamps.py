"""Diagnostic: per-phase shader-clock cycles of the K1 kernel (needs `make -C montecarlo_amd/csrc diag`)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MONTECARLO_LIB"] = os.path.join(ROOT, "scripts", "_bin", "libmontecarlo_diag.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from montecarlo_amd import Context, synth, binding  # noqa: E402

N, P = int(os.environ.get("K1_N", 1000)), int(os.environ.get("K1_P", 30000))
A, a_off, a_mod = synth.alias_table(2, P)
S, s_off = synth.modalias_strings(2, P, N)
ctx = Context(0)
ctx.alias_table_load(A, a_off, a_mod)
dS, dO = ctx.to_device(S), ctx.to_device(s_off)
dF, dC = ctx.alloc(4 * N), ctx.alloc(4 * N)
for _ in range(50):
    ctx.alias_match_dev(dS.ptr, dO.ptr, N, dF.ptr, dC.ptr)
ctx.sync()
lib = binding.load_library()
buf = np.zeros((65536, 16), dtype=np.uint64)
n = lib.mcx_debug_k1_stamps(buf.ctypes.data_as(ctypes.c_void_p), 65536)
st = buf[:n].astype(np.int64)
d = np.diff(st[:, [0, 1, 2, 2, 4, 5, 6]], axis=1)
names = ["stage", "bloom+probe", "scan", "screen+run", "tail run", "store"]
print("wave 0 gram sets+occ: median %d   probe lane 0: key+hash median %d  slot load+compare median %d p90 %d max %d" % (np.median(st[:, 3] - st[:, 1]), np.median(st[:, 14]), np.median(st[:, 15]), np.percentile(st[:, 15], 90), st[:, 15].max()))
print("blocks", n, "kernel span (cycles)", int(st[:, 6].max() - st[:, 0].min()))
for k, nm in enumerate(names):
    print("%-12s median %7d  p90 %7d  max %7d" % (nm, np.median(d[:, k]), np.percentile(d[:, k], 90), d[:, k].max()))
print("tail: fetch   median %7d   interp+orig median %7d" % (np.median(st[:, 7] - st[:, 4]), np.median(st[:, 5] - st[:, 7])))
print("tail wave0: match median %7d  orig+atomics median %7d  wait others %7d" % (np.median(st[:, 8] - st[:, 7]), np.median(st[:, 9] - st[:, 8]), np.median(st[:, 5] - st[:, 9])))
print("block total  median %7d  p90 %7d  max %7d" % (np.median(st[:, 6] - st[:, 0]), np.percentile(st[:, 6] - st[:, 0], 90), (st[:, 6] - st[:, 0]).max()))
print("start skew   median %7d  max %7d" % (np.median(st[:, 0] - st[:, 0].min()), (st[:, 0] - st[:, 0].min()).max()))
tot = st[:, 6] - st[:, 0]
raw = S.tobytes()
print("run passes of wave 0: stage-copy median %d  masks median %d  combine+atomics median %d  sub-passes median %d max %d" % (
    np.median(st[:, 10]), np.median(st[:, 11]), np.median(st[:, 12]), np.median(st[:, 13]), st[:, 13].max()))
print("slowest blocks: total | stage probe scan screen tail store | copy masks combine subpasses | string")
for b in np.argsort(-tot)[:14]:
    i0 = b * max(1, (N + 2047) // 2048 if N > 2048 else 1)
    print(int(tot[b]), d[b, :6].tolist(), st[b, 10:14].tolist(), raw[s_off[i0]:s_off[i0 + 1]][:70])
print("blocks over 30k cycles:", int((tot > 30000).sum()), " over 40k:", int((tot > 40000).sum()))
