"""Time the fleet pipeline (device-resident inputs) on one GPU.
env: FL_H hosts, FL_DEV mean devices per host, FL_P patterns, FL_REPS."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import Context, safe_module_names, synth  # noqa: E402

H = int(os.environ.get("FL_H", 100000))
DEV = int(os.environ.get("FL_DEV", 100))
P = int(os.environ.get("FL_P", 30000))
REPS = int(os.environ.get("FL_REPS", 5))
seed = 61
t0 = time.time()
A, a_off, a_mod = synth.alias_table(seed, P)
V = synth.alias_modules(P)
rowptr, col = synth.dep_graph(seed + 1, V, 10)
dev_cnt, extra_off, extra_ids = synth.fleet_hosts(seed + 2, V, H, DEV, 8)
dev_off = np.zeros(H + 1, dtype=np.uint32)
dev_off[1:] = np.cumsum(dev_cnt)
N = int(dev_off[H])
S, s_off = synth.modalias_strings(seed + 3, P, N)
safe = safe_module_names(synth.module_names(seed + 4, V))
print(f"fleet: H={H} N={N} P={P} V={V} E={len(col)} X={len(extra_ids)} bytes={len(S)} gen {time.time()-t0:.1f}s",
      flush=True)
with Context(0) as ctx:
    ctx.alias_table_load(A, a_off, a_mod)
    ctx.dep_graph_load(rowptr, col)
    d_S, d_soff = ctx.to_device(S), ctx.to_device(s_off)
    d_dev, d_xoff = ctx.to_device(dev_off), ctx.to_device(extra_off)
    d_xid, d_safe = ctx.to_device(extra_ids), ctx.to_device(safe)
    d_roff = ctx.alloc(4 * (H + 1))
    # size the list with a first call (cap 0 -> MCX_E_TOO_LARGE carries the total)
    total = ctx.fleet_roots_dev(d_S.ptr, d_soff.ptr, N, d_dev.ptr, H, d_xoff.ptr, d_xid.ptr, len(extra_ids),
                                d_safe.ptr, d_roff.ptr, None, 0, size_only=True)
    d_rid = ctx.alloc(4 * max(total, 1))
    d_first, d_count = ctx.alloc(4 * N), ctx.alloc(4 * N)
    for _ in range(2):
        ctx.fleet_roots_dev(d_S.ptr, d_soff.ptr, N, d_dev.ptr, H, d_xoff.ptr, d_xid.ptr, len(extra_ids),
                            d_safe.ptr, d_roff.ptr, d_rid.ptr, total)
    ctx.sync()
    ctx.timer_begin()
    for _ in range(REPS):
        ctx.fleet_roots_dev(d_S.ptr, d_soff.ptr, N, d_dev.ptr, H, d_xoff.ptr, d_xid.ptr, len(extra_ids),
                            d_safe.ptr, d_roff.ptr, d_rid.ptr, total)
    ms = ctx.timer_end() / REPS
    ctx.timer_begin()
    for _ in range(REPS):
        ctx.alias_match_dev(d_S.ptr, d_soff.ptr, N, d_first.ptr, d_count.ptr)
    ms_k1 = ctx.timer_end() / REPS
    print(f"fleet pipeline {ms:.3f} ms/pass  ({N / ms / 1e3:.1f} M devices/s, {H / ms / 1e3:.2f} M hosts/s), "
          f"listed {total}; alias match alone {ms_k1:.3f} ms", flush=True)
    # host-buffer entry point against the resident one (PCIe-inclusive rate)
    n1 = min(N, 1000)
    t = time.perf_counter()
    for _ in range(20):
        ctx.alias_match(S[: s_off[n1]], s_off[: n1 + 1])
    host_ms = (time.perf_counter() - t) / 20 * 1e3
    ctx.timer_begin()
    for _ in range(20):
        ctx.alias_match_dev(d_S.ptr, d_soff.ptr, n1, d_first.ptr, d_count.ptr)
    res_ms = ctx.timer_end() / 20
    print(f"alias match {n1} strings: host-buffer call {host_ms*1e3:.1f} us, resident {res_ms*1e3:.1f} us", flush=True)
