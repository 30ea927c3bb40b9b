"""Soak run of everything behind K1: dependency flags and closures, match lists, device-list
predicates with packing, the fleet pipeline - random shapes, each case checked against the CPU
oracle.  Prints a line per case; stops at the first mismatch.  env: SOAK_SECONDS, SOAK_SEED."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
from montecarlo_amd import Context, safe_module_names, synth  # noqa: E402
from montecarlo_amd.binding import pack_strings  # noqa: E402

budget = float(os.environ.get("SOAK_SECONDS", 150))
seed0 = int(os.environ.get("SOAK_SEED", 1))
rnd = random.Random(seed0)
rng = np.random.default_rng(seed0)
ctx = Context(0)
t_end = time.time() + budget
case = 0
NO = 0xFFFFFFFF


def fail(msg):
    print("MISMATCH:", msg, flush=True)
    sys.exit(1)


def random_graph(V, E, hubs, cyclic):
    src = rng.integers(0, V, size=E)
    dst = rng.integers(0, V, size=E)
    if hubs:                                    # a few modules needed by (or needing) very many
        for h in rng.integers(0, V, size=hubs):
            k = int(rng.integers(300, max(301, min(V, 6000))))
            if rng.random() < 0.7:
                src = np.concatenate([src, rng.integers(0, V, size=k)])
                dst = np.concatenate([dst, np.full(k, h)])
            else:
                src = np.concatenate([src, np.full(k, h)])
                dst = np.concatenate([dst, rng.integers(0, V, size=k)])
    if not cyclic:                              # u needs v only for v < u
        lo, hi = np.minimum(src, dst), np.maximum(src, dst)
        src, dst = hi, lo
    keep = src != dst
    e = np.unique(np.stack([src[keep], dst[keep]], axis=1), axis=0)
    rowptr = np.zeros(V + 1, dtype=np.uint32)
    rowptr[1:] = np.cumsum(np.bincount(e[:, 0], minlength=V), dtype=np.uint64).astype(np.uint32)
    return rowptr, np.ascontiguousarray(e[:, 1].astype(np.uint32))


while time.time() < t_end:
    case += 1
    kind = rnd.choice(["k2", "k2", "lists", "devices", "fleet"])
    if kind == "k2":
        V = rnd.choice([1, 2, 33, 1000, 20000, 300000])
        E = int(V * rnd.choice([0, 0.5, 3, 12]))
        rowptr, col = random_graph(V, E, rnd.choice([0, 0, 3]) if V >= 1000 else 0, rnd.random() < 0.3)
        flags = rng.choice([0, 1, 3, 4, 5, 7], size=V, p=[.1, .2, .1, .2, .3, .1]).astype(np.uint8)
        ctx.dep_graph_load(rowptr, col)
        out, ref = ctx.dep_flags(flags)
        o_out, o_ref = oracle.dep_flags(rowptr, col, flags)
        if not (np.array_equal(out, o_out) and np.array_equal(ref, o_ref)):
            fail(f"dep_flags V={V} E={len(col)}")
        # several searches on one graph: a search places its launches by the last one's frontier sizes
        nseeds = [rnd.choice([1, 5, max(1, V // 10)]) for _ in range(rnd.choice([1, 2, 3]))]
        for nseed in nseeds:
            seeds = np.zeros((V + 31) // 32, dtype=np.uint32)
            for v in rng.integers(0, V, size=nseed):
                seeds[v >> 5] |= np.uint32(1 << (int(v) & 31))
            for direction in (0, 1):
                clo, edges, levels = ctx.dep_closure(flags, seeds, direction)
                o_clo, o_edges, o_levels = oracle.dep_closure(rowptr, col, flags, seeds, direction)
                if not (np.array_equal(clo, o_clo) and edges == o_edges and levels == o_levels):
                    fail(f"closure V={V} E={len(col)} dir={direction} seeds={nseeds} {edges}/{o_edges} {levels}/{o_levels}")
        print(f"case {case}: k2 V={V} E={len(col)} seeds={nseeds} ok", flush=True)
    elif kind == "lists":
        seed = rnd.randint(1, 10**6)
        P, N, K = rnd.choice([50, 2000, 30000]), rnd.choice([1, 300, 5000, 70000]), rnd.choice([0, 0, 1, 3])
        A, a_off, _ = synth.alias_table(seed, P)
        raw = A.tobytes()
        pats = [raw[a_off[j]:a_off[j + 1]] for j in range(P)]
        A, a_off = pack_strings(pats + pats[::5])
        S, s_off = synth.modalias_strings(seed, P, N)
        ctx.alias_table_load(A, a_off, np.zeros(len(a_off) - 1, dtype=np.uint32))
        off, idx = ctx.alias_match_list(S, s_off, K=K)
        pick = sorted(rnd.sample(range(N), min(N, 40 if P < 30000 else 6)))
        Ss, ss_off = pack_strings([S[s_off[i]:s_off[i + 1]].tobytes() for i in pick])
        o_off, o_idx = oracle.alias_match_list(A, a_off, Ss, ss_off, K=K)
        for n, i in enumerate(pick):
            if not np.array_equal(idx[off[i]:off[i + 1]], o_idx[o_off[n]:o_off[n + 1]]):
                fail(f"match list seed={seed} P={P} N={N} K={K} string {i}")
        first, count = ctx.alias_match(S, s_off)
        want = np.minimum(count, K) if K else count
        if not np.array_equal(np.diff(off.astype(np.int64)), want):
            fail(f"list lengths seed={seed} P={P} N={N} K={K}")
        print(f"case {case}: lists P={P} N={N} K={K} entries={len(idx)} ok", flush=True)
    elif kind == "devices":
        seed = rnd.randint(1, 10**6)
        N, H, cap = rnd.choice([0, 1, 700, 40000, 300000]), rnd.choice([1, 3, 500, 20000]), rnd.choice([0, 1, 7, 64])
        S, s_off = synth.modalias_strings(seed, 2000, N)
        rec = synth.device_records(seed, S, s_off)
        arena, off = synth.driver_names_packed()
        dev_off = np.concatenate([[0], np.sort(rng.integers(0, N + 1, size=H - 1)), [N]]).astype(np.uint32)
        v, kept = ctx.device_filter(rec, arena, off, dev_off, cap)
        o_v, o_kept = oracle.device_filter(rec, arena, off, dev_off, cap)
        if not (np.array_equal(v & 3, o_v) and np.array_equal(kept, o_kept)):
            fail(f"device filter seed={seed} N={N} H={H} cap={cap}")
        if N:
            d_v, d_S, d_soff = ctx.to_device(v), ctx.to_device(S), ctx.to_device(s_off)
            d_dev, d_kept = ctx.to_device(dev_off), ctx.to_device(kept)
            K = int(kept[-1])
            d_S2, d_soff2 = ctx.alloc(max(len(S), 1)), ctx.alloc(4 * (K + 1))
            nbytes = ctx.fleet_compact_dev(d_v.ptr, d_S.ptr, d_soff.ptr, N, d_dev.ptr, H, d_kept.ptr, K, d_S2.ptr,
                                           d_soff2.ptr)
            keep = np.nonzero(v & 1)[0]
            raw = S.tobytes()
            want = b"".join(raw[s_off[i]:s_off[i + 1]] for i in keep) if N <= 40000 else None
            got = d_S2.download(np.uint8, max(len(S), 1))[:nbytes].tobytes()
            lens = np.diff(s_off.astype(np.int64))[keep]
            if nbytes != lens.sum() or (want is not None and got != want):
                fail(f"packing seed={seed} N={N} H={H} cap={cap}")
            for buf in (d_v, d_S, d_soff, d_dev, d_kept, d_S2, d_soff2):
                buf.free()
        print(f"case {case}: devices N={N} H={H} cap={cap} listed={int(kept[-1])} ok", flush=True)
    else:
        seed = rnd.randint(1, 10**6)
        P, H = rnd.choice([300, 3000, 30000]), rnd.choice([1, 40, 700, 5000])
        mean_dev, max_extra = rnd.choice([1, 8, 40]), rnd.choice([0, 2, 9])
        A, a_off, a_mod = synth.alias_table(seed, P)
        V = synth.alias_modules(P)
        rowptr, col = synth.dep_graph(seed + 1, V, rnd.choice([2, 6]))
        dev_cnt, extra_off, extra_ids = synth.fleet_hosts(seed + 2, V, H, mean_dev, max(max_extra, 1))
        if max_extra == 0:
            extra_off, extra_ids = np.zeros(H + 1, dtype=np.uint32), np.zeros(0, dtype=np.uint32)
        dev_off = np.zeros(H + 1, dtype=np.uint32)
        dev_off[1:] = np.cumsum(dev_cnt)
        N = int(dev_off[H])
        S, s_off = synth.modalias_strings(seed + 3, P, N)
        safe = safe_module_names(synth.module_names(seed + 4, V))
        ctx.alias_table_load(A, a_off, a_mod)
        ctx.dep_graph_load(rowptr, col)
        off, ids = ctx.fleet_roots(S, s_off, dev_off, extra_off, extra_ids, safe)
        first, _ = oracle.alias_match(A, a_off, S, s_off, use_twin=True)
        driver = np.where(first != NO, a_mod[np.minimum(first, len(a_mod) - 1)], NO).astype(np.uint32)
        hw = ctx.alias_hw_flags(V)
        w_off, w_ids = oracle.fleet_roots(rowptr, col, driver, dev_off, extra_off, extra_ids, safe, hw)
        if not (np.array_equal(off, w_off) and np.array_equal(ids, w_ids)):
            fail(f"fleet seed={seed} P={P} H={H} dev={mean_dev} extra={max_extra}")
        print(f"case {case}: fleet P={P} H={H} N={N} listed={len(ids)} ok", flush=True)
print(f"soak ok: {case} cases", flush=True)
