"""Soak run: many seeded tables and batches of every shape through the GPU matcher, each
checked against the CPU twin (itself pinned to glibc fnmatch by the CPU suite); a sample of
every case also against glibc directly.  Prints a line per case; stops at the first mismatch.
env: SOAK_SECONDS (default 150), SOAK_SEED."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
from montecarlo_amd import Context, synth  # noqa: E402
from montecarlo_amd.binding import pack_strings  # noqa: E402

budget = float(os.environ.get("SOAK_SECONDS", 150))
rnd = random.Random(int(os.environ.get("SOAK_SEED", 1)))
tw = oracle.twin()
dump_case = int(os.environ.get("SOAK_DUMP_CASE", 0))   # write the inputs of that case to /tmp and stop (no GPU)
ctx = None if dump_case else Context(0)
t_end = time.time() + budget
case = 0


def metachar_table(n_pat, n_str, alphabet_p, alphabet_s, max_p, max_s, normalize=0):
    pats = []
    while len(pats) < n_pat:
        p = bytes(rnd.choice(alphabet_p) for _ in range(rnd.randint(0, max_p)))
        if tw.twin_compile_check(p, len(p), normalize) == 0:
            pats.append(p)
    strs = [bytes(rnd.choice(alphabet_s) for _ in range(rnd.randint(0, max_s))) for _ in range(n_str)]
    return pack_strings(pats), pack_strings(strs)


while time.time() < t_end:
    case += 1
    kind = rnd.choice(["synth", "synth", "synth", "meta", "meta_long", "dup", "longstr", "bigtable", "meta_norm"])
    norm = rnd.random() < 0.3
    if kind in ("synth", "dup"):
        seed = rnd.randint(1, 10**6)
        P = rnd.choice([1, 7, 100, 1000, 5000, 30000, 60000])
        N = rnd.choice([1, 3, 64, 65, 500, 2049, 9000, 66000, 150000])
        A, a_off, _ = synth.alias_table(seed, P)
        S, s_off = synth.modalias_strings(seed + rnd.randint(0, 1), P, N)
        if kind == "dup":                      # repeated texts, table order shuffled in blocks
            raw = A.tobytes()
            pats = [raw[a_off[j]:a_off[j + 1]] for j in range(P)]
            pats = pats[::3] + pats
            A, a_off = pack_strings(pats)
        norm = False if kind == "dup" else norm
    elif kind == "longstr":
        # strings on both sides of the 64-byte and of the tile-buffer limits: tails appended to seeded strings
        seed = rnd.randint(1, 10**6)
        P, N = rnd.choice([100, 3000, 30000]), rnd.choice([5, 300, 2100, 70000])
        A, a_off, _ = synth.alias_table(seed, P)
        S0, s0 = synth.modalias_strings(seed, P, N)
        raw = S0.tobytes()
        strs = []
        for i in range(N):
            t = raw[s0[i]:s0[i + 1]]
            r = rnd.random()
            if r < 0.3:
                t += bytes(rnd.choice(b"0123456789ABCDEFscbi-") for _ in range(rnd.choice([1, 9, 20, 70, 300, 2500])))
            strs.append(t)
        S, s_off = pack_strings(strs)
    elif kind == "bigtable":
        seed = rnd.randint(1, 10**6)
        P, N = rnd.choice([300000, 1000000]), rnd.choice([40, 3000, 66000])
        A, a_off, _ = synth.alias_table(seed, P)
        S, s_off = synth.modalias_strings(seed, P, N)
        norm = False
    elif kind == "meta_norm":
        # kmod's dash rule on patterns and on queried strings, brackets and dashes everywhere
        (A, a_off), (S, s_off) = metachar_table(rnd.randint(1, 500), rnd.choice([1, 300, 2100, 66000]),
                                                b"ab-_[]*?c-", b"ab-_[]c-_", 12, 10, normalize=1)
        norm = True
    elif kind == "meta":
        (A, a_off), (S, s_off) = metachar_table(rnd.randint(1, 800), rnd.randint(1, 400), b"ab[]!^-\\*?c",
                                                b"ab[]-\\c!^*?", 9, 7)
        norm = False
    else:
        (A, a_off), (S, s_off) = metachar_table(rnd.randint(1, 300), rnd.randint(1, 300), b"abc*?[]-x:0123",
                                                b"abcx:0123-", 90, 100)
    P, N = len(a_off) - 1, len(s_off) - 1
    if dump_case:
        if case == dump_case:
            np.savez("/tmp/soak_case.npz", A=A, a_off=a_off, S=S, s_off=s_off, norm=norm)
            print("dumped case", case, kind, P, N)
            sys.exit(0)
        rnd.sample(range(N), min(N, 3 if P > 20000 else 12))
        continue
    # two tuning knobs of the library (mcx_ctx_tune), drawn per case: the group size from which a group
    # gets a second index, and the batch size from which the large-batch kernel runs
    knobs = {"MCX_SUB_MIN": rnd.choice([None, None, 3, 20]), "MCX_BULK_MIN": rnd.choice([None, None, 1, 100000000])}
    ctx.tune(sub_min=knobs["MCX_SUB_MIN"] or 0, tiles_min="auto" if knobs["MCX_BULK_MIN"] is None else knobs["MCX_BULK_MIN"],
             compile_threads=rnd.choice([0, 1, 3]))
    try:
        ctx.alias_table_load(A, a_off, np.zeros(P, dtype=np.uint32), flags=1 if norm else 0)
    except Exception as e:                     # noqa: BLE001 - a table kmod's rule refuses: skip
        print(f"case {case}: {kind} P={P} refused at load ({e})", flush=True)
        continue
    first, count = ctx.alias_match(S, s_off)
    t_first, t_count = oracle.alias_match(A, a_off, S, s_off, normalize_dash=norm, use_twin=True)
    ok = np.array_equal(first, t_first) and np.array_equal(count, t_count)
    pick = sorted(rnd.sample(range(N), min(N, 3 if P > 20000 else 12)))
    Ss, ss_off = pack_strings([S[s_off[i]:s_off[i + 1]].tobytes() for i in pick])
    g_first, g_count = oracle.alias_match(A, a_off, Ss, ss_off, normalize_dash=norm)
    ok = ok and np.array_equal(first[pick], g_first) and np.array_equal(count[pick], g_count)
    if ok and rnd.random() < 0.4 and int(count.sum()) < 5_000_000:
        # the list variant of the kernel: sizes = counts, heads = first matches, and the picked
        # strings' whole lists as glibc orders them
        K = rnd.choice([0, 0, 1, 3])
        off, idx = ctx.alias_match_list(S, s_off, K=K)
        want = count if K == 0 else np.minimum(count, K)
        ok = np.array_equal(np.diff(off.astype(np.int64)), want)
        has = want > 0
        ok = ok and np.array_equal(idx[off[:-1][has]], first[has])
        o_off, o_idx = oracle.alias_match_list(A, a_off, Ss, ss_off, K=K, normalize_dash=norm)
        got = np.concatenate([idx[off[i]:off[i + 1]] for i in pick]) if pick else np.zeros(0, np.uint32)
        ok = ok and np.array_equal(got, o_idx)
        kind += f"+list(K={K})"
    print(f"case {case}: {kind} P={P} N={N} norm={int(norm)} sub_min={knobs['MCX_SUB_MIN']} bulk_min={knobs['MCX_BULK_MIN']} "
          f"matches={int(count.sum())} {'ok' if ok else 'MISMATCH'}",
          flush=True)
    if not ok:
        bad = np.nonzero((first != t_first) | (count != t_count))[0][:5]
        print("first mismatching strings:", bad.tolist(), first[bad].tolist(), t_first[bad].tolist(), flush=True)
        sys.exit(1)
print(f"soak ok: {case} cases", flush=True)
