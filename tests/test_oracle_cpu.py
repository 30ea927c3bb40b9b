"""CPU-side checks (no GPU): the oracle against the golden vectors, the product's
host pattern compiler + matcher routine (run on the host by the CPU twin)
against glibc fnmatch, and the generators.

Golden vectors: the reference has none for this path ("parity unpinned"); the
files under tests/golden were produced in this image by glibc 2.35 fnmatch,
kmod 29 `modprobe -R` and from the text of the reference's name tables, by the
make_*.py scripts next to them.
"""
import ctypes
import json
import os
import random

import numpy as np
import pytest

from montecarlo_amd import synth
from montecarlo_amd.binding import pack_strings

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NO_MATCH = 0xFFFFFFFF


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


# ---- glibc fnmatch known answers ---------------------------------------------

def test_oracle_matches_fnmatch_kat(oracle_mod):
    kat = load("fnmatch_kat.json")["cases"]
    lib = oracle_mod.oracle()
    assert len(kat) >= 130
    for c in kat:
        p, s = bytes.fromhex(c["pattern"]), bytes.fromhex(c["string"])
        assert lib.orc_fnmatch(p, s) == c["match"], (p, s)


def test_twin_matches_fnmatch_kat(oracle_mod):
    """product compiler + matcher routine on the host == glibc on every vector"""
    kat = load("fnmatch_kat.json")["cases"]
    tw = oracle_mod.twin()
    for c in kat:
        p, s = bytes.fromhex(c["pattern"]), bytes.fromhex(c["string"])
        assert tw.twin_fnmatch(p, len(p), s, len(s)) == c["match"], (p, s)


def test_posix_subexpressions_are_rejected(oracle_mod):
    tw = oracle_mod.twin()
    for p in [b"[[:alpha:]]", b"a[[=a=]]", b"[[.a.]]", b"[a-[.c.]]", b"[x[:digit:]]*"]:
        assert tw.twin_compile_check(p, len(p), 0) == -5, p
    for p in [b"[[:zz", b"[a[", b"[:alpha:]", b"[a:]"]:
        # '[:' outside a bracket, or a '[' that is not followed by : = . is plain
        rc = tw.twin_compile_check(p, len(p), 0)
        assert rc in (0, -5)
    assert tw.twin_compile_check(b"a\x00b", 3, 0) == -5
    assert tw.twin_compile_check(b"x" * 5000, 5000, 0) == -5


def _fuzz(oracle_mod, seed, n_pat, pat_alpha, str_alpha, max_p, max_s, per_pat):
    lib, tw = oracle_mod.oracle(), oracle_mod.twin()
    rnd = random.Random(seed)
    pairs = matches = rejected = 0
    for _ in range(n_pat):
        pat = bytes(rnd.choice(pat_alpha) for _ in range(rnd.randint(0, max_p)))
        if tw.twin_compile_check(pat, len(pat), 0) != 0:
            rejected += 1
            continue
        for _ in range(per_pat):
            s = bytes(rnd.choice(str_alpha) for _ in range(rnd.randint(0, max_s)))
            want = lib.orc_fnmatch(pat, s)
            got = tw.twin_fnmatch(pat, len(pat), s, len(s))
            assert got == want, (pat, s, want, got)
            pairs += 1
            matches += want
    return pairs, matches, rejected


def test_twin_vs_glibc_fuzz_metachar_heavy(oracle_mod):
    pairs, matches, rejected = _fuzz(oracle_mod, 1, 12000, b"ab[]!^-\\*?c:.=[", b"ab[]-\\c!^:*?", 9, 6, 10)
    assert pairs > 100000 and matches > 3000
    assert rejected < 600  # only the POSIX sub-expression forms are refused


def test_twin_vs_glibc_fuzz_small_alphabet(oracle_mod):
    # two-letter alphabet: many matches, exercises '*' placement and anchoring
    pairs, matches, _ = _fuzz(oracle_mod, 2, 8000, b"ab*?*", b"ab", 10, 10, 12)
    assert pairs > 90000 and matches > 20000


def test_twin_vs_glibc_fuzz_ranges_and_high_bytes(oracle_mod):
    alpha = bytes(range(1, 256))
    pairs, matches, _ = _fuzz(oracle_mod, 3, 6000, b"[]-!^\\*?" + b"az09\x01\x7f\x80\xfe\xff", alpha, 8, 4, 10)
    assert pairs > 40000


def test_twin_vs_glibc_fuzz_long_patterns(oracle_mod):
    # many '*' (more segments than a slot's directory names), long bodies (overflow
    # array), segments longer than four chunks, strings on both sides of 64 bytes
    pairs, matches, _ = _fuzz(oracle_mod, 4, 3000, b"a*?*[ab]", b"ab", 48, 90, 12)
    assert pairs > 30000 and matches > 800
    pairs, matches, _ = _fuzz(oracle_mod, 5, 3000, b"aaaaab*?", b"ab", 30, 70, 12)
    assert pairs > 30000 and matches > 300


# ---- kmod resolution fixture ---------------------------------------------------

def test_oracle_reproduces_kmod_resolution(oracle_mod):
    fx = load("kmod_resolve.json")
    pats = [a[0].encode() for a in fx["aliases"]]
    mods = [a[1] for a in fx["aliases"]]
    A, a_off = pack_strings(pats)
    S, s_off = pack_strings([q["query"].encode() for q in fx["queries"]])
    first, count = oracle_mod.alias_match(A, a_off, S, s_off, normalize_dash=True)
    resolved = 0
    for i, q in enumerate(fx["queries"]):
        assert count[i] == len(q["modules"]), q
        if q["modules"]:
            assert mods[first[i]] == q["modules"][0], q
            resolved += 1
        else:
            assert first[i] == NO_MATCH
    assert resolved >= 50
    # the twin applies the same normalisation
    f2, c2 = oracle_mod.alias_match(A, a_off, S, s_off, normalize_dash=True, use_twin=True)
    assert np.array_equal(first, f2) and np.array_equal(count, c2)
    # and raw fnmatch differs from kmod exactly on the dash cases
    f3, c3 = oracle_mod.alias_match(A, a_off, S, s_off, normalize_dash=False)
    diff = [fx["queries"][i]["query"] for i in range(len(c3)) if c3[i] != count[i]]
    assert diff and all("-" in d or "_" in d for d in diff)


def test_oracle_match_lists_reproduce_kmod_order(oracle_mod):
    """every module kmod 29 printed for a query, in its order = the oracle's match list"""
    fx = load("kmod_resolve.json")
    pats = [a[0].encode() for a in fx["aliases"]]
    mods = [a[1] for a in fx["aliases"]]
    A, a_off = pack_strings(pats)
    S, s_off = pack_strings([q["query"].encode() for q in fx["queries"]])
    off, idx = oracle_mod.alias_match_list(A, a_off, S, s_off, normalize_dash=True)
    multi = 0
    for i, q in enumerate(fx["queries"]):
        assert [mods[j] for j in idx[off[i]:off[i + 1]]] == q["modules"], q
        multi += len(q["modules"]) > 1
    assert multi >= 1
    first, count = oracle_mod.alias_match(A, a_off, S, s_off, normalize_dash=True)
    assert np.array_equal(np.diff(off.astype(np.int64)), count)
    off2, idx2 = oracle_mod.alias_match_list(A, a_off, S, s_off, K=1, normalize_dash=True)
    assert np.array_equal(np.diff(off2.astype(np.int64)), np.minimum(count, 1))
    assert np.array_equal(idx2, first[count > 0])


# ---- synthetic tables: twin == oracle -----------------------------------------

@pytest.mark.parametrize("seed,P,N", [(2, 3000, 200), (5, 500, 300), (9, 64, 64)])
def test_twin_equals_oracle_on_synthetic(oracle_mod, seed, P, N):
    A, a_off, a_mod = synth.alias_table(seed, P)
    S, s_off = synth.modalias_strings(seed, P, N)
    f, c = oracle_mod.alias_match(A, a_off, S, s_off)
    f2, c2 = oracle_mod.alias_match(A, a_off, S, s_off, use_twin=True)
    assert np.array_equal(f, f2) and np.array_equal(c, c2)
    assert (c > 0).mean() > 0.6  # 70 % of the strings are instances of a table pattern


def prefix_heavy_table(rng, n_pat, n_str):
    """patterns that share literal prefixes of every length 0..40 and go on with
    literals, '?', brackets and '*' right after them (what the index key and the
    key-less program bodies have to get right), and strings cut from the same cloth"""
    stems = [bytes(rng.choice(list(b"abc01:")) for _ in range(44)) for _ in range(3)]
    tails = [b"", b"*", b"?", b"x", b"[ab]", b"??*", b"*z", b"[!a]c*", b"a?c", b"\\*b", b"*[0-9]*", b"abcde*",
             b"?bcd", b"[a-c][a-c]*", b"*ab*ab", b"ab?d*f"]
    pats, strs = [], []
    for _ in range(n_pat):
        st = stems[rng.integers(0, 3)]
        pats.append(st[: rng.integers(0, 41)] + tails[rng.integers(0, len(tails))])
    fill = list(b"abc01:xz")
    for _ in range(n_str):
        st = stems[rng.integers(0, 3)]
        k = rng.integers(0, 44)
        strs.append(st[:k] + bytes(rng.choice(fill) for _ in range(rng.integers(0, 7))))
    return pats, strs


def test_twin_equals_oracle_on_prefix_heavy_tables(oracle_mod):
    rng = np.random.default_rng(77)
    for _ in range(3):
        pats, strs = prefix_heavy_table(rng, 400, 300)
        A, a_off = pack_strings(pats)
        S, s_off = pack_strings(strs)
        f, c = oracle_mod.alias_match(A, a_off, S, s_off)
        f2, c2 = oracle_mod.alias_match(A, a_off, S, s_off, use_twin=True)
        assert np.array_equal(f, f2) and np.array_equal(c, c2)
        assert (c > 0).sum() > 30


def test_generators_are_deterministic_and_shardable():
    A, a_off, a_mod = synth.alias_table(2, 1000)
    A2, a_off2, _ = synth.alias_table(2, 1000)
    assert np.array_equal(A, A2) and np.array_equal(a_off, a_off2)
    # a shard generated on its own equals the slice of the whole
    B, b_off, b_mod = synth.alias_table(2, 1000, 300, 400)
    assert B.tobytes() == A[a_off[300]:a_off[400]].tobytes()
    assert np.array_equal(b_mod, a_mod[300:400])
    S, s_off = synth.modalias_strings(2, 1000, 500)
    T, t_off = synth.modalias_strings(2, 1000, 500, 100, 200)
    assert T.tobytes() == S[s_off[100]:s_off[200]].tobytes()
    assert a_mod.max() + 1 == synth.alias_modules(1000)


def test_generator_golden():
    fx = load("synth_golden.json")
    A, a_off, a_mod = synth.alias_table(fx["seed"], fx["P"])
    raw = A.tobytes()
    got = [raw[a_off[j]:a_off[j + 1]].decode() for j in range(16)]
    assert got == fx["patterns16"]
    S, s_off = synth.modalias_strings(fx["seed"], fx["P"], 16)
    sraw = S.tobytes()
    assert [sraw[s_off[i]:s_off[i + 1]].decode() for i in range(16)] == fx["strings16"]
    rowptr, col = synth.dep_graph(fx["seed"], 64, 4)
    assert rowptr.tolist() == fx["rowptr64"] and col.tolist() == fx["col64"]
    assert synth.dep_flags(fx["seed"], rowptr, col, 200).tolist() == fx["flags64"]
    assert synth.module_names(fx["seed"], 8) == fx["names8"]


# ---- dependency oracle against brute force ------------------------------------

def _closure_bruteforce(rowptr, col, flags, seeds, direction):
    V = len(rowptr) - 1
    loaded = (flags & 1).astype(bool)
    adj = np.zeros((V, V), dtype=bool)
    for u in range(V):
        for k in range(rowptr[u], rowptr[u + 1]):
            if loaded[u] and loaded[col[k]]:
                if direction == 1:
                    adj[u, col[k]] = True
                else:
                    adj[col[k], u] = True
    reach = np.array([(seeds[v >> 5] >> (v & 31)) & 1 for v in range(V)], dtype=bool) & loaded
    while True:
        nxt = reach | (adj[reach].any(axis=0) if reach.any() else np.zeros(V, bool))
        if np.array_equal(nxt, reach):
            break
        reach = nxt
    return reach


@pytest.mark.parametrize("seed,V,deg", [(3, 300, 4), (4, 1200, 6), (8, 50, 2)])
def test_dep_oracle_against_bruteforce(oracle_mod, seed, V, deg):
    rowptr, col = synth.dep_graph(seed, V, deg)
    flags = synth.dep_flags(seed, rowptr, col, 150)
    out, ref = oracle_mod.dep_flags(rowptr, col, flags)
    loaded = (flags & 1).astype(bool)
    holders = np.zeros(V, dtype=np.int64)
    for u in range(V):
        if loaded[u]:
            for k in range(rowptr[u], rowptr[u + 1]):
                if loaded[col[k]]:
                    holders[col[k]] += 1
    for v in range(V):
        f = int(flags[v])
        exp = 0
        if f & 1:
            if holders[v]:
                exp |= 1
            if (f & 2) or holders[v]:
                exp |= 2
            if (f & 4) and not holders[v]:
                exp |= 4
        assert out[v] == exp
        assert ref[v] == ((holders[v] + (1 if f & 2 else 0)) if f & 1 else NO_MATCH)
    seeds, got = synth.seed_bitmap(seed, flags, 3)
    assert got > 0
    for direction in (0, 1):
        clo, scanned, levels = oracle_mod.dep_closure(rowptr, col, flags, seeds, direction)
        want = _closure_bruteforce(rowptr, col, flags, seeds, direction)
        bits = np.array([(clo[v >> 5] >> (v & 31)) & 1 for v in range(V)], dtype=bool)
        assert np.array_equal(bits, want)
        assert levels >= 1 and scanned >= 0


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_dep_flags_threaded_baseline_equals_sequential(oracle_mod, threads):
    """The edge-parallel form of the CPU baseline (several threads, each counting the
    loaded holders of its own modules) gives the sequential scan's bytes, also on
    empty and tiny graphs."""
    for seed, V, deg in [(5, 20000, 9), (6, 7, 2), (7, 1, 1)]:
        rowptr, col = synth.dep_graph(seed, V, deg)
        flags = synth.dep_flags(seed, rowptr, col, 150)
        out, ref = oracle_mod.dep_flags(rowptr, col, flags)
        rrow, rcol = oracle_mod.holders_graph(rowptr, col)
        out_t, ref_t = oracle_mod.dep_flags_threads(rrow, rcol, flags, threads)
        assert np.array_equal(out, out_t) and np.array_equal(ref, ref_t)


def test_dep_oracle_hand_graphs(oracle_mod):
    # chain 3 -> 2 -> 1 -> 0 ("u needs v"), all loaded
    rowptr = np.array([0, 0, 1, 2, 3], dtype=np.uint32)
    col = np.array([0, 1, 2], dtype=np.uint32)
    flags = np.array([1 | 4, 1 | 4, 1 | 4, 1 | 4 | 2], dtype=np.uint8)
    out, ref = oracle_mod.dep_flags(rowptr, col, flags)
    assert out.tolist() == [1 | 2, 1 | 2, 1 | 2, 2 | 4]
    assert ref.tolist() == [1, 1, 1, 1]
    seed = np.array([1 << 1], dtype=np.uint32)  # module 1
    clo, scanned, levels = oracle_mod.dep_closure(rowptr, col, flags, seed, 0)
    assert clo[0] == 0b1110 and levels == 3  # 1 and its holders 2, 3
    clo, _, _ = oracle_mod.dep_closure(rowptr, col, flags, seed, 1)
    assert clo[0] == 0b0011  # 1 and what it needs
    # unloaded module in the middle cuts the chain
    flags2 = flags.copy()
    flags2[2] = 4
    clo, _, _ = oracle_mod.dep_closure(rowptr, col, flags2, seed, 0)
    assert clo[0] == 0b0010
    out2, ref2 = oracle_mod.dep_flags(rowptr, col, flags2)
    assert out2.tolist() == [1 | 2, 4, 0, 2 | 4] and ref2[2] == NO_MATCH
    # diamond: 3 needs 1 and 2, both need 0
    rowptr = np.array([0, 0, 1, 2, 4], dtype=np.uint32)
    col = np.array([0, 0, 1, 2], dtype=np.uint32)
    flags = np.array([5, 5, 5, 5], dtype=np.uint8)
    out, ref = oracle_mod.dep_flags(rowptr, col, flags)
    assert ref.tolist() == [2, 1, 1, 0] and out.tolist() == [3, 3, 3, 4]


# ---- safe-module name filter ---------------------------------------------------

def test_safe_name_tables(oracle_mod):
    fx = load("safe_names.json")
    lib = oracle_mod.oracle()
    lib.orc_core_count.restype = ctypes.c_int
    lib.orc_prefix_count.restype = ctypes.c_int
    assert lib.orc_core_count() == fx["n_core"] == 113
    assert lib.orc_prefix_count() == fx["n_prefix"] == 56
    assert len(fx["cases"]) >= 300
    for name, want in fx["cases"]:
        assert lib.orc_safe_name(name.encode()) == want, name
    # quirks the reference has and a replacement must keep (bare startswith)
    for name, want in [("acer_wmi", 0), ("stmmac", 0), ("sgx", 0), ("e1000", 1), ("r8169", 1), ("fanotify", 0)]:
        assert lib.orc_safe_name(name.encode()) == want


def test_hw_alias_stage(oracle_mod):
    lib = oracle_mod.oracle()
    assert lib.orc_has_hw_alias(b"pci:v00008086d*\n") == 1
    assert lib.orc_has_hw_alias(b"  usb:v*\nfoo") == 1
    assert lib.orc_has_hw_alias(b"acpi*:PNP0C09:*\ninput:b*\n") == 0
    assert lib.orc_has_hw_alias(b"fs-ext4\n\tof:N*T*Cx\n") == 1
    assert lib.orc_has_hw_alias(b"") == 0
    assert lib.orc_has_hw_alias(b"xpci:v*") == 0


def _k2_golden_case(c):
    V = c["V"]
    seeds = np.zeros((V + 31) // 32, dtype=np.uint32)
    for v in c["seeds"]:
        seeds[v >> 5] |= np.uint32(1 << (v & 31))
    return (np.array(c["rowptr"], np.uint32), np.array(c["col"], np.uint32), np.array(c["in_flags"], np.uint8), seeds)


def test_oracle_k2_golden(oracle_mod):
    """chain and diamond worked out by hand, a frozen random graph (tests/golden/make_k2_golden.py)"""
    for c in load("k2_golden.json")["cases"]:
        rowptr, col, flags, seeds = _k2_golden_case(c)
        out, ref = oracle_mod.dep_flags(rowptr, col, flags)
        assert out.tolist() == c["out_flags"] and ref.tolist() == c["refcnt"], c["name"]
        for direction, key in ((0, "closure_holders"), (1, "closure_deps")):
            clo, edges, levels = oracle_mod.dep_closure(rowptr, col, flags, seeds, direction)
            members = [v for v in range(c["V"]) if clo[v >> 5] >> (v & 31) & 1]
            assert members == c[key]["members"] and edges == c[key]["edges"] and levels == c[key]["levels"]


def test_compiled_tables_are_what_they_were(oracle_mod):
    """The table compiler runs on several host threads.  Every array it hands to the device
    (slots, overflow bodies, signatures, both hash indexes and their tag bytes, key lengths,
    bracket sets, the copies' positions, bus-tag bytes) is compared, by digest, with what the
    single-threaded compiler produced before it was split (tests/golden/compile_digests.json,
    tests/make_compile_digests.py) - with one thread and with thread counts that do not divide
    anything evenly."""
    import json

    import make_compile_digests as mk

    with open(os.path.join(os.path.dirname(__file__), "golden", "compile_digests.json")) as f:
        gold = json.load(f)
    assert len(gold) >= 9
    for key, want in gold.items():
        parts = key.split(":")
        kind, seed, P, norm = parts[0], int(parts[1]), int(parts[2]), bool(int(parts[3]))
        sub_min = int(parts[4][3:]) if len(parts) > 4 else None
        for threads in (1, 3, 16):
            got = mk.digest(kind, seed, P, norm, threads=threads, sub_min=sub_min)
            assert got == want, (key, threads, [i for i in range(13) if got[i] != want[i]])


def test_compile_errors_name_the_first_bad_pattern(oracle_mod):
    """Whatever the number of threads, a refused table is refused for its first bad pattern."""
    import ctypes

    tw = oracle_mod.twin()
    pats = [b"ok:%d*" % k for k in range(20000)]
    pats[15000] = b"bad:[[:alpha:]]"
    pats[4000] = b"bad:[[=a=]]"
    A = np.frombuffer(b"".join(pats), dtype=np.uint8).copy()
    a_off = np.zeros(len(pats) + 1, dtype=np.uint32)
    a_off[1:] = np.cumsum([len(p) for p in pats])
    tw.twin_compile_error.restype = ctypes.c_int
    tw.twin_compile_error.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                      ctypes.c_char_p, ctypes.c_uint32]
    for threads in (1, 2, 7):
        buf = ctypes.create_string_buffer(256)
        rc = tw.twin_compile_error(A.ctypes.data, a_off.ctypes.data, len(pats), threads, buf, 256)
        assert rc == -5 and buf.value.startswith(b"pattern 4000:"), (threads, rc, buf.value)
    # offsets that go backwards: the first such place is reported, unless a bad pattern comes first
    off2 = a_off.copy()
    off2[3001] = off2[3000] - 1
    for threads in (1, 5):
        buf = ctypes.create_string_buffer(256)
        rc = tw.twin_compile_error(A.ctypes.data, off2.ctypes.data, len(pats), threads, buf, 256)
        assert rc == -2 and b"not monotonic at pattern 3000" in buf.value, (threads, rc, buf.value)
