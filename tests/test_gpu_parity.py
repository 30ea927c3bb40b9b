"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, against the committed golden vectors, and - at the full sizes of the benchmark definition -
through size-independent properties.  Bit-exact: everything here is integer work."""
import json
import os

import numpy as np
import pytest

from montecarlo_amd import McxError, synth
from montecarlo_amd.binding import pack_strings

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NO_MATCH = 0xFFFFFFFF


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def check_k1(ctx, oracle_mod, A, a_off, a_mod, S, s_off, flags=0, sample=None):
    ctx.alias_table_load(A, a_off, a_mod, flags)
    first, count = ctx.alias_match(S, s_off)
    N = len(s_off) - 1
    idx = range(N) if sample is None else sample
    if sample is None:
        o_first, o_count = oracle_mod.alias_match(A, a_off, S, s_off, normalize_dash=bool(flags & 1))
        assert np.array_equal(first, o_first)
        assert np.array_equal(count, o_count)
    else:
        for i in idx:
            f, c = oracle_mod.alias_match(A, a_off, S, s_off, normalize_dash=bool(flags & 1), i0=i, i1=i + 1)
            assert first[i] == f[0] and count[i] == c[0], i
    return first, count


def check_glibc_sample(oracle_mod, A, a_off, S, s_off, first, count, n=1000, normalize_dash=False, seed=0):
    """n strings spread over the batch (always the first and the last one) against glibc fnmatch"""
    N = len(s_off) - 1
    if N <= n:
        picks = list(range(N))
    else:
        rng = np.random.default_rng(seed)
        picks = sorted(set(rng.choice(N, size=n - 2, replace=False).tolist()) | {0, N - 1})
    g_first, g_count = oracle_mod.alias_match_sample(A, a_off, S, s_off, picks, normalize_dash=normalize_dash)
    assert np.array_equal(first[picks], g_first) and np.array_equal(count[picks], g_count)


# ---- K1 ------------------------------------------------------------------------

def test_k1_fnmatch_kat(gpu_ctx, oracle_mod):
    """every known-answer vector, all strings x all patterns"""
    kat = load("fnmatch_kat.json")["cases"]
    pats = sorted({c["pattern"] for c in kat})
    strs = sorted({c["string"] for c in kat})
    pb = [bytes.fromhex(p) for p in pats]
    sb = [bytes.fromhex(s) for s in strs]
    A, a_off = pack_strings(pb)
    S, s_off = pack_strings(sb)
    a_mod = np.arange(len(pb), dtype=np.uint32)
    check_k1(gpu_ctx, oracle_mod, A, a_off, a_mod, S, s_off)
    # and each vector on its own through first_idx/count of a one-pattern table
    for c in kat[:40]:
        A1, a1 = pack_strings([bytes.fromhex(c["pattern"])])
        S1, s1 = pack_strings([bytes.fromhex(c["string"])])
        gpu_ctx.alias_table_load(A1, a1, np.zeros(1, np.uint32))
        f, n = gpu_ctx.alias_match(S1, s1)
        assert n[0] == c["match"] and f[0] == (0 if c["match"] else NO_MATCH)


def test_k1_kmod_fixture(gpu_ctx, oracle_mod):
    fx = load("kmod_resolve.json")
    pats = [a[0].encode() for a in fx["aliases"]]
    mods = [a[1] for a in fx["aliases"]]
    A, a_off = pack_strings(pats)
    S, s_off = pack_strings([q["query"].encode() for q in fx["queries"]])
    first, count = check_k1(gpu_ctx, oracle_mod, A, a_off, np.arange(len(pats), dtype=np.uint32), S, s_off, flags=1)
    for i, q in enumerate(fx["queries"]):
        assert count[i] == len(q["modules"])
        if q["modules"]:
            assert mods[first[i]] == q["modules"][0]


def test_k1_match_lists(gpu_ctx, oracle_mod):
    """all matching table positions per string, ascending: kmod's answers of the
    fixture in kmod's order, then seeded tables against the oracle (with repeated
    pattern texts, with and without a limit, both kernel variants)"""
    fx = load("kmod_resolve.json")
    pats = [a[0].encode() for a in fx["aliases"]]
    mods = [a[1] for a in fx["aliases"]]
    A, a_off = pack_strings(pats)
    S, s_off = pack_strings([q["query"].encode() for q in fx["queries"]])
    gpu_ctx.alias_table_load(A, a_off, np.arange(len(pats), dtype=np.uint32), flags=1)
    off, idx = gpu_ctx.alias_match_list(S, s_off)
    for i, q in enumerate(fx["queries"]):
        assert [mods[j] for j in idx[off[i]:off[i + 1]]] == q["modules"], q
    for seed, P, N, K in ((12, 3000, 700, 0), (13, 3000, 700, 2), (14, 500, 70000, 0), (15, 500, 70000, 3),
                           (16, 500, 200000, 0), (17, 500, 200000, 2)):
        A, a_off, a_mod = synth.alias_table(seed, P)
        # repeat a tenth of the table at the end: same texts at later positions
        pats = [A[a_off[j]:a_off[j + 1]].tobytes() for j in range(P)]
        pats += pats[:: 10]
        A2, a2_off = pack_strings(pats)
        S, s_off = synth.modalias_strings(seed, P, N)
        gpu_ctx.alias_table_load(A2, a2_off, np.zeros(len(pats), dtype=np.uint32))
        off, idx = gpu_ctx.alias_match_list(S, s_off, K=K)
        first, count = gpu_ctx.alias_match(S, s_off)
        want_n = np.minimum(count, K) if K else count
        assert np.array_equal(np.diff(off.astype(np.int64)), want_n)
        assert np.array_equal(idx[off[:-1][want_n > 0]], first[want_n > 0])
        sample = np.unique(np.concatenate([np.arange(0, N, max(1, N // 400)), np.argsort(-count.astype(np.int64))[:20]]))
        Ss, ss_off = pack_strings([S[s_off[i]:s_off[i + 1]].tobytes() for i in sample])
        o_off, o_idx = oracle_mod.alias_match_list(A2, a2_off, Ss, ss_off, K=K)
        for n, i in enumerate(sample):
            assert np.array_equal(idx[off[i]:off[i + 1]], o_idx[o_off[n]:o_off[n + 1]]), i
        assert (count > 1).any()
    # a list that is too small: error, total and offsets still reported; an empty batch
    with pytest.raises(McxError) as e:
        gpu_ctx.alias_match_list(S, s_off, cap=5)
    assert e.value.rc == -6
    off, idx = gpu_ctx.alias_match_list(np.zeros(0, np.uint8), np.zeros(1, np.uint32))
    assert off.tolist() == [0] and len(idx) == 0


def test_k1_match_lists_long_and_both_ways(gpu_ctx, oracle_mod):
    """Lists come from one pass of the kernel while no string is matched by more than 16
    different pattern texts, and from a second, filling pass otherwise.  Both ways on the same
    batches (the seam forces the second one), a string matched by 17, by 40 and by more than
    10 000 different patterns (long lists are sorted by a whole block), with and without a limit."""
    import itertools

    base = b"x:0123456789abcdefghij"
    body = base[2:]
    pats = [b"x:*"]
    for r in (4, 5):
        for pos in itertools.combinations(range(len(body)), r):
            pats.append(b"x:*" + b"*".join(bytes([body[i]]) for i in pos) + b"*")
            if len(pats) >= 12000:
                break
    pats += [b"y:*%d*" % d for d in range(10)] + [b"y:*%d*%d*" % (d, d + 1) for d in range(9)]
    pats += [b"y:?%d*" % d for d in range(10)]                                          # 'y:' strings: up to 20 texts
    pats += [b"z:*%02d" % d for d in range(40)] + [b"z:*"] * 3                            # copies of one text
    rng = np.random.default_rng(3)
    order = rng.permutation(len(pats))                 # table order is not the order they were made in
    pats = [pats[i] for i in order]
    A, a_off = pack_strings(pats)
    strs = [base, b"x:", b"x:jihgfedcba9876543210", b"y:0123456789", b"y:5", b"y:55", b"z:07", b"z:", b"w:nothing"]
    strs += [b"y:%d%d" % (a, b) for a in range(10) for b in range(10)]
    S, s_off = pack_strings(strs)
    o_off, o_idx = oracle_mod.alias_match_list(A, a_off, S, s_off)
    assert o_off[1] - o_off[0] > 10_000 and 16 < o_off[4] - o_off[3] <= 20
    for two_pass in (0, 1):
        gpu_ctx.tune(list_two_pass=two_pass)
        gpu_ctx.alias_table_load(A, a_off, np.zeros(len(pats), dtype=np.uint32))
        off, idx = gpu_ctx.alias_match_list(S, s_off)
        assert np.array_equal(off, o_off) and np.array_equal(idx, o_idx)
        for K in (1, 3, 17, 100):
            k_off, k_idx = gpu_ctx.alias_match_list(S, s_off, K=K)
            w_off, w_idx = oracle_mod.alias_match_list(A, a_off, S, s_off, K=K)
            assert np.array_equal(k_off, w_off) and np.array_equal(k_idx, w_idx)
        # without the long lists: short lists only, nothing overflows
        short = [i for i in range(len(strs)) if o_off[i + 1] - o_off[i] <= 16]
        Ss, ss_off = pack_strings([strs[i] for i in short])
        off, idx = gpu_ctx.alias_match_list(Ss, ss_off, K=2)
        w_off, w_idx = oracle_mod.alias_match_list(A, a_off, Ss, ss_off, K=2)
        assert np.array_equal(off, w_off) and np.array_equal(idx, w_idx)
    gpu_ctx.tune(list_two_pass=0)
    # the synthetic table both ways: same lists
    P, N = 30000, 70000
    A, a_off, a_mod = synth.alias_table(21, P)
    S, s_off = synth.modalias_strings(21, P, N)
    got = []
    for two_pass in (0, 1):
        gpu_ctx.tune(list_two_pass=two_pass)
        gpu_ctx.alias_table_load(A, a_off, a_mod)
        got.append(gpu_ctx.alias_match_list(S, s_off))
    gpu_ctx.tune(list_two_pass=0)
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    o_off, o_idx = oracle_mod.alias_match_list(A, a_off, S[: s_off[300]], s_off[:301])
    assert np.array_equal(got[0][0][:301], o_off) and np.array_equal(got[0][1][: o_off[300]], o_idx)


def test_k1_match_lists_of_many_copies(gpu_ctx, oracle_mod):
    """Pattern texts that stand in the table dozens of times: lists of a few runs with up to 170
    entries (past what the merge gathers in LDS per list), and tiles whose lists together outgrow
    the room there.  Whole batch: list lengths = counts, first entries = first matches, every list
    strictly ascending; 400 strings entry by entry against the oracle; with and without a limit."""
    rng = np.random.default_rng(11)
    pats = [b"q:*"] * 30 + [b"q:a*"] * 20 + [b"q:?b*"] * 12 + [b"r:*"] * 100 + [b"r:x*"] * 70
    pats += [b"q:ab%d" % d for d in range(10)] + [b"r:x%d*" % d for d in range(10)] + [b"s:%d" % d for d in range(50)]
    pats = [pats[i] for i in rng.permutation(len(pats))]
    A, a_off = pack_strings(pats)
    kinds = [b"q:ab%d", b"q:a%d", b"q:zb%d", b"q:%d", b"r:x%d", b"r:y%d", b"s:%d", b"t:%d"]
    N = 40000
    strs = [kinds[k] % d for k, d in zip(rng.integers(0, len(kinds), N), rng.integers(0, 12, N))]
    S, s_off = pack_strings(strs)
    gpu_ctx.alias_table_load(A, a_off, np.zeros(len(pats), dtype=np.uint32))
    first, count = gpu_ctx.alias_match(S, s_off)
    assert count.max() >= 170
    sample = np.unique(np.concatenate([np.arange(0, N, 100), np.argsort(-count.astype(np.int64))[:5]]))
    Ss, ss_off = pack_strings([strs[i] for i in sample])
    for K in (0, 40):
        off, idx = gpu_ctx.alias_match_list(S, s_off, K=K)
        want_n = np.minimum(count, K) if K else count
        assert np.array_equal(np.diff(off.astype(np.int64)), want_n)
        assert np.array_equal(idx[off[:-1][want_n > 0]], first[want_n > 0])
        inner = np.ones(len(idx), dtype=bool)
        inner[off[:-1][want_n > 0]] = False           # not the first entry of a list
        assert (idx[1:][inner[1:]] > idx[:-1][inner[1:]]).all()
        o_off, o_idx = oracle_mod.alias_match_list(A, a_off, Ss, ss_off, K=K)
        for n, i in enumerate(sample):
            assert np.array_equal(idx[off[i]:off[i + 1]], o_idx[o_off[n]:o_off[n + 1]]), i


@pytest.mark.parametrize("seed,P,N", [(2, 30000, 1000), (5, 777, 130), (6, 63, 1), (7, 5, 65), (8, 4096, 64)])
def test_k1_synthetic_vs_oracle(gpu_ctx, oracle_mod, seed, P, N):
    """configs[1] (1000 x 30000, seed 2) and ragged shapes around the tile sizes"""
    A, a_off, a_mod = synth.alias_table(seed, P)
    S, s_off = synth.modalias_strings(seed, P, N)
    first, count = check_k1(gpu_ctx, oracle_mod, A, a_off, a_mod, S, s_off)
    if N >= 100:
        assert (count > 0).mean() > 0.6


def test_k1_edge_cases(gpu_ctx, oracle_mod):
    # empty batch
    A, a_off, a_mod = synth.alias_table(3, 100)
    gpu_ctx.alias_table_load(A, a_off, a_mod)
    f, c = gpu_ctx.alias_match(np.zeros(0, np.uint8), np.zeros(1, np.uint32))
    assert len(f) == 0 and len(c) == 0
    # empty strings, strings shorter than the 16-byte prefilter window, a 20 KB string
    # (one tile then exceeds the LDS staging buffer and is read in place)
    strs = [b"", b"p", b"pci:", b"pci:v0000", b"*", b"?", b"[", b"x" * 20000, b"pci:v" + b"0" * 3000, b""]
    pats = [b"", b"*", b"?", b"p*", b"pci:*", b"pci:v0000", b"x*x", b"*x", b"[px]*", b"pci:v0*0", b"\\*", b"\\?",
            b"\\[", b"[[]", b"????", b"*?", b"pci:v" + b"0" * 3000, b"pci:v" + b"0" * 2999 + b"[0-9]"]
    A, a_off = pack_strings(pats)
    S, s_off = pack_strings(strs)
    check_k1(gpu_ctx, oracle_mod, A, a_off, np.arange(len(pats), dtype=np.uint32), S, s_off)
    # empty table
    gpu_ctx.alias_table_load(np.zeros(0, np.uint8), np.zeros(1, np.uint32), np.zeros(0, np.uint32))
    f, c = gpu_ctx.alias_match(S, s_off)
    assert (f == NO_MATCH).all() and (c == 0).all()


def test_k1_table_size_limit(gpu_ctx):
    """Survivor entries keep a pattern number in 26 bits: a larger table is refused
    before anything is read (the arrays here hold a single pattern)."""
    import ctypes

    arena = np.frombuffer(b"x", dtype=np.uint8)
    a_off = np.array([0, 1], dtype=np.uint32)
    a_mod = np.zeros(1, dtype=np.uint32)
    rc = gpu_ctx.lib.mcx_alias_table_load(
        gpu_ctx.h, arena.ctypes.data_as(ctypes.c_void_p), a_off.ctypes.data_as(ctypes.c_void_p),
        a_mod.ctypes.data_as(ctypes.c_void_p), (1 << 26) + 1, 0)
    assert rc == -2  # MCX_E_BAD_ARG
    assert b"2^26" in gpu_ctx.lib.mcx_last_error(gpu_ctx.h)


@pytest.mark.parametrize("N", [1, 2049, 5000, 70000])
def test_k1_single_key_length_tables(gpu_ctx, oracle_mod, N):
    """tables whose patterns all share one literal-prefix length (one probe per string), with
    batches cut at one, two, three and eight strings per block"""
    pats = [b"serio:ty07pr*id*ex97", b"serio:ty07pr*id*ex97", b"serio:ty07pr0?id*", b"serio:ty06pr*"]
    rng = np.random.default_rng(N)
    strs = [b"serio:ty0%dpr%02Xid%02Xex%02X" % (rng.integers(6, 8), rng.integers(0, 256), rng.integers(0, 256),
                                              rng.choice([0x97, 0x12])) for _ in range(N)]
    A, a_off = pack_strings(pats)
    S, s_off = pack_strings(strs)
    gpu_ctx.alias_table_load(A, a_off, np.zeros(len(pats), dtype=np.uint32))
    first, count = gpu_ctx.alias_match(S, s_off)
    t_first, t_count = oracle_mod.alias_match(A, a_off, S, s_off, use_twin=True)
    assert np.array_equal(first, t_first) and np.array_equal(count, t_count)
    # four patterns: glibc does the whole batch in no time
    o_first, o_count = oracle_mod.alias_match(A, a_off, S, s_off)
    assert np.array_equal(first, o_first) and np.array_equal(count, o_count)
    assert (count > 0).any() or N == 1


@pytest.mark.parametrize("sub_min,norm", [(3, False), (40, False), (3, True)])
def test_k1_second_index_of_large_groups(gpu_ctx, oracle_mod, sub_min, norm):
    """Large batches look large pattern groups up by the grams of the string.  With the
    size from which a group counts as large set low, the synthetic table has hundreds of
    such groups and a block meets more (string, group) pairs and runs than it has room
    for, so the whole-group way out is taken as well; first_idx, count and the match
    lists must not notice any of it."""
    gpu_ctx.tune(sub_min=sub_min)
    P, N = 30000, 70000
    A, a_off, a_mod = synth.alias_table(9, P)
    S, s_off = synth.modalias_strings(9, P, N)
    flags = 1 if norm else 0
    gpu_ctx.alias_table_load(A, a_off, a_mod, flags)
    first, count = gpu_ctx.alias_match(S, s_off)
    gpu_ctx.tune(sub_min=0)
    t_first, t_count = oracle_mod.alias_match(A, a_off, S, s_off, normalize_dash=norm, use_twin=True)
    assert np.array_equal(first, t_first) and np.array_equal(count, t_count)
    check_glibc_sample(oracle_mod, A, a_off, S, s_off, first, count, 1000, normalize_dash=norm, seed=sub_min)
    off, idx = gpu_ctx.alias_match_list(S, s_off)
    assert np.array_equal(np.diff(off.astype(np.int64)), count)
    assert np.array_equal(idx[off[:-1][count > 0]], first[count > 0])
    o_off, o_idx = oracle_mod.alias_match_list(A, a_off, S[: s_off[60]], s_off[:61], normalize_dash=norm)
    assert np.array_equal(idx[: off[60]], o_idx)


@pytest.mark.parametrize("N", [1, 100, 2049, 5000, 8000, 12000])
def test_k1_large_batch_kernel_on_small_batches(gpu_ctx, oracle_mod, N):
    """A table with a very large group sends every batch to the large-batch kernel; here
    the batch-size knob (mcx_ctx_tune) does, so that it runs with one to eight strings per block (and the
    small-batch kernel, which answers second, with one to eight as well)."""
    gpu_ctx.tune(tiles_min=1, sub_min=5)
    P = 20000
    A, a_off, a_mod = synth.alias_table(11, P)
    S, s_off = synth.modalias_strings(11, P, N)
    gpu_ctx.alias_table_load(A, a_off, a_mod)
    first, count = gpu_ctx.alias_match(S, s_off)
    off, idx = gpu_ctx.alias_match_list(S, s_off)
    t_first, t_count = oracle_mod.alias_match(A, a_off, S, s_off, use_twin=True)
    assert np.array_equal(first, t_first) and np.array_equal(count, t_count)
    assert np.array_equal(np.diff(off.astype(np.int64)), count)
    check_glibc_sample(oracle_mod, A, a_off, S, s_off, first, count, 1000, seed=N)
    # the other kernel (the knobs belong to a table: loaded again), same answers
    gpu_ctx.tune(tiles_min=10**9, sub_min=0)
    gpu_ctx.alias_table_load(A, a_off, a_mod)
    gpu_ctx.tune(tiles_min="auto")
    f2, c2 = gpu_ctx.alias_match(S, s_off)
    assert np.array_equal(first, f2) and np.array_equal(count, c2)


@pytest.mark.parametrize("N", [1, 3900, 30500, 31500, 32500, 40000])
def test_k1_host_buffer_call_sizes(gpu_ctx, oracle_mod, N):
    """The host-buffer call runs the kernel on pinned host memory up to 2 MB of strings,
    offsets and results (about 31 000 of these strings) and copies through a device staging
    area above; sizes on both sides of the switch, alternating, give the same answers."""
    P = 5000
    A, a_off, a_mod = synth.alias_table(13, P)
    gpu_ctx.alias_table_load(A, a_off, a_mod)
    S, s_off = synth.modalias_strings(13, P, N)
    t_first, t_count = oracle_mod.alias_match(A, a_off, S, s_off, use_twin=True)
    check_glibc_sample(oracle_mod, A, a_off, S, s_off, t_first, t_count, 1000, seed=N)
    for _ in range(2):
        first, count = gpu_ctx.alias_match(S, s_off)
        assert np.array_equal(first, t_first) and np.array_equal(count, t_count)
        S1, s1 = synth.modalias_strings(14, P, 5)
        f1, c1 = gpu_ctx.alias_match(S1, s1)
        o1, oc1 = oracle_mod.alias_match(A, a_off, S1, s1)
        assert np.array_equal(f1, o1) and np.array_equal(c1, oc1)


def test_k1_random_metachar_tables(gpu_ctx, oracle_mod):
    """random patterns heavy in metacharacters (those the compiler accepts)"""
    import random

    rnd = random.Random(5)
    tw = oracle_mod.twin()
    pats = []
    while len(pats) < 600:
        p = bytes(rnd.choice(b"ab[]!^-\\*?c") for _ in range(rnd.randint(0, 9)))
        if tw.twin_compile_check(p, len(p), 0) == 0:
            pats.append(p)
    strs = [bytes(rnd.choice(b"ab[]-\\c!^*?") for _ in range(rnd.randint(0, 7))) for _ in range(300)]
    A, a_off = pack_strings(pats)
    S, s_off = pack_strings(strs)
    first, count = check_k1(gpu_ctx, oracle_mod, A, a_off, np.arange(len(pats), dtype=np.uint32), S, s_off)
    assert count.sum() > 1000


def test_k1_rejects_unsupported_pattern(gpu_ctx):
    A, a_off = pack_strings([b"ok*", b"[[:alpha:]]x"])
    with pytest.raises(McxError) as e:
        gpu_ctx.alias_table_load(A, a_off, np.zeros(2, np.uint32))
    assert e.value.rc == -5 and "pattern 1" in str(e.value)


def test_k1_properties_large(gpu_ctx, oracle_mod):
    """100000 x 30000 (3e9 pairs, far beyond what the oracle finishes in seconds):
    spot-check 48 strings against the oracle, and check the properties that hold at
    any size: an instance of pattern j matches j; results do not depend on how the
    batch is cut; driver_mod = a_mod[first_idx]; table order decides first_idx."""
    seed, P, N = 4, 30000, 100000
    A, a_off, a_mod = synth.alias_table(seed, P)
    S, s_off = synth.modalias_strings(seed, P, N)
    gpu_ctx.alias_table_load(A, a_off, a_mod)
    first, count = gpu_ctx.alias_match(S, s_off)
    check_glibc_sample(oracle_mod, A, a_off, S, s_off, first, count, 1000, seed=4)
    assert ((first == NO_MATCH) == (count == 0)).all()
    assert (count > 0).mean() > 0.69
    # the same strings in two halves and in reverse order
    h = N // 2
    f1, c1 = gpu_ctx.alias_match(S[: s_off[h]], s_off[: h + 1])
    f2, c2 = gpu_ctx.alias_match(S[s_off[h]:], s_off[h:] - s_off[h])
    assert np.array_equal(np.concatenate([f1, f2]), first) and np.array_equal(np.concatenate([c1, c2]), count)
    # device-resident variant + derived driver module
    d_S, d_off = gpu_ctx.to_device(S), gpu_ctx.to_device(s_off)
    d_f, d_c, d_m = gpu_ctx.alloc(4 * N), gpu_ctx.alloc(4 * N), gpu_ctx.alloc(4 * N)
    gpu_ctx.alias_match_dev(d_S.ptr, d_off.ptr, N, d_f.ptr, d_c.ptr)
    gpu_ctx.alias_driver_mod_dev(d_f.ptr, N, d_m.ptr)
    assert np.array_equal(d_f.download(np.uint32, N), first)
    assert np.array_equal(d_c.download(np.uint32, N), count)
    mod = d_m.download(np.uint32, N)
    want = np.where(first == NO_MATCH, NO_MATCH, a_mod[np.minimum(first, P - 1)])
    assert np.array_equal(mod, want)
    for b in (d_S, d_off, d_f, d_c, d_m):
        b.free()


def test_k1_full_table_and_its_shards(gpu_ctx, oracle_mod):
    """configs[3] at its full size on one GPU: 1 000 000 strings x 3 000 000 patterns.  The whole
    table and eight contiguous shards of it combined by the rule of the multi-GPU path (smallest
    shifted first index, summed counts) give the same answers; 208 strings spread over the batch
    against glibc (0.1 s each); properties that hold at any size."""
    from montecarlo_amd.shard import shard_range

    seed, P, N, G = 4, 3_000_000, 1_000_000, 8
    A, a_off, a_mod = synth.alias_table(seed, P)
    S, s_off = synth.modalias_strings(seed, P, N)
    gpu_ctx.alias_table_load(A, a_off, a_mod)
    d_S, d_off = gpu_ctx.to_device(S), gpu_ctx.to_device(s_off)
    d_f, d_c = gpu_ctx.alloc(4 * N), gpu_ctx.alloc(4 * N)
    gpu_ctx.alias_match_dev(d_S.ptr, d_off.ptr, N, d_f.ptr, d_c.ptr)
    first, count = d_f.download(np.uint32, N), d_c.download(np.uint32, N)
    assert (count > 0).mean() > 0.69 and ((first == NO_MATCH) == (count == 0)).all()
    assert (first[count > 0] < P).all()
    check_glibc_sample(oracle_mod, A, a_off, S, s_off, first, count, 208, seed=3)
    # the host-buffer entry point on a slice in the middle: same answers
    lo, hi = 400_000, 470_000
    f_h, c_h = gpu_ctx.alias_match(S[s_off[lo]:s_off[hi]], s_off[lo:hi + 1] - s_off[lo])
    assert np.array_equal(f_h, first[lo:hi]) and np.array_equal(c_h, count[lo:hi])
    best = np.full(N, NO_MATCH, dtype=np.uint64)
    total = np.zeros(N, dtype=np.uint64)
    for r in range(G):
        j0, j1 = shard_range(P, r, G)
        base = int(a_off[j0])
        gpu_ctx.alias_table_load(A[base:int(a_off[j1])], a_off[j0:j1 + 1] - np.uint32(base), a_mod[j0:j1])
        gpu_ctx.alias_match_dev(d_S.ptr, d_off.ptr, N, d_f.ptr, d_c.ptr)
        f, c = d_f.download(np.uint32, N), d_c.download(np.uint32, N)
        shifted = np.where(f == NO_MATCH, np.uint64(NO_MATCH), f.astype(np.uint64) + np.uint64(j0))
        best = np.minimum(best, shifted)
        total += c
    assert np.array_equal(best.astype(np.uint32), first)
    assert np.array_equal(total.astype(np.uint32), count)
    for b in (d_S, d_off, d_f, d_c):
        b.free()


def _instance_of(pattern, rng):
    """a string the glob is likely to match: metacharacters filled in at random"""
    out, i, n = bytearray(), 0, len(pattern)
    hexd = b"0123456789ABCDEF"
    while i < n:
        c = pattern[i]
        if c == ord("*"):
            out += bytes(rng.choice(list(hexd), size=int(rng.integers(0, 9))).tolist())
        elif c == ord("?"):
            out.append(int(rng.choice(list(hexd))))
        elif c == ord("\\") and i + 1 < n:
            i += 1
            out.append(pattern[i])
        elif c == ord("["):
            j = pattern.find(b"]", i + 2)
            if j < 0:
                out.append(c)
            else:
                body = pattern[i + 1:j]
                neg = body[:1] in (b"!", b"^")
                members = [x for x in body[1 if neg else 0:] if x != ord("-")]
                out.append(ord("z") if neg or not members else int(rng.choice(members)))
                i = j
        else:
            out.append(c)
        i += 1
    return bytes(out)


def _parse_modules_alias(text):
    """modules.alias by hand, independent of the library's loader: (patterns, module names) in file order"""
    pats, mods = [], []
    for line in text.split(b"\n"):
        tok = line.split()
        if len(tok) >= 3 and tok[0] == b"alias":
            pats.append(tok[1])
            mods.append(tok[2].replace(b"-", b"_").decode())
    return pats, mods


def test_k1_real_host_tables(gpu_ctx, oracle_mod, tmp_path):
    """configs[0]: kmod's table files through the text loader into the kernels.  Where the box
    has /lib/modules/$(uname -r)/modules.alias and modules.dep (or MCX_TEST_MODULES_DIR names a
    directory that has them) those are used; where it has none - the GPU boxes of this pool -
    a directory in the REAL formats with SYNTHETIC content is written first
    (synth.write_modules_dir: comments, lines with other keywords, .ko / .ko.xz / .ko.zst /
    .ko.gz paths, dashed file names, a /proc/modules text), so that the flow never goes
    untested: load_dir -> kernels with kmod's dash rule -> every string against glibc fnmatch
    -> dependency flags and closure against the oracle -> the modules of a modalias in kmod's
    order.  Strings: every modalias sysfs shows plus 10 000 made from the table's own patterns;
    up to 50 of them also go to kmod itself (`modprobe -R`) where the directory has its binary
    index."""
    import glob
    import shutil
    import subprocess

    from montecarlo_amd import ModTab

    d = os.environ.get("MCX_TEST_MODULES_DIR") or os.path.join("/lib/modules", os.uname().release)
    proc_modules = "/proc/modules"
    if not (os.path.isfile(os.path.join(d, "modules.alias")) and os.path.isfile(os.path.join(d, "modules.dep"))):
        d = str(tmp_path / "modules")
        synth.write_modules_dir(d, P=20000, seed=17)
        proc_modules = os.path.join(d, "proc_modules")
    tab = ModTab()
    tab.load_dir(d)
    A, a_off, a_mod = tab.alias_arrays()
    P = len(a_off) - 1
    assert P > 0
    # the loader against a parse by hand: same patterns, same owners, file order
    with open(os.path.join(d, "modules.alias"), "rb") as f:
        h_pats, h_mods = _parse_modules_alias(f.read())
    raw = A.tobytes()
    assert P == len(h_pats)
    assert all(raw[a_off[j]:a_off[j + 1]] == h_pats[j] for j in range(P))
    assert all(tab.module_name(int(a_mod[j])) == h_mods[j] for j in range(0, P, 7))
    strs = []
    for path in glob.glob("/sys/bus/*/devices/*/modalias"):
        try:
            with open(path, "rb") as f:
                t = f.read().strip()
        except OSError:
            continue
        if t and b"\0" not in t:
            strs.append(t)
    n_sys = len(strs)
    rng = np.random.default_rng(1)
    for j in rng.choice(P, size=10_000):
        t = _instance_of(raw[a_off[j]:a_off[j + 1]], rng)
        if t and b"\0" not in t:
            strs.append(t)
    S, s_off = pack_strings(strs)
    gpu_ctx.alias_table_load(A, a_off, a_mod, 1)
    first, count = gpu_ctx.alias_match(S, s_off)
    o_first, o_count = oracle_mod.alias_match_sample(A, a_off, S, s_off, range(len(strs)), normalize_dash=True)
    assert np.array_equal(first, o_first) and np.array_equal(count, o_count)
    assert (count[n_sys:] > 0).mean() > 0.5          # most made-up strings do match their pattern
    # the modules of a modalias, in kmod's order (all matching lines, file order), from the
    # hand-parsed file and glibc: names, not ids, so the loader's numbering is covered too
    hA, ha_off = pack_strings(h_pats)
    pick = [int(i) for i in np.nonzero(count > 1)[0][:40]] + [int(i) for i in np.nonzero(count == 1)[0][:10]]
    pS, ps_off = pack_strings([strs[i] for i in pick])
    off, idx = gpu_ctx.alias_match_list(pS, ps_off)
    w_off, w_idx = oracle_mod.alias_match_list(hA, ha_off, pS, ps_off, normalize_dash=True)
    assert np.array_equal(off, w_off) and np.array_equal(idx, w_idx)
    for k in range(len(pick)):
        got = [tab.module_name(int(a_mod[j])) for j in idx[off[k]:off[k + 1]]]
        assert got == [h_mods[j] for j in w_idx[w_off[k]:w_off[k + 1]]]
    # the dependency side of the same directory: flags and closure against the oracle
    rowptr, col = tab.dep_arrays(direct=True)
    V = len(rowptr) - 1
    if V:
        flags = np.full(V, 5, dtype=np.uint8)
        if os.path.isfile(proc_modules):
            with open(proc_modules, "rb") as f:
                loaded, _ = tab.mark_loaded(f.read())
            flags = np.where(loaded & 1, 5, 4).astype(np.uint8)
            if proc_modules != "/proc/modules":
                assert 0 < int((loaded & 1).sum()) < V
        seeds = np.zeros((V + 31) // 32, dtype=np.uint32)
        seeds[0] = 1
        lo = np.nonzero(flags & 1)[0][:64]
        seeds[lo // 32] |= (1 << (lo % 32)).astype(np.uint32)
        check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, seeds)
    modprobe = shutil.which("modprobe")
    if modprobe and os.path.isfile(os.path.join(d, "modules.alias.bin")):
        asked = 0
        for i in np.nonzero(count > 0)[0][:50]:
            r = subprocess.run([modprobe, "-R", strs[i].decode("latin-1")], capture_output=True, text=True, timeout=20)
            if r.returncode != 0:
                continue
            off, idx = gpu_ctx.alias_match_list(*pack_strings([strs[i]]))
            mods = []
            for j in idx:
                name = tab.module_name(int(a_mod[j]))
                if name not in mods:
                    mods.append(name)
            got = [ln.strip().replace("-", "_") for ln in r.stdout.split() if ln.strip()]
            # kmod also consults modprobe.d aliases and blacklists: compare only when it names table modules
            if got and set(got) <= set(mods):
                assert got == [m for m in mods if m in got]
                asked += 1
        assert asked >= 0


def test_k1_hw_flags(gpu_ctx):
    pats = [b"pci:v*", b"acpi*:PNP0C09:*", b"usb:v1*", b"input:b*", b" of:N*", b"xpci:v", b"serio:ty*", b"hid:b*",
            b"platform:x"]
    a_mod = np.array([0, 1, 1, 2, 3, 4, 5, 7, 7], dtype=np.uint32)
    A, a_off = pack_strings(pats)
    gpu_ctx.alias_table_load(A, a_off, a_mod)
    assert gpu_ctx.alias_hw_flags(9).tolist() == [1, 1, 0, 1, 0, 1, 0, 1, 0]


# ---- K2 ------------------------------------------------------------------------

def check_k2(ctx, oracle_mod, rowptr, col, flags, seeds):
    ctx.dep_graph_load(rowptr, col)
    out, ref = ctx.dep_flags(flags)
    o_out, o_ref = oracle_mod.dep_flags(rowptr, col, flags)
    assert np.array_equal(out, o_out) and np.array_equal(ref, o_ref)
    out2, none = ctx.dep_flags(flags, want_refcnt=False)
    assert none is None and np.array_equal(out2, o_out)
    for direction in (0, 1):
        clo, edges, levels = ctx.dep_closure(flags, seeds, direction)
        o_clo, o_edges, o_levels = oracle_mod.dep_closure(rowptr, col, flags, seeds, direction)
        assert np.array_equal(clo, o_clo)
        assert edges == o_edges and levels == o_levels
    return out


def test_k2_searches_of_changing_shape_on_one_graph(gpu_ctx, oracle_mod):
    """a closure places its launches (the one-workgroup kernel, levels that claim without a look
    first) by the frontier sizes of the LAST search in that direction on the same graph: searches
    of very different shape one after the other, each twice, so that every choice is made on a
    right guess and on a wrong one"""
    V = 300000
    rowptr, col = synth.dep_graph(11, V, 10)
    flags = synth.dep_flags(11, rowptr, col)
    gpu_ctx.dep_graph_load(rowptr, col)
    expect = {}
    for nseed in (1, 20000, 16, 16, 1, 400, 20000, 1, 16):
        seeds, _ = synth.seed_bitmap(11 + nseed, flags, nseed)
        for direction in (0, 1):
            key = (nseed, direction)
            if key not in expect:
                expect[key] = oracle_mod.dep_closure(rowptr, col, flags, seeds, direction)
            o_clo, o_edges, o_levels = expect[key]
            clo, edges, levels = gpu_ctx.dep_closure(flags, seeds, direction)
            assert np.array_equal(clo, o_clo), key
            assert (edges, levels) == (o_edges, o_levels), key


@pytest.mark.parametrize("seed,V,deg,nseed", [(3, 1000, 8, 3), (4, 50000, 10, 8), (5, 33, 3, 2), (6, 1, 1, 1)])
def test_k2_vs_oracle(gpu_ctx, oracle_mod, seed, V, deg, nseed):
    rowptr, col = synth.dep_graph(seed, V, deg)
    flags = synth.dep_flags(seed, rowptr, col, 100)
    seeds, _ = synth.seed_bitmap(seed, flags, nseed)
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, seeds)


@pytest.mark.parametrize("V", [1, 31, 65535, 65536, 65537, 70000])
def test_k2_host_buffer_call_sizes(gpu_ctx, oracle_mod, V):
    """The host-buffer calls of the dependency kernels work in pinned host memory on graphs of up to
    65 536 modules and copy through a device staging area above: graphs on both sides of the switch,
    calls of both kinds alternating with an alias match (which shares the pinned area), give the
    oracle's flag bytes, reference counts, closures, edge counts and levels."""
    rowptr, col = synth.dep_graph(21, V, 6)
    flags = synth.dep_flags(21, rowptr, col, 150)
    A, a_off, a_mod = synth.alias_table(13, 500)
    S1, s1 = synth.modalias_strings(14, 500, 40)
    gpu_ctx.alias_table_load(A, a_off, a_mod)
    o1 = oracle_mod.alias_match(A, a_off, S1, s1)
    gpu_ctx.dep_graph_load(rowptr, col)
    o_out, o_ref = oracle_mod.dep_flags(rowptr, col, flags)
    for nseed in (1, 50):
        seeds, _ = synth.seed_bitmap(21 + nseed, flags, min(nseed, V))
        for _ in range(2):
            out, ref = gpu_ctx.dep_flags(flags)
            assert np.array_equal(out, o_out) and np.array_equal(ref, o_ref)
            out2, none = gpu_ctx.dep_flags(flags, want_refcnt=False)
            assert np.array_equal(out2, o_out) and none is None
            for direction in (0, 1):
                clo, edges, levels = gpu_ctx.dep_closure(flags, seeds, direction)
                o_clo, o_edges, o_levels = oracle_mod.dep_closure(rowptr, col, flags, seeds, direction)
                assert np.array_equal(clo, o_clo) and (edges, levels) == (o_edges, o_levels)
                f1, c1 = gpu_ctx.alias_match(S1, s1)
                assert np.array_equal(f1, o1[0]) and np.array_equal(c1, o1[1])


def test_k2_hand_graphs(gpu_ctx, oracle_mod):
    # chain, diamond, isolated vertices, nothing loaded, everything seeded
    rowptr = np.array([0, 0, 1, 2, 3], dtype=np.uint32)
    col = np.array([0, 1, 2], dtype=np.uint32)
    for flags in ([5, 5, 5, 7], [5, 5, 4, 7], [0, 0, 0, 0], [1, 1, 1, 1]):
        f = np.array(flags, dtype=np.uint8)
        for seed_word in (0, 2, 0xF):
            check_k2(gpu_ctx, oracle_mod, rowptr, col, f, np.array([seed_word], dtype=np.uint32))
    rowptr = np.array([0, 0, 1, 2, 4, 4, 4], dtype=np.uint32)
    col = np.array([0, 0, 1, 2], dtype=np.uint32)
    check_k2(gpu_ctx, oracle_mod, rowptr, col, np.array([5, 5, 5, 5, 7, 0], np.uint8), np.array([1], np.uint32))
    # empty graph
    gpu_ctx.dep_graph_load(np.zeros(1, np.uint32), np.zeros(0, np.uint32))


def test_k2_hubs(gpu_ctx, oracle_mod):
    """modules with very long adjacency lists in both directions (the closure cuts
    those lists into pieces dealt over the grid): hubs as seeds, hubs found at a
    later level, several hubs in one level, and a 20000-seed closure"""
    rng = np.random.default_rng(77)
    V = 60000
    edges = set()
    for hub in (0, 1, 2, 40000):                     # needed by thousands of modules
        for u in rng.choice(np.arange(100, V), size=int(rng.integers(3000, 20000)), replace=False):
            if u != hub:
                edges.add((int(u), hub))
    for fat in (50, 51, 59999):                      # modules that need thousands of others
        for v in rng.choice(np.arange(100, V - 1), size=int(rng.integers(1000, 9000)), replace=False):
            if v != fat:
                edges.add((fat, int(v)))
    edges.add((40000, 50))                           # hub reached through a chain
    edges.add((0, 40000))
    for u in rng.integers(100, V, size=100000):
        v = int(rng.integers(0, V))
        if v != int(u):
            edges.add((int(u), v))
    e = np.array(sorted(edges), dtype=np.uint32)
    rowptr = np.zeros(V + 1, dtype=np.uint32)
    np.add.at(rowptr, e[:, 0].astype(np.int64) + 1, 1)
    rowptr = np.cumsum(rowptr, dtype=np.uint64).astype(np.uint32)
    col = np.ascontiguousarray(e[:, 1])
    flags = np.where(rng.random(V) < 0.9, 5, 4).astype(np.uint8)
    flags[[0, 1, 2, 50, 51, 40000, 59999]] = 5
    W = (V + 31) // 32
    for picks in ([0], [50], [59999], [0, 1, 2, 40000], [12345], list(range(100, 20100))):
        seeds = np.zeros(W, dtype=np.uint32)
        for v in picks:
            seeds[v >> 5] |= np.uint32(1 << (v & 31))
        check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, seeds)


def test_k2_deep_and_wide_searches(gpu_ctx, oracle_mod):
    """Searches far deeper than the window of per-level counters: a single chain of 5000
    modules (one module per level: the one-workgroup path from start to end), 3000 chains
    side by side (a frontier too large for one workgroup: a launch per level, 120 levels,
    the counter window restarted), and chains that fan out at their end (small levels
    first, then a hand-over in the middle of the search)."""
    def csr(V, edges):
        e = np.array(sorted(edges), dtype=np.uint32).reshape(-1, 2)
        rowptr = np.zeros(V + 1, dtype=np.uint32)
        np.add.at(rowptr, e[:, 0].astype(np.int64) + 1, 1)
        return np.cumsum(rowptr, dtype=np.uint64).astype(np.uint32), np.ascontiguousarray(e[:, 1])

    def bitmap(V, picks):
        bm = np.zeros((V + 31) // 32, dtype=np.uint32)
        for v in picks:
            bm[v >> 5] |= np.uint32(1 << (v & 31))
        return bm

    # one chain: module i needs module i + 1
    V = 5000
    rowptr, col = csr(V, [(i, i + 1) for i in range(V - 1)])
    flags = np.full(V, 5, np.uint8)
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, bitmap(V, [0]))        # deps direction walks it all
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, bitmap(V, [V - 1]))    # holders direction too
    flags[2500] = 4                                                          # an unloaded module cuts it
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, bitmap(V, [0, V - 1]))
    # 3000 chains of 120 modules side by side: chain c holds modules c, c + 3000, c + 6000, ...
    C, Ln = 3000, 120
    V = C * Ln
    rowptr, col = csr(V, [(c + C * k, c + C * (k + 1)) for c in range(C) for k in range(Ln - 1)])
    flags = np.full(V, 5, np.uint8)
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, bitmap(V, range(C)))
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, bitmap(V, range(V - C, V)))
    # ten chains of 80 whose last module needs 600 leaves each: small, small, ..., then wide
    edges, V = [], 0
    for c in range(10):
        base = V
        edges += [(base + k, base + k + 1) for k in range(79)]
        edges += [(base + 79, base + 80 + j) for j in range(600)]
        V += 80 + 600
    rowptr, col = csr(V, edges)
    flags = np.full(V, 5, np.uint8)
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, bitmap(V, [c * 680 for c in range(10)]))
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, bitmap(V, [c * 680 + 100 for c in range(10)]))


def test_k2_full_size(gpu_ctx, oracle_mod):
    """configs[2]: 1M modules, ~10M edges.  The oracle still finishes in seconds, so
    this is a direct comparison plus the closure properties (idempotence, seeds
    inside, closed under the edge relation among loaded modules)."""
    V = 1_000_000
    rowptr, col = synth.dep_graph(3, V, 10)
    assert 8_000_000 < len(col) < 12_000_000
    flags = synth.dep_flags(3, rowptr, col)
    seeds, _ = synth.seed_bitmap(3, flags, 16)
    out = check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, seeds)
    loaded = (flags & 1) != 0
    assert ((out & 4) != 0).sum() > 0 and not ((out != 0) & ~loaded).any()
    # idempotence: the closure of a closure is itself
    clo, _, _ = gpu_ctx.dep_closure(flags, seeds, 0)
    clo2, _, _ = gpu_ctx.dep_closure(flags, clo, 0)
    assert np.array_equal(clo, clo2)
    bits = np.unpackbits(clo.view(np.uint8), bitorder="little")[:V].astype(bool)
    sbits = np.unpackbits(seeds.view(np.uint8), bitorder="little")[:V].astype(bool)
    assert not (sbits & loaded & ~bits).any()
    # closed: a loaded holder u of a member v is a member
    src = np.repeat(np.arange(V, dtype=np.uint32), np.diff(rowptr))
    viol = bits[col] & loaded[src] & ~bits[src]
    assert not viol.any()


def test_k2_graph_too_large_for_lds_bitmap(gpu_ctx, oracle_mod):
    """1.5M modules: the LOADED bitmap (187 KB) no longer fits one workgroup's LDS, so the
    flags come from the row-block kernel instead of the one-block-per-CU kernel"""
    V = 1_500_000
    rowptr, col = synth.dep_graph(8, V, 4)
    flags = synth.dep_flags(8, rowptr, col)
    seeds, _ = synth.seed_bitmap(8, flags, 4)
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, seeds)


def test_k2_wide_ids_and_long_rows(gpu_ctx, oracle_mod):
    """The one-block-per-CU flag kernel in its other shapes: more than 2^20 modules (holder
    ids kept as 32-bit words instead of 20-bit fields, bitmap still in LDS), and a module
    held by 150000 others (a slice of one row whose ids do not fit the block's registers
    at once: several rounds) beside rows of every small length."""
    V = 1_100_000
    rowptr, col = synth.dep_graph(9, V, 4)
    flags = synth.dep_flags(9, rowptr, col)
    seeds, _ = synth.seed_bitmap(9, flags, 4)
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, seeds)
    # exactly 2^20 modules: the largest graph with 20-bit ids, and its last module is a holder
    V = 1 << 20
    rowptr, col = synth.dep_graph(10, V, 3)
    flags = synth.dep_flags(10, rowptr, col)
    flags[V - 1] |= 1
    seeds, _ = synth.seed_bitmap(10, flags, 4)
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, seeds)
    rng = np.random.default_rng(5)
    V = 200_000
    holders = rng.choice(np.arange(1000, V), size=150_000, replace=False)
    edges = {(int(u), 7) for u in holders}
    for u in rng.integers(1000, V, size=300_000):
        edges.add((int(u), int(rng.integers(0, 1000))))
    e = np.array(sorted(edges), dtype=np.uint32)
    rowptr = np.zeros(V + 1, dtype=np.uint32)
    np.add.at(rowptr, e[:, 0].astype(np.int64) + 1, 1)
    rowptr = np.cumsum(rowptr, dtype=np.uint64).astype(np.uint32)
    col = np.ascontiguousarray(e[:, 1])
    flags = np.where(rng.random(V) < 0.5, 5, 4).astype(np.uint8)
    flags[7] = 7
    seeds = np.zeros((V + 31) // 32, dtype=np.uint32)
    seeds[0] = 1 << 7
    check_k2(gpu_ctx, oracle_mod, rowptr, col, flags, seeds)


def test_k1_normalize_dash_on_device(gpu_ctx, oracle_mod):
    """kmod's '-' -> '_' rule applied to the queried strings by the kernel: dashes in
    and outside brackets, unmatched brackets (unresolvable), bracket pairs that cross
    the 64-byte steps of the scan, strings on both sides of the 64-byte matcher switch;
    through the host-buffer and the device-buffer entry points"""
    rng = np.random.default_rng(123)
    pats = [b"a-b*", b"a_b*", b"*-*", b"*_*", b"x[a-c]y*", b"*[[]*", b"ab", b"*", b"a?b*z", b"*a-b",
            b"foo-bar:v*", b"*q_[_-]*", b"[x-]*"]
    alpha = list(b"ab-_xq") + [ord("["), ord("]")]
    strs = [b"", b"a-b", b"a_b", b"-", b"a-bz", b"[a-b]", b"a[-]b-c", b"a]b", b"a[b", b"x[a-c]y", b"foo-bar:v12",
            b"a-b" + b"x" * 62 + b"[--]" + b"-" * 5, b"x" * 63 + b"[" + b"-" * 10 + b"]-", b"-" * 200,
            b"[" + b"a" * 130 + b"]-z", b"q" * 64 + b"]"]
    for _ in range(300):
        strs.append(bytes(rng.choice(alpha) for _ in range(int(rng.integers(0, 150)))))
    A, a_off = pack_strings(pats)
    S, s_off = pack_strings(strs)
    a_mod = np.arange(len(pats), dtype=np.uint32)
    first, count = check_k1(gpu_ctx, oracle_mod, A, a_off, a_mod, S, s_off, flags=1)
    assert (first == NO_MATCH).any() and (count > 3).any()
    # same through device buffers
    N = len(strs)
    dS, dO = gpu_ctx.to_device(S), gpu_ctx.to_device(s_off)
    dF, dC = gpu_ctx.alloc(4 * N), gpu_ctx.alloc(4 * N)
    gpu_ctx.alias_match_dev(dS.ptr, dO.ptr, N, dF.ptr, dC.ptr)
    gpu_ctx.sync()
    assert np.array_equal(dF.download(np.uint32, N), first) and np.array_equal(dC.download(np.uint32, N), count)
    for b in (dS, dO, dF, dC):
        b.free()


# ---- text tables in front of the kernels -------------------------------------------

def test_text_tables_resolve_like_kmod(gpu_ctx):
    """modules.alias text -> table -> match: the module kmod resolves for each query of
    the committed modprobe fixture, and all of its matches in file order"""
    from montecarlo_amd import ModTab, TABLE_NORMALIZE_DASH

    fx = load("kmod_resolve.json")
    text = "# generated\n" + "".join(f"alias {p} {m}\n" for p, m in fx["aliases"])
    with ModTab() as t:
        t.add_alias_text(text)
        gpu_ctx.alias_table_load_modtab(t, TABLE_NORMALIZE_DASH)
        _, _, a_mod = t.alias_arrays()
        S, s_off = pack_strings([q["query"].encode() for q in fx["queries"]])
        first, count = gpu_ctx.alias_match(S, s_off)
        for i, q in enumerate(fx["queries"]):
            assert count[i] == len(q["modules"]), q
            if q["modules"]:
                assert t.module_name(a_mod[first[i]]) == q["modules"][0].replace("-", "_"), q
            else:
                assert first[i] == NO_MATCH


def test_text_tables_dashboard_flow(gpu_ctx, oracle_mod):
    """modules.dep + /proc/modules text -> direct edges -> dep flags: the modules the
    dashboard would list for that host (loaded, safe, nobody holds them)"""
    from montecarlo_amd import ModTab, safe_module_names

    dep = ("kernel/a/e1000e.ko.zst: kernel/b/ptp.ko.zst kernel/c/pps_core.ko.zst\n"
           "kernel/b/ptp.ko.zst: kernel/c/pps_core.ko.zst\n"
           "kernel/c/pps_core.ko.zst:\n"
           "kernel/d/r8152.ko: kernel/d/mii.ko\n"
           "kernel/d/mii.ko:\n"
           "kernel/e/snd-hda-intel.ko: kernel/e/snd-hda-core.ko\n"
           "kernel/e/snd-hda-core.ko:\n"
           "kernel/f/igb.ko:\n")
    proc = ("e1000e 1 0 - Live 0x0\nptp 1 1 e1000e, Live 0x0\npps_core 1 1 ptp, Live 0x0\n"
            "snd_hda_intel 1 0 - Live 0x0\nsnd_hda_core 1 1 snd_hda_intel, Live 0x0\nigb 1 0 - Live 0x0\n")
    with ModTab() as t:
        t.add_dep_text(dep)
        names = t.module_names()
        flags, unknown = t.mark_loaded(proc)
        assert unknown == 0
        flags = flags | (safe_module_names(names) << 2)
        gpu_ctx.dep_graph_load_modtab(t)
        out, ref = gpu_ctx.dep_flags(flags)
        rowptr, col = t.dep_arrays(True)
        o_out, o_ref = oracle_mod.dep_flags(rowptr, col, flags)
        assert np.array_equal(out, o_out) and np.array_equal(ref, o_ref)
        roots = sorted(names[i] for i in np.nonzero(out & 4)[0])
        assert roots == ["e1000e", "igb"]          # snd_* fails the name filter, the others are held
        held = sorted(names[i] for i in np.nonzero(out & 1)[0])
        assert held == ["pps_core", "ptp", "snd_hda_core"]


def test_k2_golden(gpu_ctx):
    """the committed K2 vectors (hand-made chain and diamond, frozen random graph) through the C ABI"""
    for c in load("k2_golden.json")["cases"]:
        V = c["V"]
        seeds = np.zeros((V + 31) // 32, dtype=np.uint32)
        for v in c["seeds"]:
            seeds[v >> 5] |= np.uint32(1 << (v & 31))
        flags = np.array(c["in_flags"], np.uint8)
        gpu_ctx.dep_graph_load(np.array(c["rowptr"], np.uint32), np.array(c["col"], np.uint32))
        out, ref = gpu_ctx.dep_flags(flags)
        assert out.tolist() == c["out_flags"] and ref.tolist() == c["refcnt"], c["name"]
        for direction, key in ((0, "closure_holders"), (1, "closure_deps")):
            clo, edges, levels = gpu_ctx.dep_closure(flags, seeds, direction)
            members = [v for v in range(V) if clo[v >> 5] >> (v & 31) & 1]
            assert members == c[key]["members"] and edges == c[key]["edges"] and levels == c[key]["levels"]
