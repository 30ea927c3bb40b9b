"""CPU oracle and CPU twin - TEST INFRASTRUCTURE ONLY.

Loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
the checker.  Nothing in montecarlo_amd/ imports this package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_vp = ctypes.c_void_p
_ORC = None
_TWIN = None
NO_MATCH = 0xFFFFFFFF


def build():
    """Compile the oracle libraries (plain gcc/g++, seconds)."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def _load(name):
    path = os.path.join(_HERE, "_build", name)
    if not os.path.exists(path):
        build()
    return ctypes.CDLL(path)


def oracle():
    global _ORC
    if _ORC is None:
        lib = _load("libmcx_oracle.so")
        u32, u64 = ctypes.c_uint32, ctypes.c_uint64
        lib.orc_fnmatch.restype = ctypes.c_int
        lib.orc_fnmatch.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        lib.orc_alias_match.restype = u64
        lib.orc_alias_match.argtypes = [_vp, _vp, u32, _vp, _vp, u32, u32, ctypes.c_int, _vp, _vp]
        lib.orc_dep_flags.restype = None
        lib.orc_dep_flags.argtypes = [_vp, _vp, u32, _vp, _vp, _vp]
        lib.orc_dep_flags_pull.restype = None
        lib.orc_dep_flags_pull.argtypes = [_vp, _vp, _vp, _vp, _vp, u32, u32]
        lib.orc_dep_closure.restype = u64
        lib.orc_dep_closure.argtypes = [_vp, _vp, u32, u32, _vp, _vp, _vp, ctypes.c_int, ctypes.POINTER(u32)]
        lib.orc_safe_name.restype = ctypes.c_int
        lib.orc_safe_name.argtypes = [ctypes.c_char_p]
        lib.orc_has_hw_alias.restype = ctypes.c_int
        lib.orc_has_hw_alias.argtypes = [ctypes.c_char_p]
        lib.orc_fleet_roots.restype = u64
        lib.orc_fleet_roots.argtypes = [_vp, _vp, u32, _vp, _vp, u32, _vp, _vp, _vp, _vp, _vp, _vp]
        lib.orc_fleet_roots_bound.restype = u64
        lib.orc_fleet_roots_bound.argtypes = [_vp, _vp, u32, _vp, _vp, _vp, u32, _vp, _vp, _vp, _vp, _vp, _vp]
        lib.orc_alias_match_list.restype = u64
        lib.orc_alias_match_list.argtypes = [_vp, _vp, u32, _vp, _vp, u32, ctypes.c_int, u32, _vp, _vp]
        lib.orc_fleet_in_use.restype = None
        lib.orc_fleet_in_use.argtypes = [_vp, _vp, u32, _vp, _vp, _vp, _vp, u32, _vp, _vp, u32, _vp]
        lib.orc_device_filter.restype = None
        lib.orc_device_filter.argtypes = [_vp, u32, _vp, _vp, u32, _vp, u32, u32, _vp, _vp]
        _ORC = lib
    return _ORC


def twin():
    global _TWIN
    if _TWIN is None:
        lib = _load("libmcx_twin.so")
        u32 = ctypes.c_uint32
        lib.twin_compile_check.restype = ctypes.c_int
        lib.twin_compile_check.argtypes = [ctypes.c_char_p, u32, ctypes.c_int]
        lib.twin_fnmatch.restype = ctypes.c_int
        lib.twin_fnmatch.argtypes = [ctypes.c_char_p, u32, ctypes.c_char_p, u32]
        lib.twin_alias_match.restype = ctypes.c_int
        lib.twin_alias_match.argtypes = [_vp, _vp, u32, _vp, _vp, u32, u32, ctypes.c_int, _vp, _vp]
        lib.twin_table_create.restype = _vp
        lib.twin_table_create.argtypes = [_vp, _vp, u32, ctypes.c_int, u32, u32]
        lib.twin_table_destroy.restype = None
        lib.twin_table_destroy.argtypes = [_vp]
        lib.twin_table_match.restype = ctypes.c_int
        lib.twin_table_match.argtypes = [_vp, _vp, _vp, u32, u32, _vp, _vp]
        lib.twin_table_digest.restype = None
        lib.twin_table_digest.argtypes = [_vp, _vp]
        _TWIN = lib
    return _TWIN


def _p(a):
    return a.ctypes.data_as(_vp)


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a if a.size else np.zeros(1, dtype=dt)


def alias_match(arena, a_off, S, s_off, normalize_dash=False, i0=0, i1=None, use_twin=False):
    """(first_idx, count) for strings i0..i1 by glibc fnmatch (or the CPU twin)."""
    a_off = np.ascontiguousarray(a_off, dtype=np.uint32)
    s_off = np.ascontiguousarray(s_off, dtype=np.uint32)
    P, N = len(a_off) - 1, len(s_off) - 1
    i1 = N if i1 is None else i1
    first = np.full(N, NO_MATCH, dtype=np.uint32)
    count = np.zeros(N, dtype=np.uint32)
    A, SS = _c(arena, np.uint8), _c(S, np.uint8)
    if use_twin:
        rc = twin().twin_alias_match(_p(A), _p(a_off), P, _p(SS), _p(s_off), i0, i1, int(normalize_dash),
                                     _p(first), _p(count))
        if rc != 0:
            raise RuntimeError(f"twin_alias_match rc={rc}")
    else:
        oracle().orc_alias_match(_p(A), _p(a_off), P, _p(SS), _p(s_off), i0, i1, int(normalize_dash),
                                 _p(first), _p(count))
    return first[i0:i1], count[i0:i1]


def alias_match_sample(arena, a_off, S, s_off, picks, normalize_dash=False, threads=None):
    """(first_idx, count) of the strings `picks` by glibc fnmatch, the strings dealt over
    `threads` threads (the C loop runs without the interpreter lock)"""
    from concurrent.futures import ThreadPoolExecutor

    picks = [int(i) for i in picks]
    a_off = np.ascontiguousarray(a_off, dtype=np.uint32)
    s_off = np.ascontiguousarray(s_off, dtype=np.uint32)
    P = len(a_off) - 1
    raw = np.ascontiguousarray(S, dtype=np.uint8)
    parts = [raw[s_off[i]:s_off[i + 1]] for i in picks]
    SS = _c(np.concatenate(parts) if parts else np.zeros(0, np.uint8), np.uint8)
    ss_off = np.zeros(len(picks) + 1, dtype=np.uint32)
    ss_off[1:] = np.cumsum([len(x) for x in parts], dtype=np.uint64)
    n = len(picks)
    first = np.full(max(n, 1), NO_MATCH, dtype=np.uint32)
    count = np.zeros(max(n, 1), dtype=np.uint32)
    A = _c(arena, np.uint8)
    threads = threads or max(1, min(16, len(os.sched_getaffinity(0))))
    cut = [n * t // threads for t in range(threads + 1)]
    lib = oracle()

    def work(t):
        if cut[t + 1] > cut[t]:
            lib.orc_alias_match(_p(A), _p(a_off), P, _p(SS), _p(ss_off), cut[t], cut[t + 1], int(normalize_dash),
                                _p(first), _p(count))

    with ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(work, range(threads)))
    return first[:n], count[:n]


def hw_alias_flags(arena, a_off, a_mod, n_modules):
    """hw[m] = 1 iff module m owns an alias line that starts (after blanks) with one of the
    six bus tags of reference desktop/ui.py:1540, decided pattern by pattern on the host"""
    a_off = np.ascontiguousarray(a_off, dtype=np.uint32)
    raw = np.ascontiguousarray(arena, dtype=np.uint8).tobytes()
    hw = np.zeros(n_modules, dtype=np.uint8)
    lib = oracle()
    for j in range(len(a_off) - 1):
        m = int(a_mod[j])
        if m < n_modules and not hw[m] and lib.orc_has_hw_alias(raw[a_off[j]:a_off[j + 1]]):
            hw[m] = 1
    return hw


def alias_match_list(arena, a_off, S, s_off, K=0, normalize_dash=False):
    """(list_off, list_idx): every matching table position per string in table order, first K if K > 0"""
    a_off = np.ascontiguousarray(a_off, dtype=np.uint32)
    s_off = np.ascontiguousarray(s_off, dtype=np.uint32)
    P, N = len(a_off) - 1, len(s_off) - 1
    A, SS = _c(arena, np.uint8), _c(S, np.uint8)
    off = np.zeros(N + 1, dtype=np.uint32)
    args = (_p(A), _p(a_off), P, _p(SS), _p(s_off), N, int(normalize_dash), int(K), _p(off))
    total = oracle().orc_alias_match_list(*args, None)
    idx = np.zeros(max(total, 1), dtype=np.uint32)
    oracle().orc_alias_match_list(*args, _p(idx))
    return off, idx[:total]


def dep_flags(rowptr, col, in_flags):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.uint32)
    V = len(rowptr) - 1
    out = np.zeros(max(V, 1), dtype=np.uint8)
    ref = np.zeros(max(V, 1), dtype=np.uint32)
    oracle().orc_dep_flags(_p(rowptr), _p(_c(col, np.uint32)), V, _p(_c(in_flags, np.uint8)), _p(out), _p(ref))
    return out[:V], ref[:V]


def holders_graph(rowptr, col):
    """(rrow, rcol): module v is held by rcol[rrow[v] .. rrow[v+1]), holders ascending"""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.uint32)
    col = np.ascontiguousarray(col, dtype=np.uint32)
    V = len(rowptr) - 1
    src = np.repeat(np.arange(V, dtype=np.uint32), np.diff(rowptr.astype(np.int64)))
    rcol = np.ascontiguousarray(src[np.argsort(col, kind="stable")])
    rrow = np.zeros(V + 1, dtype=np.uint32)
    rrow[1:] = np.cumsum(np.bincount(col, minlength=V), dtype=np.uint64).astype(np.uint32)
    return rrow, rcol


def holders_cut(rrow, threads):
    """module ranges with equal shares of edges + modules, so that neither a hub nor a
    run of leaves makes a long range; a property of the graph, computed once"""
    V, E = len(rrow) - 1, int(rrow[-1])
    work = np.asarray(rrow, dtype=np.int64) + np.arange(V + 1, dtype=np.int64)
    cut = np.searchsorted(work, [(E + V) * t // threads for t in range(threads + 1)], side="left")
    cut[0], cut[-1] = 0, V
    return [int(c) for c in cut]


def dep_flags_threads(rrow, rcol, in_flags, threads, pool=None, cut=None):
    """dep_flags by `threads` threads over the graph of holders (holders_graph), every
    thread owning one range of holders_cut.  pool: a ThreadPoolExecutor to reuse (one is
    made and dropped otherwise)."""
    from concurrent.futures import ThreadPoolExecutor

    rrow = np.ascontiguousarray(rrow, dtype=np.uint32)
    rcol, in_flags = _c(rcol, np.uint32), _c(in_flags, np.uint8)
    V = len(rrow) - 1
    out = np.empty(max(V, 1), dtype=np.uint8)
    ref = np.empty(max(V, 1), dtype=np.uint32)
    cut = cut or holders_cut(rrow, threads)
    lib = oracle()
    own = pool is None
    pool = pool or ThreadPoolExecutor(max_workers=threads)
    try:
        list(pool.map(lambda t: lib.orc_dep_flags_pull(_p(rrow), _p(rcol), _p(in_flags), _p(out), _p(ref),
                                                       cut[t], cut[t + 1]), range(threads)))
    finally:
        if own:
            pool.shutdown()
    return out[:V], ref[:V]


def dep_closure(rowptr, col, in_flags, seed_bitmap, direction=0):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.uint32)
    V, E = len(rowptr) - 1, len(col)
    W = (V + 31) // 32
    out = np.zeros(max(W, 1), dtype=np.uint32)
    lv = ctypes.c_uint32()
    scanned = oracle().orc_dep_closure(_p(rowptr), _p(_c(col, np.uint32)), V, E, _p(_c(in_flags, np.uint8)),
                                       _p(_c(seed_bitmap, np.uint32)), _p(out), direction, ctypes.byref(lv))
    return out[:W], scanned, lv.value


def safe_name(name):
    return bool(oracle().orc_safe_name(name.encode() if isinstance(name, str) else name))


def fleet_roots(rowptr, col, driver_mod, dev_off, extra_off, extra_ids, safe_name_bits, hw_alias, bound_mod=None):
    """(root_off, root_ids) of the per-host dashboard loop; driver_mod[i] is the
    module device i resolves to (NO_MATCH = none).  With bound_mod the used drivers are
    the modules named like the devices' bound drivers (reference desktop/ui.py:1358-1359)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.uint32)
    dev_off = np.ascontiguousarray(dev_off, dtype=np.uint32)
    extra_off = np.ascontiguousarray(extra_off, dtype=np.uint32)
    V, H = len(rowptr) - 1, len(dev_off) - 1
    root_off = np.zeros(H + 1, dtype=np.uint32)
    args = [_p(rowptr), _p(_c(col, np.uint32)), V, _p(_c(driver_mod, np.uint32))]
    fn = oracle().orc_fleet_roots
    if bound_mod is not None:
        args.append(_p(_c(bound_mod, np.uint32)))
        fn = oracle().orc_fleet_roots_bound
    args += [_p(dev_off), H, _p(extra_off), _p(_c(extra_ids, np.uint32)), _p(_c(safe_name_bits, np.uint8)),
             _p(_c(hw_alias, np.uint8)), _p(root_off)]
    total = fn(*args, None)
    ids = np.zeros(max(total, 1), dtype=np.uint32)
    fn(*args, _p(ids))
    return root_off, ids[:total]


def fleet_in_use(root_off, root_ids, rec, rec_off, drv_arena, drv_off, mod_arena, mod_off):
    """the "(In Use)" byte of every listed module: mc_driver_is_in_use restated over the hosts'
    device records (all of them, listed or not), driver names and module names packed"""
    root_off = np.ascontiguousarray(root_off, dtype=np.uint32)
    root_ids = _c(root_ids, np.uint32)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint32)
    drv_off = np.ascontiguousarray(drv_off, dtype=np.uint32)
    mod_off = np.ascontiguousarray(mod_off, dtype=np.uint32)
    H = len(root_off) - 1
    n = int(root_off[H])
    out = np.zeros(max(n, 1), dtype=np.uint8)
    recb = np.ascontiguousarray(rec).view(np.uint8) if len(rec) else np.zeros(16, np.uint8)
    oracle().orc_fleet_in_use(_p(root_off), _p(root_ids), H, _p(recb), _p(rec_off), _p(_c(drv_arena, np.uint8)),
                              _p(drv_off), len(drv_off) - 1, _p(_c(mod_arena, np.uint8)), _p(mod_off),
                              len(mod_off) - 1, _p(out))
    return out[:n]


def device_filter(rec, drv_arena, drv_off, dev_off, max_per_host=0):
    """(verdict & 3, kept_off) of the per-host device list over packed records
    (rec: structured or raw array of 16-byte records)"""
    rec = np.ascontiguousarray(rec)
    assert rec.itemsize == 16
    drv_off = np.ascontiguousarray(drv_off, dtype=np.uint32)
    dev_off = np.ascontiguousarray(dev_off, dtype=np.uint32)
    N, H = len(rec), len(dev_off) - 1
    verdict = np.zeros(max(N, 1), dtype=np.uint8)
    kept_off = np.zeros(H + 1, dtype=np.uint32)
    oracle().orc_device_filter(_p(rec), N, _p(_c(drv_arena, np.uint8)), _p(drv_off), len(drv_off) - 1, _p(dev_off),
                               H, int(max_per_host), _p(verdict), _p(kept_off))
    return verdict[:N], kept_off


class TwinTable:
    """the CPU twin's compiled table, kept between calls (indexed CPU baseline)"""

    def __init__(self, arena, a_off, normalize_dash=False, threads=0, sub_min=0):
        """threads / sub_min: the table compiler's options (0 = its defaults)"""
        a_off = np.ascontiguousarray(a_off, dtype=np.uint32)
        self.h = twin().twin_table_create(_p(_c(arena, np.uint8)), _p(a_off), len(a_off) - 1, int(normalize_dash),
                                          int(threads), int(sub_min))
        if not self.h:
            raise RuntimeError("twin_table_create failed")

    def match(self, S, s_off, first, count, i0, i1):
        """fills first[i0:i1], count[i0:i1] (uint32 arrays of N entries)"""
        twin().twin_table_match(self.h, _p(S), _p(s_off), i0, i1, _p(first), _p(count))

    def close(self):
        if self.h:
            twin().twin_table_destroy(self.h)
            self.h = None
