"""ctypes binding of the mcx_* C ABI (include/mcx.h).

Mirrors the way the reference loads its library: ``ctypes.CDLL(LIB_PATH)`` with
explicit argtypes/restype and caller-allocated buffers (reference
desktop/ui.py:53-103).  The environment variable ``MONTECARLO_LIB`` overrides
the library path, like the reference's ``MONTECARLO_DEV`` switch
(desktop/ui.py:44-51).
"""
import ctypes
import os

import numpy as np

NO_MATCH = 0xFFFFFFFF
F_LOADED, F_BOUND, F_SAFE = 1, 2, 4
O_HAS_HOLDERS, O_IN_USE, O_ROOT = 1, 2, 4
CLOSURE_HOLDERS, CLOSURE_DEPS = 0, 1
TABLE_NORMALIZE_DASH = 1
TUNE_COMPILE_THREADS, TUNE_SUB_MIN, TUNE_TILES_MIN, TUNE_LIST_TWO_PASS = 1, 2, 3, 4
TUNE_AUTO = 0xFFFFFFFFFFFFFFFF

# packed device record of the fleet inventory (mcx_dev_rec_t, 16 bytes)
DEV_REC = np.dtype([("subsystem", "u1"), ("devtype", "u1"), ("attrs", "u1"), ("iface_class", "u1"),
                    ("pci_class", "<u4"), ("parent_class", "u1"), ("own_class", "u1"), ("reserved", "<u2"),
                    ("driver", "<u4")])
SUB_USB, SUB_PCI, SUB_HID, SUB_SCSI, SUB_PCMCIA, SUB_OTHER = 0, 1, 2, 3, 4, 255
DT_NONE, DT_USB_DEVICE, DT_USB_INTERFACE, DT_SCSI_HOST, DT_SCSI_TARGET, DT_SCSI_GENERIC, DT_OTHER = range(7)
DA_PCI_CLASS, DA_IFACE_CLASS, DA_PARENT, DA_PARENT_CLASS, DA_OWN_CLASS, DA_SCSI_MODEL, DA_SCSI_VENDOR = (
    1, 2, 4, 8, 16, 32, 64)
DRV_NONE, DRV_NOSLASH = 0xFFFFFFFF, 0xFFFFFFFE
DV_LISTED, DV_EXCLUDED = 1, 2

_c_u32p = ctypes.POINTER(ctypes.c_uint32)
_c_u8p = ctypes.POINTER(ctypes.c_uint8)
_vp = ctypes.c_void_p

_LIB = None


class McxError(RuntimeError):
    def __init__(self, rc, what, detail=""):
        self.rc = rc
        super().__init__(f"{what}: rc={rc} ({detail})" if detail else f"{what}: rc={rc}")


def library_path():
    env = os.environ.get("MONTECARLO_LIB")
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmontecarlo.so")


def load_library():
    """Load libmontecarlo.so; raises OSError when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise OSError(
            f"{path} not found: build it with `make -C montecarlo_amd/csrc` "
            "(there is no CPU fallback for the batch path)"
        )
    lib = ctypes.CDLL(path)
    sig = {
        "mcx_ctx_create": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.POINTER(_vp)]),
        "mcx_ctx_destroy": (None, [_vp]),
        "mcx_sync": (ctypes.c_int, [_vp]),
        "mcx_ctx_stream": (_vp, [_vp]),
        "mcx_ctx_tune": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_uint64]),
        "mcx_strerror": (ctypes.c_char_p, [ctypes.c_int]),
        "mcx_last_error": (ctypes.c_char_p, [_vp]),
        "mcx_dev_alloc": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
        "mcx_dev_free": (ctypes.c_int, [_vp, _vp]),
        "mcx_h2d": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
        "mcx_d2h": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
        "mcx_timer_begin": (ctypes.c_int, [_vp]),
        "mcx_timer_end": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
        "mcx_alias_table_load": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_uint32, ctypes.c_uint32]),
        "mcx_alias_match": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint32, _vp, _vp]),
        "mcx_alias_match_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint32, _vp, _vp]),
        "mcx_alias_match_list": (
            ctypes.c_int,
            [_vp, _vp, _vp, ctypes.c_uint32, ctypes.c_uint32, _vp, _vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_alias_match_list_dev": (
            ctypes.c_int,
            [_vp, _vp, _vp, ctypes.c_uint32, ctypes.c_uint32, _vp, _vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_alias_driver_mod_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, _vp]),
        "mcx_alias_hw_flags": (ctypes.c_int, [_vp, ctypes.c_uint32, _vp]),
        "mcx_alias_table_info": (
            ctypes.c_int,
            [_vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_dep_graph_load": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint32, ctypes.c_uint32]),
        "mcx_dep_flags": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
        "mcx_dep_flags_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
        "mcx_dep_flags_noalloc_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
        "mcx_dep_closure": (
            ctypes.c_int,
            [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)],
        ),
        "mcx_dep_closure_dev": (
            ctypes.c_int,
            [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)],
        ),
        "mcx_modtab_create": (_vp, []),
        "mcx_modtab_destroy": (None, [_vp]),
        "mcx_modtab_last_error": (ctypes.c_char_p, [_vp]),
        "mcx_modtab_add_alias_text": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t]),
        "mcx_modtab_add_dep_text": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t]),
        "mcx_modtab_load_dir": (ctypes.c_int, [_vp, ctypes.c_char_p]),
        "mcx_modtab_counts": (
            ctypes.c_int,
            [_vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_modtab_alias_arrays": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
        "mcx_modtab_dep_arrays": (
            ctypes.c_int,
            [_vp, ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_uint32)],
        ),
        "mcx_modtab_module_name": (ctypes.c_char_p, [_vp, ctypes.c_uint32]),
        "mcx_modtab_module_id": (ctypes.c_uint32, [_vp, ctypes.c_char_p]),
        "mcx_modtab_module_names": (ctypes.c_int, [_vp, _vp, ctypes.c_uint64, _vp, ctypes.POINTER(ctypes.c_uint64)]),
        "mcx_modtab_mark_loaded": (
            ctypes.c_int,
            [_vp, ctypes.c_char_p, ctypes.c_size_t, _vp, ctypes.POINTER(ctypes.c_uint32)],
        ),
        "mcx_alias_table_load_modtab": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32]),
        "mcx_dep_graph_load_modtab": (ctypes.c_int, [_vp, _vp]),
        "mcx_safe_module_names": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, _vp]),
        "mcx_safe_table_sizes": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]),
        "mcx_tables_load_file": (ctypes.c_int, [ctypes.c_char_p]),
        "mcx_tables_load_text": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t]),
        "mcx_tables_reset": (None, []),
        "mcx_tables_dump": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
        "mcx_tables_info": (
            ctypes.c_int,
            [ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int)],
        ),
        "mcx_fleet_roots": (
            ctypes.c_int,
            [_vp, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64,
             ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_device_filter": (
            ctypes.c_int,
            [_vp, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, ctypes.c_uint32, _vp, _vp],
        ),
        "mcx_device_filter_dev": (
            ctypes.c_int,
            [_vp, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, ctypes.c_uint32, _vp, _vp,
             ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_fleet_compact_dev": (
            ctypes.c_int,
            [_vp, _vp, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, _vp, _vp, _vp,
             ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_fleet_inventory_roots_dev": (
            ctypes.c_int,
            [_vp, _vp, _vp, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, ctypes.c_uint32,
             _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_fleet_roots_dev": (
            ctypes.c_int,
            [_vp, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp,
             ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_fleet_roots_bound": (
            ctypes.c_int,
            [_vp, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64,
             ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_fleet_roots_bound_dev": (
            ctypes.c_int,
            [_vp, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp,
             ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_fleet_inventory_roots_bound_dev": (
            ctypes.c_int,
            [_vp, _vp, _vp, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint32,
             ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, ctypes.c_uint64,
             ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_driver_modules": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp]),
        "mcx_driver_marks": (
            ctypes.c_int,
            [_vp, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint64,
             ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_fleet_roots_labeled": (
            ctypes.c_int,
            [_vp, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
             ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_fleet_roots_labeled_dev": (
            ctypes.c_int,
            [_vp, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp,
             ctypes.c_uint32, _vp, _vp, _vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)],
        ),
        "mcx_fleet_inventory_roots_labeled_dev": (
            ctypes.c_int,
            [_vp, _vp, _vp, _vp, ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, ctypes.c_uint32,
             ctypes.c_uint32, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64,
             ctypes.POINTER(ctypes.c_uint64)],
        ),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None else None


def _as(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def pack_strings(strings):
    """list of bytes -> (arena uint8[nbytes], offsets uint32[n+1])"""
    off = np.zeros(len(strings) + 1, dtype=np.uint32)
    if strings:
        off[1:] = np.cumsum([len(s) for s in strings], dtype=np.uint64).astype(np.uint32)
    arena = np.frombuffer(b"".join(strings), dtype=np.uint8).copy() if strings else np.zeros(0, np.uint8)
    return arena, off


class DeviceBuffer:
    """hipMalloc'ed bytes owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = _vp()
        ctx._check(ctx.lib.mcx_dev_alloc(ctx.h, self.nbytes, ctypes.byref(p)), "mcx_dev_alloc")
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.mcx_h2d(self.ctx.h, self.ptr, _ptr(arr), arr.nbytes), "mcx_h2d")
        return self

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.mcx_d2h(self.ctx.h, _ptr(out), self.ptr, out.nbytes), "mcx_d2h")
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.mcx_dev_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """One GPU context (one HIP stream).  Not thread safe: one per thread."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.mcx_ctx_create(int(device), stream, ctypes.byref(h))
        if rc != 0:
            raise McxError(rc, "mcx_ctx_create", self.lib.mcx_strerror(rc).decode())
        self.h = h.value
        self.P = 0
        self.V = 0
        self.E = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.mcx_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            detail = self.lib.mcx_last_error(self.h).decode(errors="replace")
            raise McxError(rc, what, detail or self.lib.mcx_strerror(rc).decode())

    def tune(self, compile_threads=None, sub_min=None, tiles_min=None, list_two_pass=None):
        """Tuning knobs for tables loaded after the call (none changes a result):
        compile_threads (0 = one per core), sub_min (0 = default), tiles_min (batch size
        from which the tile kernel runs; "auto" = the library's rule), list_two_pass."""
        if compile_threads is not None:
            self._check(self.lib.mcx_ctx_tune(self.h, TUNE_COMPILE_THREADS, int(compile_threads)), "mcx_ctx_tune")
        if sub_min is not None:
            self._check(self.lib.mcx_ctx_tune(self.h, TUNE_SUB_MIN, int(sub_min)), "mcx_ctx_tune")
        if tiles_min is not None:
            v = TUNE_AUTO if tiles_min == "auto" else int(tiles_min)
            self._check(self.lib.mcx_ctx_tune(self.h, TUNE_TILES_MIN, v), "mcx_ctx_tune")
        if list_two_pass is not None:
            self._check(self.lib.mcx_ctx_tune(self.h, TUNE_LIST_TWO_PASS, int(bool(list_two_pass))), "mcx_ctx_tune")

    def sync(self):
        self._check(self.lib.mcx_sync(self.h), "mcx_sync")

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 1)).upload(arr)

    def timer_begin(self):
        self._check(self.lib.mcx_timer_begin(self.h), "mcx_timer_begin")

    def timer_end(self):
        ms = ctypes.c_float()
        self._check(self.lib.mcx_timer_end(self.h, ctypes.byref(ms)), "mcx_timer_end")
        return ms.value

    # ---- K1 ----------------------------------------------------------------
    def alias_table_load(self, arena, a_off, a_mod, flags=0):
        arena = _as(arena, np.uint8)
        a_off = _as(a_off, np.uint32)
        a_mod = _as(a_mod, np.uint32)
        P = len(a_off) - 1
        if len(a_mod) != P:
            raise ValueError("a_mod must have one entry per pattern")
        self._check(
            self.lib.mcx_alias_table_load(self.h, _ptr(arena), _ptr(a_off), _ptr(a_mod), P, flags),
            "mcx_alias_table_load",
        )
        self.P = P

    def alias_match(self, S, s_off):
        S = _as(S, np.uint8)
        s_off = _as(s_off, np.uint32)
        N = len(s_off) - 1
        first = np.empty(N, dtype=np.uint32)
        count = np.empty(N, dtype=np.uint32)
        self._check(
            self.lib.mcx_alias_match(self.h, _ptr(S), _ptr(s_off), N, _ptr(first), _ptr(count)),
            "mcx_alias_match",
        )
        return first, count

    def alias_match_dev(self, d_S, d_s_off, N, d_first, d_count):
        self._check(
            self.lib.mcx_alias_match_dev(self.h, d_S, d_s_off, N, d_first, d_count), "mcx_alias_match_dev"
        )

    def alias_driver_mod_dev(self, d_first, N, d_out):
        self._check(self.lib.mcx_alias_driver_mod_dev(self.h, d_first, N, d_out), "mcx_alias_driver_mod_dev")

    def alias_hw_flags(self, n_modules):
        out = np.zeros(n_modules, dtype=np.uint8)
        self._check(self.lib.mcx_alias_hw_flags(self.h, n_modules, _ptr(out)), "mcx_alias_hw_flags")
        return out

    def alias_table_info(self):
        P = ctypes.c_uint32()
        b = ctypes.c_uint64()
        self._check(self.lib.mcx_alias_table_info(self.h, ctypes.byref(P), ctypes.byref(b)), "mcx_alias_table_info")
        return P.value, b.value

    # ---- K2 ----------------------------------------------------------------
    def dep_graph_load(self, rowptr, col):
        rowptr = _as(rowptr, np.uint32)
        col = _as(col, np.uint32)
        V = len(rowptr) - 1
        E = len(col)
        self._check(self.lib.mcx_dep_graph_load(self.h, _ptr(rowptr), _ptr(col), V, E), "mcx_dep_graph_load")
        self.V, self.E = V, E

    def dep_flags(self, in_flags, want_refcnt=True):
        in_flags = _as(in_flags, np.uint8)
        if len(in_flags) != self.V:
            raise ValueError("in_flags must have V entries")
        out = np.empty(self.V, dtype=np.uint8)
        ref = np.empty(self.V, dtype=np.uint32) if want_refcnt else None
        self._check(self.lib.mcx_dep_flags(self.h, _ptr(in_flags), _ptr(out), _ptr(ref)), "mcx_dep_flags")
        return out, ref

    def dep_flags_dev(self, d_in, d_out, d_ref=None):
        self._check(self.lib.mcx_dep_flags_dev(self.h, d_in, d_out, d_ref), "mcx_dep_flags_dev")

    def dep_flags_noalloc_dev(self, d_in, d_out, d_ref=None):
        """mcx_dep_flags_dev without any allocation: the bitmap lives in a static device array"""
        self._check(self.lib.mcx_dep_flags_noalloc_dev(self.h, d_in, d_out, d_ref), "mcx_dep_flags_noalloc_dev")

    def dep_closure(self, in_flags, seed_bitmap, direction=CLOSURE_HOLDERS):
        in_flags = _as(in_flags, np.uint8)
        seed_bitmap = _as(seed_bitmap, np.uint32)
        W = (self.V + 31) // 32
        if len(in_flags) != self.V or len(seed_bitmap) != W:
            raise ValueError("in_flags needs V entries and seed_bitmap ceil(V/32) words")
        out = np.zeros(W, dtype=np.uint32)
        edges = ctypes.c_uint64()
        levels = ctypes.c_uint32()
        self._check(
            self.lib.mcx_dep_closure(
                self.h, _ptr(in_flags), _ptr(seed_bitmap), _ptr(out), direction, ctypes.byref(edges),
                ctypes.byref(levels)
            ),
            "mcx_dep_closure",
        )
        return out, edges.value, levels.value

    def dep_closure_dev(self, d_in, d_seed, d_out, direction=CLOSURE_HOLDERS, want_stats=True):
        edges = ctypes.c_uint64()
        levels = ctypes.c_uint32()
        self._check(
            self.lib.mcx_dep_closure_dev(
                self.h, d_in, d_seed, d_out, direction,
                ctypes.byref(edges) if want_stats else None,
                ctypes.byref(levels) if want_stats else None,
            ),
            "mcx_dep_closure_dev",
        )
        return edges.value, levels.value

    def alias_table_load_modtab(self, tab, flags=0):
        self._check(self.lib.mcx_alias_table_load_modtab(self.h, tab.h, flags), "mcx_alias_table_load_modtab")
        self.P = tab.counts()[0]

    def dep_graph_load_modtab(self, tab):
        self._check(self.lib.mcx_dep_graph_load_modtab(self.h, tab.h), "mcx_dep_graph_load_modtab")
        rowptr, col = tab.dep_arrays(True)
        self.V, self.E = len(rowptr) - 1, len(col)

    def fleet_roots(self, S, s_off, dev_off, extra_off, extra_ids, safe_name, cap=None):
        """(root_off, root_ids) of mcx_fleet_roots; sizes root_ids itself when cap is None"""
        S, s_off = _as(S, np.uint8), _as(s_off, np.uint32)
        dev_off, extra_off = _as(dev_off, np.uint32), _as(extra_off, np.uint32)
        extra_ids, safe_name = _as(extra_ids, np.uint32), _as(safe_name, np.uint8)
        N, H = len(s_off) - 1, len(dev_off) - 1
        if len(extra_off) != H + 1 or len(safe_name) != self.V:
            raise ValueError("extra_off needs H+1 entries and safe_name V entries")
        root_off = np.zeros(H + 1, dtype=np.uint32)
        total = ctypes.c_uint64()
        ids = np.zeros(int(cap) if cap else 1, dtype=np.uint32)
        args = (self.h, _ptr(S), _ptr(s_off), N, _ptr(dev_off), H, _ptr(extra_off), _ptr(extra_ids),
                _ptr(safe_name), _ptr(root_off))
        rc = self.lib.mcx_fleet_roots(*args, _ptr(ids), int(cap) if cap else 0, ctypes.byref(total))
        if rc == -6 and cap is None and total.value < 2**32:
            ids = np.zeros(total.value, dtype=np.uint32)
            rc = self.lib.mcx_fleet_roots(*args, _ptr(ids), total.value, ctypes.byref(total))
        self._check(rc, "mcx_fleet_roots")
        return root_off, ids[: total.value]

    def fleet_roots_bound(self, S, s_off, dev_off, extra_off, extra_ids, safe_name, bound_mod, cap=None):
        """mcx_fleet_roots_bound: used drivers = bound_mod[i] (module named like device i's bound driver)"""
        S, s_off = _as(S, np.uint8), _as(s_off, np.uint32)
        dev_off, extra_off = _as(dev_off, np.uint32), _as(extra_off, np.uint32)
        extra_ids, safe_name, bound_mod = _as(extra_ids, np.uint32), _as(safe_name, np.uint8), _as(bound_mod, np.uint32)
        N, H = len(s_off) - 1, len(dev_off) - 1
        if len(extra_off) != H + 1 or len(safe_name) != self.V or len(bound_mod) != N:
            raise ValueError("extra_off needs H+1 entries, safe_name V entries, bound_mod N entries")
        root_off = np.zeros(H + 1, dtype=np.uint32)
        total = ctypes.c_uint64()
        ids = np.zeros(int(cap) if cap else 1, dtype=np.uint32)
        args = (self.h, _ptr(S), _ptr(s_off), N, _ptr(dev_off), H, _ptr(extra_off), _ptr(extra_ids),
                _ptr(safe_name), _ptr(bound_mod), _ptr(root_off))
        rc = self.lib.mcx_fleet_roots_bound(*args, _ptr(ids), int(cap) if cap else 0, ctypes.byref(total))
        if rc == -6 and cap is None and total.value < 2**32:
            ids = np.zeros(total.value, dtype=np.uint32)
            rc = self.lib.mcx_fleet_roots_bound(*args, _ptr(ids), total.value, ctypes.byref(total))
        self._check(rc, "mcx_fleet_roots_bound")
        return root_off, ids[: total.value]

    def fleet_roots_labeled(self, S, s_off, dev_off, extra_off, extra_ids, safe_name, bound_mod, inuse_off,
                            inuse_ids):
        """mcx_fleet_roots_labeled: (root_off, root_ids, root_in_use); host h's marked modules are
        inuse_ids[inuse_off[h]:inuse_off[h+1]] (see driver_marks)"""
        S, s_off = _as(S, np.uint8), _as(s_off, np.uint32)
        dev_off, extra_off = _as(dev_off, np.uint32), _as(extra_off, np.uint32)
        extra_ids, safe_name, bound_mod = _as(extra_ids, np.uint32), _as(safe_name, np.uint8), _as(bound_mod, np.uint32)
        inuse_off, inuse_ids = _as(inuse_off, np.uint32), _as(inuse_ids, np.uint32)
        N, H = len(s_off) - 1, len(dev_off) - 1
        if len(extra_off) != H + 1 or len(inuse_off) != H + 1 or len(safe_name) != self.V or len(bound_mod) != N:
            raise ValueError("extra_off and inuse_off need H+1 entries, safe_name V entries, bound_mod N entries")
        root_off = np.zeros(H + 1, dtype=np.uint32)
        total = ctypes.c_uint64()
        args = (self.h, _ptr(S), _ptr(s_off), N, _ptr(dev_off), H, _ptr(extra_off), _ptr(extra_ids),
                _ptr(safe_name), _ptr(bound_mod), _ptr(inuse_off), _ptr(inuse_ids), _ptr(root_off))
        rc = self.lib.mcx_fleet_roots_labeled(*args, None, None, 0, ctypes.byref(total))
        if rc not in (0, -6):
            self._check(rc, "mcx_fleet_roots_labeled")
        n = total.value
        ids, use = np.zeros(max(n, 1), dtype=np.uint32), np.zeros(max(n, 1), dtype=np.uint8)
        if n:
            self._check(self.lib.mcx_fleet_roots_labeled(*args, _ptr(ids), _ptr(use), n, ctypes.byref(total)),
                        "mcx_fleet_roots_labeled")
        return root_off, ids[:n], use[:n]

    def fleet_roots_labeled_dev(self, d_S, d_s_off, N, d_dev_off, H, d_extra_off, d_extra_ids, X, d_safe, d_bound_mod,
                                d_inuse_off, d_inuse_ids, Y, d_root_off, d_root_ids, d_root_in_use, cap,
                                size_only=False):
        total = ctypes.c_uint64()
        rc = self.lib.mcx_fleet_roots_labeled_dev(self.h, d_S, d_s_off, N, d_dev_off, H, d_extra_off, d_extra_ids, X,
                                                  d_safe, d_bound_mod, d_inuse_off, d_inuse_ids, Y, d_root_off,
                                                  d_root_ids, d_root_in_use, cap, ctypes.byref(total))
        if not (size_only and rc == -6 and cap < total.value < 2**32):
            self._check(rc, "mcx_fleet_roots_labeled_dev")
        return total.value

    def fleet_roots_dev(self, d_S, d_s_off, N, d_dev_off, H, d_extra_off, d_extra_ids, X, d_safe, d_root_off,
                        d_root_ids, cap, size_only=False):
        """returns the total; with size_only a too-small root_ids is not an error"""
        total = ctypes.c_uint64()
        rc = self.lib.mcx_fleet_roots_dev(self.h, d_S, d_s_off, N, d_dev_off, H, d_extra_off, d_extra_ids, X,
                                          d_safe, d_root_off, d_root_ids, cap, ctypes.byref(total))
        if not (size_only and rc == -6 and cap < total.value < 2**32):
            self._check(rc, "mcx_fleet_roots_dev")
        return total.value


    def alias_match_list(self, S, s_off, K=0, cap=None):
        """(list_off, list_idx) of mcx_alias_match_list; sizes list_idx itself when cap is None"""
        S, s_off = _as(S, np.uint8), _as(s_off, np.uint32)
        N = len(s_off) - 1
        off = np.zeros(N + 1, dtype=np.uint32)
        total = ctypes.c_uint64()
        idx = np.zeros(int(cap) if cap else 1, dtype=np.uint32)
        args = (self.h, _ptr(S) if len(S) else None, _ptr(s_off), N, int(K), _ptr(off))
        rc = self.lib.mcx_alias_match_list(*args, _ptr(idx), int(cap) if cap else 0, ctypes.byref(total))
        if rc == -6 and cap is None and total.value < 2**32:
            idx = np.zeros(total.value, dtype=np.uint32)
            rc = self.lib.mcx_alias_match_list(*args, _ptr(idx), total.value, ctypes.byref(total))
        self._check(rc, "mcx_alias_match_list")
        return off, idx[: total.value]

    def alias_match_list_dev(self, d_S, d_s_off, N, K, d_list_off, d_list_idx, cap, size_only=False):
        """returns the total; with size_only a too-small list_idx is not an error"""
        total = ctypes.c_uint64()
        rc = self.lib.mcx_alias_match_list_dev(self.h, d_S, d_s_off, N, int(K), d_list_off, d_list_idx, int(cap),
                                               ctypes.byref(total))
        if not (size_only and rc == -6 and cap < total.value < 2**32):
            self._check(rc, "mcx_alias_match_list_dev")
        return total.value

    # ---- fleet inventory predicates ------------------------------------------
    def device_filter(self, rec, drv_arena, drv_off, dev_off, max_per_host=0):
        """(verdict uint8[N], kept_off uint32[H+1]) of mcx_device_filter"""
        rec = np.ascontiguousarray(rec, dtype=DEV_REC)
        drv_arena, drv_off, dev_off = _as(drv_arena, np.uint8), _as(drv_off, np.uint32), _as(dev_off, np.uint32)
        N, H, D = len(rec), len(dev_off) - 1, len(drv_off) - 1
        verdict = np.zeros(max(N, 1), dtype=np.uint8)
        kept_off = np.zeros(H + 1, dtype=np.uint32)
        self._check(
            self.lib.mcx_device_filter(self.h, _ptr(rec), N, _ptr(drv_arena), _ptr(drv_off), D, _ptr(dev_off), H,
                                       int(max_per_host), _ptr(verdict), _ptr(kept_off)),
            "mcx_device_filter",
        )
        return verdict[:N], kept_off

    def device_filter_dev(self, d_rec, N, d_drv_arena, d_drv_off, D, d_dev_off, H, max_per_host, d_verdict,
                          d_kept_off):
        """returns the number of listed devices"""
        total = ctypes.c_uint64()
        self._check(
            self.lib.mcx_device_filter_dev(self.h, d_rec, N, d_drv_arena, d_drv_off, D, d_dev_off, H,
                                           int(max_per_host), d_verdict, d_kept_off, ctypes.byref(total)),
            "mcx_device_filter_dev",
        )
        return total.value

    def fleet_compact_dev(self, d_verdict, d_S, d_s_off, N, d_dev_off, H, d_kept_off, kept_total, d_S_out,
                          d_s_off_out, d_src=None, want_bytes=True):
        """returns the bytes of the packed strings (None when want_bytes is False: no wait)"""
        nbytes = ctypes.c_uint64()
        self._check(
            self.lib.mcx_fleet_compact_dev(self.h, d_verdict, d_S, d_s_off, N, d_dev_off, H, d_kept_off,
                                           int(kept_total), d_S_out, d_s_off_out, d_src,
                                           ctypes.byref(nbytes) if want_bytes else None),
            "mcx_fleet_compact_dev",
        )
        return nbytes.value if want_bytes else None


    def fleet_inventory_roots_dev(self, d_rec, d_S, d_s_off, N, d_drv_arena, d_drv_off, D, d_dev_off, H,
                                  max_per_host, d_extra_off, d_extra_ids, X, d_safe, d_verdict, d_root_off,
                                  d_root_ids, cap, size_only=False):
        """records -> per-host module lists; returns the total (see fleet_roots_dev)"""
        total = ctypes.c_uint64()
        rc = self.lib.mcx_fleet_inventory_roots_dev(self.h, d_rec, d_S, d_s_off, N, d_drv_arena, d_drv_off, D,
                                                    d_dev_off, H, int(max_per_host), d_extra_off, d_extra_ids, X,
                                                    d_safe, d_verdict, d_root_off, d_root_ids, cap,
                                                    ctypes.byref(total))
        if not (size_only and rc == -6 and cap < total.value < 2**32):
            self._check(rc, "mcx_fleet_inventory_roots_dev")
        return total.value

    def fleet_inventory_roots_bound_dev(self, d_rec, d_S, d_s_off, N, d_drv_arena, d_drv_off, D, d_drv_mod, d_dev_off,
                                        H, max_per_host, d_extra_off, d_extra_ids, X, d_safe, d_verdict, d_root_off,
                                        d_root_ids, cap, size_only=False):
        """the same with the used drivers taken from the records' bound drivers (d_drv_mod: D module ids)"""
        total = ctypes.c_uint64()
        rc = self.lib.mcx_fleet_inventory_roots_bound_dev(self.h, d_rec, d_S, d_s_off, N, d_drv_arena, d_drv_off, D,
                                                          d_drv_mod, d_dev_off, H, int(max_per_host), d_extra_off,
                                                          d_extra_ids, X, d_safe, d_verdict, d_root_off, d_root_ids,
                                                          cap, ctypes.byref(total))
        if not (size_only and rc == -6 and cap < total.value < 2**32):
            self._check(rc, "mcx_fleet_inventory_roots_bound_dev")
        return total.value


    def fleet_inventory_roots_labeled_dev(self, d_rec, d_S, d_s_off, N, d_drv_arena, d_drv_off, D, d_drv_mod,
                                          d_mark_off, d_mark_ids, d_dev_off, H, max_per_host, d_extra_off,
                                          d_extra_ids, X, d_safe, d_verdict, d_root_off, d_root_ids, d_root_in_use,
                                          cap, size_only=False):
        """the bound-driver inventory call with the "(In Use)" byte of every listed module (d_mark_off /
        d_mark_ids: driver_marks of the driver-name table, on the device)"""
        total = ctypes.c_uint64()
        rc = self.lib.mcx_fleet_inventory_roots_labeled_dev(self.h, d_rec, d_S, d_s_off, N, d_drv_arena, d_drv_off, D,
                                                            d_drv_mod, d_mark_off, d_mark_ids, d_dev_off, H,
                                                            int(max_per_host), d_extra_off, d_extra_ids, X, d_safe,
                                                            d_verdict, d_root_off, d_root_ids, d_root_in_use, cap,
                                                            ctypes.byref(total))
        if not (size_only and rc == -6 and cap < total.value < 2**32):
            self._check(rc, "mcx_fleet_inventory_roots_labeled_dev")
        return total.value


def driver_marks(driver_names, module_names):
    """(mark_off, mark_ids): the modules each driver name marks "(In Use)" - those named like it
    and those whose name variant (rtw88_X -> rtw_X, rtlX -> rtw_X) it is"""
    lib = load_library()
    d_arena, d_off = pack_strings([n.encode() if isinstance(n, str) else n for n in driver_names])
    m_arena, m_off = pack_strings([n.encode() if isinstance(n, str) else n for n in module_names])
    D = len(driver_names)
    mark_off = np.zeros(D + 1, dtype=np.uint32)
    total = ctypes.c_uint64()
    args = (_ptr(d_arena) if len(d_arena) else None, _ptr(d_off), D, _ptr(m_arena) if len(m_arena) else None,
            _ptr(m_off), len(module_names), _ptr(mark_off))
    rc = lib.mcx_driver_marks(*args, None, 0, ctypes.byref(total))
    if rc not in (0, -6):
        raise McxError(rc, "mcx_driver_marks")
    ids = np.zeros(max(total.value, 1), dtype=np.uint32)
    if total.value:
        rc = lib.mcx_driver_marks(*args, _ptr(ids), total.value, ctypes.byref(total))
        if rc != 0:
            raise McxError(rc, "mcx_driver_marks")
    return mark_off, ids[: total.value]


def driver_modules(driver_names, module_names):
    """drv_mod[d] = id of the module named exactly like driver d (no '-' / '_' folding), NO_MATCH if none"""
    lib = load_library()
    d_arena, d_off = pack_strings([n.encode() if isinstance(n, str) else n for n in driver_names])
    m_arena, m_off = pack_strings([n.encode() if isinstance(n, str) else n for n in module_names])
    out = np.zeros(max(len(driver_names), 1), dtype=np.uint32)
    rc = lib.mcx_driver_modules(_ptr(d_arena) if len(d_arena) else None, _ptr(d_off), len(driver_names),
                                _ptr(m_arena) if len(m_arena) else None, _ptr(m_off), len(module_names), _ptr(out))
    if rc != 0:
        raise McxError(rc, "mcx_driver_modules")
    return out[: len(driver_names)]


def safe_module_names(names):
    """name stages of the reference's safe-module filter for a list of str/bytes -> uint8[n]"""
    lib = load_library()
    arena, off = pack_strings([n.encode() if isinstance(n, str) else n for n in names])
    out = np.zeros(max(len(names), 1), dtype=np.uint8)
    rc = lib.mcx_safe_module_names(_ptr(arena) if len(arena) else None, _ptr(off), len(names), _ptr(out))
    if rc != 0:
        raise McxError(rc, "mcx_safe_module_names")
    return out[: len(names)]


def tables_load(path=None, text=None):
    """replace the process-wide name tables from a file or from text"""
    lib = load_library()
    if path is not None:
        rc = lib.mcx_tables_load_file(os.fsencode(path))
    else:
        raw = text.encode() if isinstance(text, str) else text
        rc = lib.mcx_tables_load_text(raw, len(raw))
    if rc != 0:
        raise McxError(rc, "mcx_tables_load", "unreadable file or malformed table text")


def tables_reset():
    load_library().mcx_tables_reset()


def tables_dump():
    lib = load_library()
    need = ctypes.c_size_t()
    lib.mcx_tables_dump(None, 0, ctypes.byref(need))
    buf = ctypes.create_string_buffer(need.value)
    rc = lib.mcx_tables_dump(buf, need.value, ctypes.byref(need))
    if rc != 0:
        raise McxError(rc, "mcx_tables_dump")
    return buf.value.decode()


def tables_info():
    """dict(format, core, prefix, hw_alias_prefix, infra_driver, name_variant, is_default)"""
    lib = load_library()
    fmt, is_default = ctypes.c_uint32(), ctypes.c_int()
    counts = (ctypes.c_uint32 * 5)()
    lib.mcx_tables_info(ctypes.byref(fmt), counts, ctypes.byref(is_default))
    keys = ("core", "prefix", "hw_alias_prefix", "infra_driver", "name_variant")
    return dict(format=fmt.value, is_default=bool(is_default.value), **dict(zip(keys, counts)))


class ModTab:
    """Text tables (modules.alias, modules.dep, /proc/modules) parsed by the library."""

    def __init__(self):
        self.lib = load_library()
        self.h = self.lib.mcx_modtab_create()
        if not self.h:
            raise MemoryError("mcx_modtab_create")

    def close(self):
        if self.h:
            self.lib.mcx_modtab_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise McxError(rc, what, self.lib.mcx_modtab_last_error(self.h).decode(errors="replace"))

    def add_alias_text(self, text):
        b = text.encode() if isinstance(text, str) else text
        self._check(self.lib.mcx_modtab_add_alias_text(self.h, b, len(b)), "mcx_modtab_add_alias_text")

    def add_dep_text(self, text):
        b = text.encode() if isinstance(text, str) else text
        self._check(self.lib.mcx_modtab_add_dep_text(self.h, b, len(b)), "mcx_modtab_add_dep_text")

    def load_dir(self, path):
        self._check(self.lib.mcx_modtab_load_dir(self.h, path.encode()), "mcx_modtab_load_dir")

    def counts(self):
        p, v, nb = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        self._check(self.lib.mcx_modtab_counts(self.h, ctypes.byref(p), ctypes.byref(v), ctypes.byref(nb)),
                    "mcx_modtab_counts")
        return p.value, v.value, nb.value

    def alias_arrays(self):
        """copies of (arena uint8, a_off uint32[P+1], a_mod uint32[P])"""
        P, _, nb = self.counts()
        a, o, m = _vp(), _vp(), _vp()
        self._check(self.lib.mcx_modtab_alias_arrays(self.h, ctypes.byref(a), ctypes.byref(o), ctypes.byref(m)),
                    "mcx_modtab_alias_arrays")
        arena = np.frombuffer(ctypes.string_at(a.value, nb), dtype=np.uint8).copy() if nb else np.zeros(0, np.uint8)
        a_off = np.frombuffer(ctypes.string_at(o.value, 4 * (P + 1)), dtype=np.uint32).copy()
        a_mod = np.frombuffer(ctypes.string_at(m.value, 4 * P), dtype=np.uint32).copy() if P else np.zeros(0, np.uint32)
        return arena, a_off, a_mod

    def dep_arrays(self, direct=True):
        """copies of (rowptr uint32[V+1], col uint32[E])"""
        _, V, _ = self.counts()
        r, c, e = _vp(), _vp(), ctypes.c_uint32()
        self._check(self.lib.mcx_modtab_dep_arrays(self.h, int(direct), ctypes.byref(r), ctypes.byref(c),
                                                   ctypes.byref(e)), "mcx_modtab_dep_arrays")
        rowptr = np.frombuffer(ctypes.string_at(r.value, 4 * (V + 1)), dtype=np.uint32).copy()
        col = np.frombuffer(ctypes.string_at(c.value, 4 * e.value), dtype=np.uint32).copy() if e.value else \
            np.zeros(0, np.uint32)
        return rowptr, col

    def module_name(self, mid):
        s = self.lib.mcx_modtab_module_name(self.h, int(mid))
        return s.decode() if s is not None else None

    def module_id(self, name):
        return self.lib.mcx_modtab_module_id(self.h, name.encode() if isinstance(name, str) else name)

    def module_names(self):
        _, V, _ = self.counts()
        return [self.module_name(m) for m in range(V)]

    def mark_loaded(self, proc_modules_text, in_flags=None):
        """(in_flags uint8[V] with LOADED set, number of names not in the table)"""
        _, V, _ = self.counts()
        f = np.zeros(max(V, 1), dtype=np.uint8) if in_flags is None else in_flags
        b = proc_modules_text.encode() if isinstance(proc_modules_text, str) else proc_modules_text
        unknown = ctypes.c_uint32()
        self._check(self.lib.mcx_modtab_mark_loaded(self.h, b, len(b), _ptr(f), ctypes.byref(unknown)),
                    "mcx_modtab_mark_loaded")
        return f[:V], unknown.value
