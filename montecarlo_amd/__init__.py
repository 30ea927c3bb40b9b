"""montecarlo_amd - MI355X build of the libmontecarlo batch hot path.

The compute lives in ``lib/libmontecarlo.so`` (hand-written HIP kernels for
gfx950 behind a plain C ABI, ``include/mcx.h``).  This package is the thin
ctypes binding around it, the same way the reference's own Python code binds
``libmontecarlo.so`` (reference desktop/ui.py:53-103).  There is no CPU
fallback: without the built library or without a HIP device every call raises.
"""
from .binding import (  # noqa: F401
    McxError,
    Context,
    ModTab,
    load_library,
    library_path,
    NO_MATCH,
    F_LOADED,
    F_BOUND,
    F_SAFE,
    O_HAS_HOLDERS,
    O_IN_USE,
    O_ROOT,
    CLOSURE_HOLDERS,
    CLOSURE_DEPS,
    TABLE_NORMALIZE_DASH,
    safe_module_names,
    driver_modules,
    driver_marks,
    pack_strings,
    DEV_REC,
    tables_load,
    tables_reset,
    tables_dump,
    tables_info,
)
from . import synth  # noqa: F401

__version__ = "0.1.0"
