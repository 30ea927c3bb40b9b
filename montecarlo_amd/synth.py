"""Synthetic inputs (binding of lib/libmcxsynth.so, csrc/mcx_synth.c).

Every item is a pure function of (seed, kind, index), so ranks can generate
their own shards.  Arrays are numpy: arenas are uint8, offsets uint32.
"""
import ctypes
import os

import numpy as np

_vp = ctypes.c_void_p
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmcxsynth.so")
        if not os.path.exists(path):
            raise OSError(f"{path} not found: build it with `make -C montecarlo_amd/csrc`")
        lib = ctypes.CDLL(path)
        u64, u32 = ctypes.c_uint64, ctypes.c_uint32
        lib.mcxs_alias_table.restype = u64
        lib.mcxs_alias_table.argtypes = [u64, u32, u32, _vp, u64, _vp, _vp]
        lib.mcxs_alias_modules.restype = u32
        lib.mcxs_alias_modules.argtypes = [u32]
        lib.mcxs_modalias_strings.restype = u64
        lib.mcxs_modalias_strings.argtypes = [u64, u32, u64, u64, _vp, u64, _vp]
        lib.mcxs_dep_graph.restype = u64
        lib.mcxs_dep_graph.argtypes = [u64, u32, u32, _vp, _vp, u64]
        lib.mcxs_dep_flags.restype = None
        lib.mcxs_dep_flags.argtypes = [u64, _vp, _vp, u32, u32, _vp]
        lib.mcxs_seed_bitmap.restype = u32
        lib.mcxs_seed_bitmap.argtypes = [u64, u32, u32, _vp, _vp]
        lib.mcxs_module_names.restype = u64
        lib.mcxs_module_names.argtypes = [u64, u32, u32, _vp, u64, _vp]
        lib.mcxs_fleet_hosts.restype = u64
        lib.mcxs_fleet_hosts.argtypes = [u64, u32, u64, u64, u32, u32, _vp, _vp, _vp, u64]
        _LIB = lib
    return _LIB


def _p(a):
    return a.ctypes.data_as(_vp)


def alias_table(seed, P, j0=0, j1=None):
    """patterns j0..j1 of a P-pattern table -> (arena, a_off, a_mod)"""
    j1 = P if j1 is None else j1
    lib = _lib()
    n = j1 - j0
    a_off = np.zeros(n + 1, dtype=np.uint32)
    a_mod = np.zeros(n, dtype=np.uint32)
    nbytes = lib.mcxs_alias_table(seed, j0, j1, None, 0, None, None)
    arena = np.zeros(max(nbytes, 1), dtype=np.uint8)
    lib.mcxs_alias_table(seed, j0, j1, _p(arena), nbytes, _p(a_off), _p(a_mod))
    return arena[:nbytes], a_off, a_mod


def alias_modules(P):
    return _lib().mcxs_alias_modules(P)


def modalias_strings(seed, P, N, i0=0, i1=None):
    """strings i0..i1 of an N-string batch drawn against a P-pattern table"""
    i1 = N if i1 is None else i1
    lib = _lib()
    n = i1 - i0
    s_off = np.zeros(n + 1, dtype=np.uint32)
    nbytes = lib.mcxs_modalias_strings(seed, P, i0, i1, None, 0, None)
    if nbytes >= 2**32:
        raise ValueError("string arena beyond 4 GiB: generate in shards")
    S = np.zeros(max(nbytes, 1), dtype=np.uint8)
    lib.mcxs_modalias_strings(seed, P, i0, i1, _p(S), nbytes, _p(s_off))
    return S[:nbytes], s_off


def dep_graph(seed, V, mean_deg=10):
    lib = _lib()
    rowptr = np.zeros(V + 1, dtype=np.uint32)
    E = lib.mcxs_dep_graph(seed, V, mean_deg, None, None, 0)
    col = np.zeros(max(E, 1), dtype=np.uint32)
    lib.mcxs_dep_graph(seed, V, mean_deg, _p(rowptr), _p(col), E)
    return rowptr, col[:E]


def dep_flags(seed, rowptr, col, load_pm=50):
    V = len(rowptr) - 1
    f = np.zeros(max(V, 1), dtype=np.uint8)
    colc = np.ascontiguousarray(col if len(col) else np.zeros(1, np.uint32), dtype=np.uint32)
    _lib().mcxs_dep_flags(seed, _p(np.ascontiguousarray(rowptr, dtype=np.uint32)), _p(colc), V, load_pm, _p(f))
    return f[:V]


def seed_bitmap(seed, in_flags, n_seeds):
    V = len(in_flags)
    bm = np.zeros(max((V + 31) // 32, 1), dtype=np.uint32)
    f = np.ascontiguousarray(in_flags if V else np.zeros(1, np.uint8), dtype=np.uint8)
    got = _lib().mcxs_seed_bitmap(seed, V, n_seeds, _p(f), _p(bm))
    return bm[: (V + 31) // 32], got


def module_names(seed, V, m0=0, m1=None):
    m1 = V if m1 is None else m1
    lib = _lib()
    off = np.zeros(m1 - m0 + 1, dtype=np.uint32)
    nbytes = lib.mcxs_module_names(seed, m0, m1, None, 0, None)
    buf = np.zeros(max(nbytes, 1), dtype=np.uint8)
    lib.mcxs_module_names(seed, m0, m1, _p(buf), nbytes, _p(off))
    raw = buf[:nbytes].tobytes()
    return [raw[off[k]:off[k + 1]].decode() for k in range(m1 - m0)]


def module_names_packed(seed, V):
    """all V module names as (arena uint8, offsets uint32[V+1])"""
    lib = _lib()
    off = np.zeros(V + 1, dtype=np.uint32)
    nbytes = lib.mcxs_module_names(seed, 0, V, None, 0, None)
    buf = np.zeros(max(nbytes, 1), dtype=np.uint8)
    lib.mcxs_module_names(seed, 0, V, _p(buf), nbytes, _p(off))
    return buf[:nbytes], off


def fleet_hosts(seed, V, H, mean_dev=100, max_extra=8, h0=0, h1=None):
    """hosts h0..h1 of an H-host fleet -> (dev_cnt, extra_off, extra_ids)"""
    h1 = H if h1 is None else h1
    lib = _lib()
    n = h1 - h0
    dev_cnt = np.zeros(max(n, 1), dtype=np.uint32)
    extra_cnt = np.zeros(max(n, 1), dtype=np.uint32)
    X = lib.mcxs_fleet_hosts(seed, V, h0, h1, mean_dev, max_extra, _p(dev_cnt), _p(extra_cnt), None, 0)
    ids = np.zeros(max(X, 1), dtype=np.uint32)
    lib.mcxs_fleet_hosts(seed, V, h0, h1, mean_dev, max_extra, None, None, _p(ids), X)
    extra_off = np.zeros(n + 1, dtype=np.uint32)
    extra_off[1:] = np.cumsum(extra_cnt[:n], dtype=np.uint64).astype(np.uint32)
    return dev_cnt[:n], extra_off, ids[:X]


# ---- fleet inventory records ----------------------------------------------------

DRIVER_NAMES = [
    # the 8 names the reference drops PCI devices for, host controllers, hubs
    "pcieport", "pci_bridge", "pciehp", "pcie_aspm", "pcie_pme", "pcie_edr", "shpchp", "piix4_smbus",
    "xhci_hcd", "ehci-pci", "ehci_hcd", "ohci_hcd", "uhci_hcd", "xhci-pci-renesas", "hub", "dwc2_hcd_x",
    # near misses
    "pcieport2", "pcie", "hubs", "hcX", "shpchp_", "hc", "Hub",
    # ordinary drivers
    "e1000e", "igb", "ixgbe", "nvme", "ahci", "virtio-pci", "amdgpu", "i915", "snd_hda_intel", "iwlwifi", "r8169",
    "usbhid", "usb-storage", "uas", "btusb", "cdc_acm", "r8152", "uvcvideo", "snd-usb-audio", "hid-generic", "sd",
    "sr", "mpt3sas", "megaraid_sas", "pcmcia_x", "serial_cs",
]


def driver_names_packed():
    from .binding import pack_strings
    return pack_strings([n.encode() for n in DRIVER_NAMES])


def device_records(seed, S, s_off):
    """One packed record (binding.DEV_REC) per modalias string: the subsystem
    follows the string's bus tag, everything else is drawn so that every test of
    the reference's device list fires on a few per cent of the devices."""
    from . import binding as b
    rng = np.random.default_rng(seed)
    s_off = np.asarray(s_off, dtype=np.int64)
    N = len(s_off) - 1
    rec = np.zeros(N, dtype=b.DEV_REC)
    if N == 0:
        return rec
    S = np.asarray(S, dtype=np.uint8)
    first = S[np.minimum(s_off[:-1], max(len(S) - 1, 0))]
    r = rng.random(N)
    sub = np.full(N, b.SUB_OTHER, dtype=np.uint8)
    sub[first == ord("p")] = b.SUB_PCI            # pci: (and platform:, told apart below)
    second = S[np.minimum(s_off[:-1] + 1, max(len(S) - 1, 0))]
    sub[(first == ord("p")) & (second != ord("c"))] = b.SUB_OTHER
    sub[first == ord("u")] = b.SUB_USB
    sub[first == ord("h")] = b.SUB_HID
    other = sub == b.SUB_OTHER
    sub[other & (r < 0.35)] = b.SUB_SCSI
    sub[other & (r >= 0.35) & (r < 0.45)] = b.SUB_PCMCIA
    rec["subsystem"] = sub
    D = len(DRIVER_NAMES)
    drv = rng.integers(23, D, size=N).astype(np.uint32)          # ordinary drivers
    special = rng.random(N)
    drv[special < 0.10] = rng.integers(0, 23, size=int((special < 0.10).sum()))   # infra, hcd, hub, near misses
    drv[(special >= 0.10) & (special < 0.30)] = b.DRV_NONE
    drv[(special >= 0.30) & (special < 0.31)] = b.DRV_NOSLASH
    rec["driver"] = drv
    # pci: class codes, a tenth of them infrastructure classes
    cls = rng.integers(0, 1 << 24, size=N).astype(np.uint32)
    pick = rng.random(N)
    cls[pick < 0.04] = 0x060400 | (cls[pick < 0.04] & 0xFF)
    cls[(pick >= 0.04) & (pick < 0.07)] = 0x0C0500
    cls[(pick >= 0.07) & (pick < 0.10)] = 0x088000
    cls[(pick >= 0.10) & (pick < 0.12)] = 0x0C0330          # usb controller: base 0x0c but not SMBus
    cls[(pick >= 0.12) & (pick < 0.60)] = 0x020000
    rec["pci_class"] = cls
    attrs = np.zeros(N, dtype=np.uint8)
    a = rng.random((7, N))
    is_usb, is_pci, is_scsi = sub == b.SUB_USB, sub == b.SUB_PCI, sub == b.SUB_SCSI
    attrs[is_pci & (a[0] < 0.95)] |= b.DA_PCI_CLASS
    attrs[is_usb & (a[1] < 0.93)] |= b.DA_IFACE_CLASS
    attrs[is_usb & (a[2] < 0.95)] |= b.DA_PARENT
    attrs[is_usb & (a[3] < 0.93)] |= b.DA_PARENT_CLASS
    attrs[is_usb & (a[4] < 0.20)] |= b.DA_OWN_CLASS
    attrs[is_scsi & (a[5] < 0.80)] |= b.DA_SCSI_MODEL
    attrs[is_scsi & (a[6] < 0.80)] |= b.DA_SCSI_VENDOR
    attrs[~(is_usb | is_pci | is_scsi) & (a[0] < 0.1)] |= np.uint8(rng.integers(1, 128))
    rec["attrs"] = attrs
    classes = np.array([0x00, 0x01, 0x02, 0x03, 0x08, 0x09, 0x0A, 0x0E, 0xE0, 0xFF, 0x90, 0x80], dtype=np.uint8)
    rec["iface_class"] = classes[rng.integers(0, len(classes), size=N)]
    rec["parent_class"] = classes[rng.integers(0, len(classes), size=N)]
    rec["own_class"] = classes[rng.integers(0, len(classes), size=N)]
    dt = np.zeros(N, dtype=np.uint8)
    u = rng.random(N)
    dt[is_usb] = np.where(u[is_usb] < 0.85, b.DT_USB_INTERFACE, np.where(u[is_usb] < 0.95, b.DT_USB_DEVICE, b.DT_NONE))
    dt[is_scsi] = rng.integers(0, 7, size=int(is_scsi.sum()))
    dt[~(is_usb | is_scsi)] = rng.integers(0, 7, size=int((~(is_usb | is_scsi)).sum()))
    rec["devtype"] = dt
    return rec


def write_modules_dir(out, P=20000, seed=17, load_pm=300):
    """A /lib/modules/<version>-like directory in the formats depmod writes, with synthetic
    content: `modules.alias` ("alias <glob> <module>" lines in table order, '#' comments, blank
    lines, lines with other keywords), `modules.dep` ("<path>: <everything it needs>" with
    .ko / .ko.xz / .ko.zst / .ko.gz paths and dashed file names), and `proc_modules`, a
    /proc/modules text of a loaded set that is closed under "needs".  Returns the module names
    in id order (first appearance in modules.alias, then modules.dep)."""
    os.makedirs(out, exist_ok=True)
    A, a_off, a_mod = alias_table(seed, P)
    V = alias_modules(P)
    names, seen = [], set()
    for m, name in enumerate(module_names(seed + 1, V)):
        if name in seen:  # one file per module: a second "ext4" gets a number
            name = f"{name}_{m:x}"
        seen.add(name)
        names.append(name)
    raw = A.tobytes()
    suffix = (".ko", ".ko.xz", ".ko.zst", ".ko.gz")
    sub = ("kernel/drivers/net/ethernet/intel", "kernel/drivers/usb/serial", "kernel/sound/pci/hda", "kernel/fs/nls",
           "updates/dkms")

    def path_of(m):
        # file names keep their dashes: the module name is the base name with '-' as '_'
        fname = names[m].replace("_", "-") if m % 3 == 0 else names[m]
        return f"{sub[m % len(sub)]}/{fname}{suffix[m % len(suffix)]}"

    with open(os.path.join(out, "modules.alias"), "w") as f:
        f.write("# Aliases extracted from modules themselves.\n")
        for j in range(P):
            if j % 997 == 5:
                f.write("\n# a comment in the middle\n")
            if j % 1499 == 7:
                f.write(f"softdep {names[a_mod[j]]} pre: {names[(a_mod[j] + 1) % V]}\n")
                f.write(f"options {names[a_mod[j]]} debug=1\n")
            # the alias side may write the module with dashes as well (kmod folds them)
            mod = names[a_mod[j]].replace("_", "-") if j % 11 == 0 else names[a_mod[j]]
            f.write(f"alias {raw[a_off[j]:a_off[j + 1]].decode()} {mod}\n")
        f.write(f"alias symbol:mcx_exported_symbol {names[0]}\n")
    rowptr, col = dep_graph(seed + 2, V, 3)
    closure = [None] * V

    def needs(u):
        # iterative: everything u needs, directly or not, nearest first
        stack = [u]
        while stack:
            w = stack[-1]
            if closure[w] is not None:
                stack.pop()
                continue
            kids = [int(v) for v in col[rowptr[w]:rowptr[w + 1]]]
            todo = [v for v in kids if closure[v] is None]
            if todo:
                stack.extend(todo)
                continue
            acc, seen = [], set()
            for v in kids:
                for x in [v] + closure[v]:
                    if x not in seen:
                        seen.add(x)
                        acc.append(x)
            closure[w] = acc
            stack.pop()
        return closure[u]

    with open(os.path.join(out, "modules.dep"), "w") as f:
        for u in range(V):
            deps = " ".join(path_of(w) for w in needs(u))
            f.write(f"{path_of(u)}:{' ' + deps if deps else ''}\n")
    flags = dep_flags(seed + 3, rowptr, col, load_pm)
    with open(os.path.join(out, "proc_modules"), "w") as f:
        for u in range(V):
            if flags[u] & 1:
                f.write(f"{names[u]} {16384 + 4096 * (u % 50)} 0 - Live 0x0000000000000000\n")
        f.write("not_in_the_tables 16384 0 - Live 0x0000000000000000\n")
    return names
