"""Shared pieces of the "(In Use)" / "(Idle)" label tests (CPU: the oracle against values
worked out by hand; GPU: the library against the oracle)."""
import numpy as np

from montecarlo_amd import binding as b
from montecarlo_amd.binding import pack_strings

NO_MATCH = 0xFFFFFFFF


def variant(name):
    """the second driver name of reference montecarlo/libmontecarlo.c:552-565, or None"""
    if name.startswith("rtw88_"):
        return "rtw_" + name[6:]
    if name.startswith("rtl"):
        return "rtw_" + name[3:]
    return None


def inuse_lists(rec, dev_off, mark_off, mark_ids, D):
    """per host: the modules marked by the bound drivers of its pci / usb / pcmcia records"""
    H = len(dev_off) - 1
    off, ids = np.zeros(H + 1, dtype=np.uint32), []
    for h in range(H):
        for i in range(int(dev_off[h]), int(dev_off[h + 1])):
            d = int(rec["driver"][i])
            if rec["subsystem"][i] in (b.SUB_PCI, b.SUB_USB, b.SUB_PCMCIA) and d < D:
                ids.extend(mark_ids[mark_off[d]:mark_off[d + 1]].tolist())
        off[h + 1] = len(ids)
    return off, np.array(ids, dtype=np.uint32)


def labeled_expectation(oracle_mod, f, rec, table, names, max_per_host):
    """device list -> per-host loop with the bound drivers -> labels, all by the oracle"""
    from montecarlo_amd import driver_modules
    arena, off = pack_strings([n.encode() for n in table])
    m_arena, m_off = pack_strings([n.encode() for n in names])
    o_v, o_kept = oracle_mod.device_filter(rec, arena, off, f["dev_off"], max_per_host)
    keep = np.nonzero(o_v & 1)[0]
    raw = f["S"].tobytes()
    Sk, sk_off = pack_strings([raw[f["s_off"][i]:f["s_off"][i + 1]] for i in keep])
    first, _ = oracle_mod.alias_match(f["A"], f["a_off"], Sk, sk_off, use_twin=True)
    resolved = np.where(first != NO_MATCH, f["a_mod"][np.minimum(first, len(f["a_mod"]) - 1)], NO_MATCH).astype(np.uint32)
    by_name = {}
    for v, n in enumerate(names):
        by_name.setdefault(n, v)
    drv_mod = np.array([by_name.get(n, NO_MATCH) for n in table], dtype=np.uint32)
    assert np.array_equal(drv_mod, driver_modules(table, names))
    d = rec["driver"][keep]
    bound = np.where(d < len(table), drv_mod[np.minimum(d, len(table) - 1)], NO_MATCH).astype(np.uint32)
    hw = oracle_mod.hw_alias_flags(f["A"], f["a_off"], f["a_mod"], f["V"])
    w_off, w_ids = oracle_mod.fleet_roots(f["rowptr"], f["col"], resolved, o_kept, f["extra_off"], f["extra_ids"],
                                          f["safe"], hw, bound_mod=bound)
    w_use = oracle_mod.fleet_in_use(w_off, w_ids, rec, f["dev_off"], arena, off, m_arena, m_off)
    return dict(arena=arena, off=off, o_v=o_v, o_kept=o_kept, Sk=Sk, sk_off=sk_off, bound=bound, drv_mod=drv_mod,
                w_off=w_off, w_ids=w_ids, w_use=w_use)



HAND_WANT = {"rtw88_8821cu": 1, "rtl8xxxu": 1, "e1000_a": 0, "igb_b": 1, "tg3_c": 1, "nvme_e": 0}
HAND_LISTED = [1, 1, 1, 1, 0, 0, 0, 0]
HAND_LIMIT = 4   # usb x2, pci e1000-a, pci ahci_d are listed; the tg3_c and the hid device pass but come too late


def hand_case():
    """One host, the cases the label exists for (reference desktop/ui.py:1579-1590,
    montecarlo/libmontecarlo.c:552-565, 573-725).  Every module below is loaded through a
    listed device whose modalias resolves to it; what differs is what is bound where:
      rtw88_8821cu  listed device unbound; an unlisted usb hub interface bound to rtw_8821cu
                    (the variant name)                                    -> listed, (In Use)
      rtl8xxxu      listed device bound to rtw_8xxxu: the names differ, so it is no used
                    driver (ui.py:1567), but the variant marks it          -> listed, (In Use)
      e1000_a       listed device bound to "e1000-a" (dashes: no such module), no other link
                                                                           -> listed, (Idle)
      igb_b         bound only to a pci bridge (class 0x0604: never listed) -> listed, (In Use)
      tg3_c         bound to the 4th device, behind the list's limit of 3  -> listed, (In Use)
      ahci_d        listed device bound to it by its own name              -> a used driver: not listed
      nvme_e        bound only on the hid bus (not one of the three buses looked at), to a device
                    behind the limit (a listed one would make it a used driver) -> listed, (Idle)"""
    names = ["rtw88_8821cu", "rtl8xxxu", "e1000_a", "igb_b", "tg3_c", "ahci_d", "nvme_e"]
    pats = [b"usb:v0001*", b"usb:v0002*", b"pci:v0003*", b"pci:v0004*", b"pci:v0005*", b"pci:v0006*", b"pci:v0007*"]
    A, a_off = pack_strings(pats)
    a_mod = np.arange(7, dtype=np.uint32)
    table = ["rtw_8821cu", "rtw_8xxxu", "e1000-a", "igb_b", "tg3_c", "ahci_d", "nvme_e", "hub"]
    DRV = {n: k for k, n in enumerate(table)}

    def dev(s, sub, driver, devtype=b.DT_NONE, attrs=0, cls=0x020000, iface=0x03, parent=0x00):
        return (s, sub, driver, devtype, attrs, cls, iface, parent)

    usb_if = dict(devtype=b.DT_USB_INTERFACE, attrs=b.DA_IFACE_CLASS | b.DA_PARENT | b.DA_PARENT_CLASS)
    pci = dict(attrs=b.DA_PCI_CLASS)
    devs = [   # in enumeration order: the list takes the first HAND_LIMIT that pass
        dev(b"usb:v0001p1", b.SUB_USB, b.DRV_NONE, **usb_if),                      # resolves to rtw88_8821cu, unbound
        dev(b"usb:v0002p1", b.SUB_USB, DRV["rtw_8xxxu"], **usb_if),                # rtl8xxxu under its variant
        dev(b"pci:v0003d1", b.SUB_PCI, DRV["e1000-a"], **pci),                     # dashed driver name
        dev(b"pci:v0006d1", b.SUB_PCI, DRV["ahci_d"], **pci),                      # bound by its own name: used
        dev(b"pci:v0008d1", b.SUB_PCI, DRV["tg3_c"], **pci),                       # passes, but behind the limit
        dev(b"hid:b0003", b.SUB_HID, DRV["nvme_e"]),                               # the same, and on the hid bus
        dev(b"usb:vFFFFp1", b.SUB_USB, DRV["rtw_8821cu"], **dict(usb_if, iface=0x09)),   # hub interface: not listed
        dev(b"pci:vFFFFd1", b.SUB_PCI, DRV["igb_b"], attrs=b.DA_PCI_CLASS, cls=0x060400),  # bridge: not listed
    ]
    S, s_off = pack_strings([d[0] for d in devs])
    rec = np.zeros(len(devs), dtype=b.DEV_REC)
    for k, (_, sub, driver, devtype, attrs, cls, iface, parent) in enumerate(devs):
        rec[k] = (sub, devtype, attrs, iface, cls, parent, 0, 0, driver)
    V = len(names)
    # extras load the modules whose own device is not listed or does not resolve to them
    f = dict(A=A, a_off=a_off, a_mod=a_mod, V=V, rowptr=np.zeros(V + 1, np.uint32), col=np.zeros(0, np.uint32),
             dev_off=np.array([0, len(devs)], np.uint32), extra_off=np.array([0, 5], np.uint32),
             extra_ids=np.array([3, 4, 5, 6, 2], np.uint32), S=S, s_off=s_off, safe=np.ones(V, np.uint8), N=len(devs),
             H=1)
    return f, rec, table, names
