"""Device-list predicates over packed records: the CPU oracle against verdicts
worked out by hand from the reference's tests on one sysfs node
(montecarlo/libmontecarlo.c:233-440, 729-874).  The reference holds no fixtures
for these functions, so the hand table below is what pins the oracle."""
import os

import numpy as np

import oracle
from montecarlo_amd import binding as b
from montecarlo_amd import synth
from montecarlo_amd.binding import pack_strings

DRV = ["pcieport", "xhci_hcd", "hub", "e1000e", "usbhid", "piix4_smbus", "ehci-pci", "hubby", "usb-storage"]
ALL_USB = b.DA_IFACE_CLASS | b.DA_PARENT | b.DA_PARENT_CLASS

# (subsystem, devtype, attrs, iface_class, pci_class, parent_class, own_class, driver) -> (listed, excluded)
HAND = [
    # PCI: class tests need a readable class attribute
    ((b.SUB_PCI, 0, b.DA_PCI_CLASS, 0, 0x020000, 0, 0, 3), (1, 0)),            # ethernet, e1000e
    ((b.SUB_PCI, 0, b.DA_PCI_CLASS, 0, 0x060400, 0, 0, b.DRV_NONE), (0, 0)),   # PCI bridge
    ((b.SUB_PCI, 0, b.DA_PCI_CLASS, 0, 0x0C0500, 0, 0, b.DRV_NONE), (0, 0)),   # SMBus
    ((b.SUB_PCI, 0, b.DA_PCI_CLASS, 0, 0x0C0330, 0, 0, 1), (1, 0)),            # xHCI controller: listed (hcd test is usb only)
    ((b.SUB_PCI, 0, b.DA_PCI_CLASS, 0, 0x088000, 0, 0, b.DRV_NONE), (0, 0)),   # system peripheral
    ((b.SUB_PCI, 0, 0, 0, 0x060400, 0, 0, b.DRV_NONE), (1, 0)),                # class unreadable: no class test
    ((b.SUB_PCI, 0, 0, 0, 0, 0, 0, 0), (0, 0)),                                # pcieport driver
    ((b.SUB_PCI, 0, 0, 0, 0, 0, 0, 5), (0, 0)),                                # piix4_smbus driver
    ((b.SUB_PCI, 0, 0, 0, 0, 0, 0, b.DRV_NOSLASH), (0, 0)),                    # link target without '/'
    ((b.SUB_PCI, 0, b.DA_PCI_CLASS, 0, 0x010802, 0, 8, b.DRV_NONE), (1, 0)),   # own_class without its attribute bit
    # USB: only interfaces with a usb_device parent, no hubs, no host controllers
    ((b.SUB_USB, b.DT_USB_INTERFACE, ALL_USB, 0x03, 0, 0x00, 0, 4), (1, 0)),
    ((b.SUB_USB, b.DT_USB_DEVICE, ALL_USB, 0x03, 0, 0x00, 0, 4), (0, 0)),
    ((b.SUB_USB, b.DT_NONE, ALL_USB, 0x03, 0, 0x00, 0, 4), (0, 0)),
    ((b.SUB_USB, b.DT_USB_INTERFACE, ALL_USB, 0x09, 0, 0x00, 0, 2), (0, 0)),   # hub interface
    ((b.SUB_USB, b.DT_USB_INTERFACE, ALL_USB, 0x03, 0, 0x09, 0, 4), (0, 0)),   # parent is a hub
    ((b.SUB_USB, b.DT_USB_INTERFACE, b.DA_IFACE_CLASS, 0x03, 0, 0, 0, 4), (0, 0)),   # no parent
    ((b.SUB_USB, b.DT_USB_INTERFACE, b.DA_PARENT, 0x09, 0, 0x09, 0, 4), (1, 0)),     # class attributes unreadable
    ((b.SUB_USB, b.DT_USB_INTERFACE, ALL_USB, 0x03, 0, 0x00, 0, 1), (0, 0)),   # xhci_hcd
    ((b.SUB_USB, b.DT_USB_INTERFACE, ALL_USB, 0x03, 0, 0x00, 0, 2), (0, 0)),   # hub driver
    ((b.SUB_USB, b.DT_USB_INTERFACE, ALL_USB, 0x03, 0, 0x00, 0, 7), (1, 0)),   # "hubby" is not "hub"
    ((b.SUB_USB, b.DT_USB_INTERFACE, ALL_USB, 0x03, 0, 0x00, 0, 6), (1, 0)),   # ehci-pci: no "hcd" in it
    ((b.SUB_USB, b.DT_USB_INTERFACE, ALL_USB, 0x03, 0, 0x00, 0, b.DRV_NOSLASH), (1, 0)),  # "unknown"
    ((b.SUB_USB, b.DT_USB_INTERFACE, ALL_USB, 0x08, 0, 0x00, 0, 8), (1, 1)),   # mass storage interface
    ((b.SUB_USB, b.DT_USB_INTERFACE, ALL_USB, 0x03, 0, 0x08, 0, 8), (1, 1)),   # mass storage parent
    ((b.SUB_USB, b.DT_USB_DEVICE, ALL_USB | b.DA_OWN_CLASS, 0x03, 0, 0x00, 0x08, 8), (0, 1)),
    ((b.SUB_USB, b.DT_USB_INTERFACE, b.DA_IFACE_CLASS | b.DA_PARENT_CLASS, 0x03, 0, 0x08, 0, 8), (0, 0)),  # parent class without parent
    # SCSI
    ((b.SUB_SCSI, b.DT_OTHER, b.DA_SCSI_MODEL | b.DA_SCSI_VENDOR, 0, 0, 0, 0, b.DRV_NONE), (1, 0)),
    ((b.SUB_SCSI, b.DT_OTHER, b.DA_SCSI_MODEL, 0, 0, 0, 0, b.DRV_NONE), (1, 0)),
    ((b.SUB_SCSI, b.DT_OTHER, 0, 0, 0, 0, 0, b.DRV_NONE), (0, 0)),
    ((b.SUB_SCSI, b.DT_NONE, b.DA_SCSI_MODEL, 0, 0, 0, 0, b.DRV_NONE), (0, 0)),
    ((b.SUB_SCSI, b.DT_SCSI_HOST, b.DA_SCSI_MODEL, 0, 0, 0, 0, b.DRV_NONE), (0, 0)),
    ((b.SUB_SCSI, b.DT_SCSI_TARGET, b.DA_SCSI_MODEL, 0, 0, 0, 0, b.DRV_NONE), (0, 0)),
    ((b.SUB_SCSI, b.DT_SCSI_GENERIC, b.DA_SCSI_MODEL, 0, 0, 0, 0, b.DRV_NONE), (0, 0)),
    ((b.SUB_SCSI, b.DT_OTHER, b.DA_SCSI_VENDOR, 0, 0, 0, 0, 1), (1, 0)),       # hcd driver name outside usb
    # HID, PCMCIA: always; anything else: never
    ((b.SUB_HID, 0, 0, 0, 0, 0, 0, b.DRV_NONE), (1, 0)),
    ((b.SUB_HID, 0, 0, 0, 0x060400, 0, 0, 0), (1, 0)),
    ((b.SUB_PCMCIA, 0, 0, 0, 0, 0, 0, 2), (1, 0)),
    ((b.SUB_OTHER, 0, 0, 0, 0, 0, 0, 3), (0, 0)),
    ((7, b.DT_USB_INTERFACE, ALL_USB, 3, 0, 0, 0, 4), (0, 0)),
]


def hand_records():
    rec = np.zeros(len(HAND), dtype=b.DEV_REC)
    for k, (f, _) in enumerate(HAND):
        rec[k] = (f[0], f[1], f[2], f[3], f[4], f[5], f[6], 0, f[7])
    want = np.array([w[0] | (w[1] << 1) for _, w in HAND], dtype=np.uint8)
    return rec, want


def test_record_layout():
    assert b.DEV_REC.itemsize == 16
    assert [b.DEV_REC.fields[n][1] for n in ("subsystem", "devtype", "attrs", "iface_class", "pci_class",
                                             "parent_class", "own_class", "driver")] == [0, 1, 2, 3, 4, 8, 9, 12]


def test_oracle_hand_table():
    rec, want = hand_records()
    arena, off = pack_strings([d.encode() for d in DRV])
    v, kept = oracle.device_filter(rec, arena, off, np.array([0, len(rec)], np.uint32), 0)
    assert v.tolist() == want.tolist()
    assert kept.tolist() == [0, int((want & 1).sum())]


def test_oracle_per_host_limit():
    rec, want = hand_records()
    arena, off = pack_strings([d.encode() for d in DRV])
    dev_off = np.array([0, 0, 12, 12, 30, len(rec)], dtype=np.uint32)
    for cap in (1, 3, 64):
        v, kept = oracle.device_filter(rec, arena, off, dev_off, cap)
        for h in range(5):
            lo, hi = dev_off[h], dev_off[h + 1]
            passing = np.nonzero(want[lo:hi] & 1)[0][:cap] + lo
            assert np.nonzero(v[lo:hi] & 1)[0].tolist() == (passing - lo).tolist()
            assert kept[h + 1] - kept[h] == len(passing)
        assert np.array_equal(v & 2, want & 2)          # the limit does not touch the exclusion bit


def test_synthetic_records_cover_every_branch():
    S, s_off = synth.modalias_strings(9, 2000, 20000)
    rec = synth.device_records(9, S, s_off)
    arena, off = synth.driver_names_packed()
    v, kept = oracle.device_filter(rec, arena, off, np.array([0, len(rec)], np.uint32), 0)
    assert kept[1] == (v & 1).sum() and 0.3 < (v & 1).mean() < 0.9 and (v & 2).any()
    for sub in (b.SUB_USB, b.SUB_PCI, b.SUB_SCSI):
        m = rec["subsystem"] == sub
        assert (v[m] & 1).any() and not (v[m] & 1).all()
    assert (v[rec["subsystem"] == b.SUB_HID] & 1).all() and not (v[rec["subsystem"] == b.SUB_OTHER] & 1).any()


def records_from_tree(root):
    """What an inventory agent would pack for every node of the five
    subsystems of a sysfs tree: one record per device, the driver names found"""
    subs = {"usb": b.SUB_USB, "pci": b.SUB_PCI, "hid": b.SUB_HID, "scsi": b.SUB_SCSI, "pcmcia": b.SUB_PCMCIA}
    devtypes = {"usb_device": b.DT_USB_DEVICE, "usb_interface": b.DT_USB_INTERFACE, "scsi_host": b.DT_SCSI_HOST,
                "scsi_target": b.DT_SCSI_TARGET, "scsi_generic": b.DT_SCSI_GENERIC}

    def attr(d, name):
        try:
            with open(os.path.join(d, name)) as f:
                return f.readline().rstrip("\n")
        except OSError:
            return None

    def devtype_of(d):
        for line in open(os.path.join(d, "uevent")):
            if line.startswith("DEVTYPE="):
                return line[8:].strip()
        return None

    paths, recs, drivers = [], [], []
    for sub, code in subs.items():
        bd = os.path.join(root, "sys", "bus", sub, "devices")
        if not os.path.isdir(bd):
            continue
        for name in sorted(os.listdir(bd)):
            d = os.path.realpath(os.path.join(bd, name))
            attrs, f = 0, dict(iface=0, pci=0, parent=0, own=0)
            dt = devtype_of(d)
            devtype = b.DT_NONE if dt is None else devtypes.get(dt, b.DT_OTHER)
            if attr(d, "class") is not None:
                attrs |= b.DA_PCI_CLASS
                f["pci"] = int(attr(d, "class"), 16)
            if attr(d, "bInterfaceClass") is not None:
                attrs |= b.DA_IFACE_CLASS
                f["iface"] = int(attr(d, "bInterfaceClass"), 16)
            if attr(d, "bDeviceClass") is not None:
                attrs |= b.DA_OWN_CLASS
                f["own"] = int(attr(d, "bDeviceClass"), 16)
            up = os.path.dirname(d)
            while up.startswith(os.path.join(root, "sys", "devices")):       # nearest usb_device above
                if os.path.exists(os.path.join(up, "uevent")) and devtype_of(up) == "usb_device" and \
                        os.path.basename(os.path.realpath(os.path.join(up, "subsystem"))) == "usb":
                    attrs |= b.DA_PARENT
                    if attr(up, "bDeviceClass") is not None:
                        attrs |= b.DA_PARENT_CLASS
                        f["parent"] = int(attr(up, "bDeviceClass"), 16)
                    break
                up = os.path.dirname(up)
            if attr(d, "model") is not None:
                attrs |= b.DA_SCSI_MODEL
            if attr(d, "vendor") is not None and code == b.SUB_SCSI:
                attrs |= b.DA_SCSI_VENDOR
            drv = b.DRV_NONE
            link = os.path.join(d, "driver")
            if os.path.islink(link):
                dn = os.path.basename(os.readlink(link))
                if dn not in drivers:
                    drivers.append(dn)
                drv = drivers.index(dn)
            paths.append(d)
            recs.append((code, devtype, attrs, f["iface"], f["pci"], f["parent"], f["own"], 0, drv))
    return paths, np.array(recs, dtype=b.DEV_REC), drivers


def test_packed_filter_agrees_with_scalar_walk(tree):
    """two restatements of the reference against each other: the library's scalar
    mc_list_all_devices / mc_is_excluded_device walking a fake sysfs tree, and the
    oracle's predicates over records packed from the same tree"""
    import ctypes

    from montecarlo_amd import binding
    from test_legacy_abi import MCDeviceInfo

    # the test build of the library: the only one that looks below MCX_SYSFS_ROOT
    lib = ctypes.CDLL(os.path.join(os.path.dirname(binding.library_path()), "libmontecarlo_testseams.so"))
    lib.mc_list_all_devices.argtypes = [ctypes.POINTER(MCDeviceInfo), ctypes.c_int]
    lib.mc_is_excluded_device.argtypes = [ctypes.c_char_p]
    buf = (MCDeviceInfo * 64)()
    n = lib.mc_list_all_devices(buf, 64)
    listed_scalar = {os.path.realpath(buf[k].syspath.decode()) for k in range(n)}
    paths, rec, drivers = records_from_tree(tree)
    assert len(rec) == 24
    arena, off = pack_strings([d.encode() for d in drivers])
    v, kept = oracle.device_filter(rec, arena, off, np.array([0, len(rec)], np.uint32), 0)
    listed_packed = {p for p, x in zip(paths, v) if x & 1}
    assert listed_packed == listed_scalar and kept[1] == n
    for p, x in zip(paths, v):
        assert lib.mc_is_excluded_device(p.encode()) == ((x >> 1) & 1), p
