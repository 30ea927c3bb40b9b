"""The scalar mc_* ABI (include/libmontecarlo.h) on a fake sysfs / procfs tree.

The reference cannot be pointed at a fake tree (its paths are absolute, reference
montecarlo/libmontecarlo.c:449,480,501,592) and cannot be built in this image
(libudev.h is absent), so the expected values below are derived by hand from the
reference source, function by function; the struct layout and the behaviour on
this container's real /sys are those recorded from the reference itself when the
survey ran it here (496-byte record, 7 candidate drivers, 6 virtio PCI devices)."""
import ctypes
import os

import pytest


class MCDeviceInfo(ctypes.Structure):
    # reference heads/libmontecarlo.h:14-20, bound the way desktop/ui.py binds it
    _fields_ = [("syspath", ctypes.c_char * 256), ("vidpid", ctypes.c_char * 32), ("product", ctypes.c_char * 128),
                ("driver", ctypes.c_char * 64), ("subsystem", ctypes.c_char * 16)]


@pytest.fixture(scope="module")
def lib():
    from montecarlo_amd import binding

    lib = binding.load_library()
    lib.mc_read_sysattr.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    lib.mc_get_ids.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    lib.mc_get_ids.restype = None
    lib.mc_list_all_devices.argtypes = [ctypes.POINTER(MCDeviceInfo), ctypes.c_int]
    lib.mc_get_device_subsystem.argtypes = [ctypes.c_char_p]
    lib.mc_get_device_subsystem.restype = ctypes.c_char_p
    for name in ("mc_try_load_driver", "mc_unload_driver", "mc_dmesg_has_activity", "mc_module_has_holders",
                 "mc_get_module_refcount", "mc_driver_is_in_use", "mc_dev_has_driver", "mc_is_excluded_device"):
        getattr(lib, name).argtypes = [ctypes.c_char_p]
    lib.mc_is_infrastructure_device.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.mc_list_loaded_modules.argtypes = [ctypes.c_char_p, ctypes.c_int]
    return lib


def mkdev(root, devpath, bus, attrs=None, uevent="", driver=None):
    """a device directory under <root>/sys/devices with its bus entry"""
    d = os.path.join(root, "sys", "devices", devpath)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "uevent"), "w") as f:
        f.write(uevent)
    for k, v in (attrs or {}).items():
        with open(os.path.join(d, k), "w") as f:
            f.write(v + "\n")
    busdir = os.path.join(root, "sys", "bus", bus)
    os.makedirs(os.path.join(busdir, "devices"), exist_ok=True)
    os.makedirs(os.path.join(busdir, "drivers"), exist_ok=True)
    os.symlink(busdir, os.path.join(d, "subsystem"))
    name = os.path.basename(devpath)
    os.symlink(d, os.path.join(busdir, "devices", name))
    if driver:
        dd = os.path.join(busdir, "drivers", driver)
        os.makedirs(dd, exist_ok=True)
        os.symlink(dd, os.path.join(d, "driver"))
        os.symlink(d, os.path.join(dd, name))
        for fn in ("bind", "unbind", "uevent"):
            open(os.path.join(dd, fn), "w").close()
    return d


@pytest.fixture()
def tree(tmp_path, monkeypatch):
    root = str(tmp_path)
    p = "pci0000:00/"
    mkdev(root, p + "0000:00:00.0", "pci", {"class": "0x060000", "vendor": "0x8086", "device": "0x0c00"})
    mkdev(root, p + "0000:00:1c.0", "pci", {"class": "0x060400", "vendor": "0x8086", "device": "0x9d10"}, driver="pcieport")
    mkdev(root, p + "0000:00:1f.2", "pci", {"class": "0x010601", "vendor": "0x8086", "device": "0x9d03"}, driver="ahci")
    mkdev(root, p + "0000:00:1f.3", "pci", {"class": "0x0c0500", "vendor": "0x8086", "device": "0x9d23"})
    mkdev(root, p + "0000:02:00.0", "pci", {"class": "0x020000", "vendor": "0x10ec", "device": "0x8168", "label": "Onboard LAN"})
    mkdev(root, p + "0000:03:00.0", "pci", {"class": "0x088000", "vendor": "0x8086", "device": "0x1234"})
    mkdev(root, p + "0000:04:00.0", "pci", {"class": "0x030000", "vendor": "0x1002", "device": "0x67df"}, driver="pcie_pme")
    mkdev(root, p + "0000:05:00.0", "pci", {"vendor": "0x14e4"})  # no class, no device id
    u = p + "0000:00:14.0/usb1"
    mkdev(root, u, "usb", {"bDeviceClass": "09", "idVendor": "1d6b", "idProduct": "0002"}, "DEVTYPE=usb_device\n", "usb")
    mkdev(root, u + "/1-0:1.0", "usb", {"bInterfaceClass": "09", "bInterfaceNumber": "00"}, "DEVTYPE=usb_interface\n", "hub")
    mkdev(root, u + "/1-1", "usb", {"bDeviceClass": "00", "idVendor": "046d", "idProduct": "c52b", "product": "USB Receiver",
                                     "manufacturer": "Logitech"}, "DEVTYPE=usb_device\n", "usb")
    mkdev(root, u + "/1-1/1-1:1.0", "usb", {"bInterfaceClass": "03", "bInterfaceNumber": "00"}, "DEVTYPE=usb_interface\n", "usbhid")
    mkdev(root, u + "/1-1/1-1:1.1", "usb", {"bInterfaceClass": "03"}, "DEVTYPE=usb_interface\n")
    mkdev(root, u + "/1-2", "usb", {"bDeviceClass": "00", "idVendor": "0781", "idProduct": "5567"}, "DEVTYPE=usb_device\n", "usb")
    mkdev(root, u + "/1-2/1-2:1.0", "usb", {"bInterfaceClass": "08", "bInterfaceNumber": "00"}, "DEVTYPE=usb_interface\n", "usb-storage")
    mkdev(root, u + "/1-3", "usb", {"bDeviceClass": "ff"}, "DEVTYPE=usb_device\n", "usb")
    mkdev(root, u + "/1-3/1-3:1.0", "usb", {"bInterfaceClass": "ff", "bInterfaceNumber": "00"}, "DEVTYPE=usb_interface\n", "dummy_hcd")
    mkdev(root, u + "/1-4", "usb", {"bDeviceClass": "08", "idVendor": "0bda", "idProduct": "c811"}, "DEVTYPE=usb_device\n", "usb")
    mkdev(root, u + "/1-4/1-4:1.0", "usb", {"bInterfaceClass": "ff", "bInterfaceNumber": "00"}, "DEVTYPE=usb_interface\n", "rtw_8821cu")
    mkdev(root, u + "/1-1/1-1:1.0/0003:046D:C52B.0001", "hid", {}, "", "hid-generic")
    s = p + "0000:00:1f.2/ata1/host0"
    mkdev(root, s, "scsi", {}, "DEVTYPE=scsi_host\n")
    mkdev(root, s + "/target0:0:0", "scsi", {}, "DEVTYPE=scsi_target\n")
    mkdev(root, s + "/target0:0:0/0:0:0:0", "scsi", {"model": "Samsung SSD 860 ", "vendor": "ATA     "}, "DEVTYPE=scsi_device\n", "sd")
    mkdev(root, s + "/target0:0:0/0:0:0:1", "scsi", {}, "DEVTYPE=scsi_device\n")  # no model, no vendor
    # things in drivers/ that are not driver directories
    open(os.path.join(root, "sys/bus/pci/drivers/README"), "w").close()
    os.symlink("/nonexistent", os.path.join(root, "sys/bus/pci/drivers/alink"))
    os.makedirs(os.path.join(root, "sys/bus/pci/drivers/e1000"))
    for fn in ("bind", "unbind", "new_id", "remove_id", "module"):
        open(os.path.join(root, "sys/bus/pci/drivers/e1000", fn), "w").close()
    os.makedirs(os.path.join(root, "sys/bus/i2c/drivers/dummy"))
    # modules
    for mod, holders, refcnt in (("snd", ["snd_pcm", "snd_timer"], "5"), ("e1000", [], "0"), ("weird", None, "x")):
        md = os.path.join(root, "sys/module", mod)
        os.makedirs(md)
        if holders is not None:
            os.makedirs(os.path.join(md, "holders"))
            for h in holders:
                os.symlink("/nonexistent", os.path.join(md, "holders", h))
        with open(os.path.join(md, "refcnt"), "w") as f:
            f.write(refcnt + "\n")
    os.makedirs(os.path.join(root, "proc"))
    with open(os.path.join(root, "proc/modules"), "w") as f:
        f.write("snd_pcm 106496 2 snd_hda_intel, Live 0x0000000000000000\n"
                "snd 90112 9 snd_pcm,snd_timer, Live 0x0000000000000000\n"
                "e1000 151552 0 - Live 0x0000000000000000\n"
                "a_module_name_that_is_much_longer_than_sixty_three_characters_in_total 1 0 - Live 0x0\n")
    monkeypatch.setenv("MCX_SYSFS_ROOT", root)
    return root


def test_struct_layout():
    assert ctypes.sizeof(MCDeviceInfo) == 496
    offs = [getattr(MCDeviceInfo, f).offset for f, _ in MCDeviceInfo._fields_]
    assert offs == [0, 256, 288, 416, 480]


def test_candidate_drivers(lib, tree):
    buf = ((ctypes.c_char * 128) * 256)()
    n = lib.mc_list_candidate_drivers(buf, 256)
    names = sorted(bytes(buf[i]).split(b"\0")[0].decode() for i in range(n))
    # directories only (lstat), of the 8 scanned buses; hid drivers dir holds hid-generic
    assert names == sorted(["pcieport", "ahci", "pcie_pme", "e1000", "usb", "hub", "usbhid", "usb-storage", "dummy_hcd",
                            "rtw_8821cu", "hid-generic", "sd", "dummy"])
    assert lib.mc_list_candidate_drivers(buf, 3) == 3
    assert lib.mc_list_candidate_drivers(buf, 0) == 0


def test_list_all_devices(lib, tree):
    devs = (MCDeviceInfo * 64)()
    n = lib.mc_list_all_devices(devs, 64)
    got = [(os.path.basename(d.syspath.decode()), d.vidpid.decode(), d.product.decode(), d.driver.decode(),
            d.subsystem.decode()) for d in devs[:n]]
    assert got == [
        # syspath order; bridges, SMBus, system peripherals, pcieport/pcie_pme-bound devices are gone
        ("1-1:1.0", "046d:c52b", "Logitech USB Receiver (If: 00)", "usbhid", "usb"),
        ("0003:046D:C52B.0001", "HID", "HID: HID Device", "hid-generic", "hid"),
        ("1-1:1.1", "046d:c52b", "Logitech USB Receiver", "None", "usb"),
        ("1-2:1.0", "0781:5567", " Unknown Device (If: 00)", "usb-storage", "usb"),
        # 1-3:1.0 is bound to *hcd*: skipped; usb1 / 1-0:1.0 are hubs or not interfaces
        ("1-4:1.0", "0bda:c811", " Unknown Device (If: 00)", "rtw_8821cu", "usb"),
        ("0000:00:1f.2", "0x8086:0x9d03", "PCI Device 0000:00:1f.2", "ahci", "pci"),
        ("0:0:0:0", "SCSI", "ATA      Samsung SSD 860 ", "sd", "scsi"),  # a child path sorts after its parent
        ("0000:02:00.0", "0x10ec:0x8168", "Onboard LAN", "None", "pci"),
        ("0000:05:00.0", "????:????", "PCI Device 0000:05:00.0", "None", "pci"),
    ]
    # the count stops silently at max
    assert lib.mc_list_all_devices(devs, 2) == 2


def test_predicates(lib, tree):
    dev = lambda rel: os.path.join(tree, "sys/devices", rel).encode()  # noqa: E731
    u = "pci0000:00/0000:00:14.0/usb1"
    assert lib.mc_is_infrastructure_device(dev("pci0000:00/0000:00:00.0"), b"pci") == 1
    assert lib.mc_is_infrastructure_device(dev("pci0000:00/0000:00:1f.3"), b"pci") == 1
    assert lib.mc_is_infrastructure_device(dev("pci0000:00/0000:03:00.0"), b"pci") == 1
    assert lib.mc_is_infrastructure_device(dev("pci0000:00/0000:04:00.0"), b"pci") == 1  # by driver name
    assert lib.mc_is_infrastructure_device(dev("pci0000:00/0000:00:1f.2"), b"pci") == 0
    assert lib.mc_is_infrastructure_device(dev("pci0000:00/0000:00:1f.2"), b"usb") == 0
    assert lib.mc_is_infrastructure_device(dev("pci0000:00/nonexistent"), b"pci") == 0
    assert lib.mc_is_infrastructure_device(None, b"pci") == 0
    s = "pci0000:00/0000:00:1f.2/ata1/host0"
    assert lib.mc_is_infrastructure_device(dev(s), b"scsi") == 1
    assert lib.mc_is_infrastructure_device(dev(s + "/target0:0:0"), b"scsi") == 1
    assert lib.mc_is_infrastructure_device(dev(s + "/target0:0:0/0:0:0:0"), b"scsi") == 0
    assert lib.mc_is_infrastructure_device(dev(s + "/target0:0:0/0:0:0:1"), b"scsi") == 1
    # mass storage: on the interface, on the device itself, on the parent device
    assert lib.mc_is_excluded_device(dev(u + "/1-2/1-2:1.0")) == 1
    assert lib.mc_is_excluded_device(dev(u + "/1-4")) == 1
    assert lib.mc_is_excluded_device(dev(u + "/1-4/1-4:1.0")) == 1
    assert lib.mc_is_excluded_device(dev(u + "/1-1/1-1:1.0")) == 0
    assert lib.mc_is_excluded_device(dev("nowhere")) == 0
    assert lib.mc_dev_has_driver(dev(u + "/1-1/1-1:1.0")) == 1
    assert lib.mc_dev_has_driver(dev(u + "/1-1/1-1:1.1")) == 0
    assert lib.mc_dev_has_driver(dev("nowhere")) == 0
    # the reference's length arithmetic makes this NULL for every real syspath
    assert lib.mc_get_device_subsystem(dev("pci0000:00/0000:00:1f.2")) is None
    assert lib.mc_get_device_subsystem(dev("nowhere")) == b"unknown"


def test_sysattr_and_ids(lib, tree):
    buf = ctypes.create_string_buffer(64)
    d = os.path.join(tree, "sys/devices/pci0000:00/0000:00:14.0/usb1/1-1")
    assert lib.mc_read_sysattr(os.path.join(d, "product").encode(), buf, 64) == 1 and buf.value == b"USB Receiver"
    assert lib.mc_read_sysattr(os.path.join(d, "nope").encode(), buf, 64) == 0
    v, p = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    lib.mc_get_ids(d.encode(), v, p)
    assert (v.value, p.value) == (b"046d", b"c52b")
    lib.mc_get_ids(os.path.join(tree, "sys/devices/pci0000:00/0000:00:1f.2").encode(), v, p)
    assert (v.value, p.value) == (b"0000", b"0000")


def test_module_state(lib, tree):
    assert lib.mc_module_has_holders(b"snd") == 1
    assert lib.mc_module_has_holders(b"e1000") == 0
    assert lib.mc_module_has_holders(b"weird") == 0      # no holders directory
    assert lib.mc_module_has_holders(b"absent") == 0
    assert lib.mc_get_module_refcount(b"snd") == 5
    assert lib.mc_get_module_refcount(b"e1000") == 0
    assert lib.mc_get_module_refcount(b"weird") == -1    # unparsable
    assert lib.mc_get_module_refcount(b"absent") == -1
    assert lib.mc_driver_is_in_use(b"ahci") == 1          # a device symlink under pci/drivers/ahci
    assert lib.mc_driver_is_in_use(b"e1000") == 0         # only bind/unbind/new_id/remove_id/module files
    assert lib.mc_driver_is_in_use(b"snd") == 1           # holders
    assert lib.mc_driver_is_in_use(b"rtw88_8821cu") == 1  # via its rtw_8821cu variant on the usb bus
    assert lib.mc_driver_is_in_use(b"rtl8821cu") == 1     # rtl... -> rtw_...
    assert lib.mc_driver_is_in_use(b"absent") == 0
    assert lib.mc_driver_is_in_use(b"") == 0
    assert lib.mc_driver_is_in_use(None) == 0
    out = ctypes.create_string_buffer(40960)
    n = lib.mc_list_loaded_modules(out, 40960)
    names = out.raw.split(b"\0")[:n]
    # %63s cuts the long name at 63 bytes
    assert n == 4 and names[:3] == [b"snd_pcm", b"snd", b"e1000"] and len(names[3]) == 63
    # the reference keeps a name only while written + len + 1 < max_size (strictly)
    small = ctypes.create_string_buffer(16)
    assert lib.mc_list_loaded_modules(small, 13) == 2     # "snd_pcm\0snd\0" is 12 bytes
    assert lib.mc_list_loaded_modules(small, 12) == 1
    assert lib.mc_list_loaded_modules(small, 8) == 0


def test_load_unload_and_dmesg(lib, tree, monkeypatch, capfd):
    monkeypatch.setenv("MCX_MODPROBE", "true")
    assert lib.mc_try_load_driver(b"e1000") == 1
    assert lib.mc_unload_driver(b"e1000") == 1
    monkeypatch.setenv("MCX_MODPROBE", "false")
    assert lib.mc_try_load_driver(b"e1000") == 0
    assert lib.mc_unload_driver(b"e1000") == -1
    # a name a shell would act on never reaches one
    monkeypatch.setenv("MCX_MODPROBE", "true")
    assert lib.mc_try_load_driver(b"e1000; touch /tmp/pwned") == 0
    assert lib.mc_unload_driver(b"$(reboot)") == -1
    monkeypatch.setenv("MCX_DMESG_CMD", "printf 'usb 1-1: new device\\ne1000: link up\\n'")
    assert lib.mc_dmesg_has_activity(b"e1000") == 1
    assert lib.mc_dmesg_has_activity(b"iwlwifi") == 0


def test_real_sysfs_matches_reference_probe(lib, monkeypatch):
    """On this image's own /sys the reference library, when it was built and run
    here during the survey, listed 7 candidate drivers, 6 virtio PCI devices and no
    loaded modules.  Only meaningful where that same /sys is present."""
    monkeypatch.delenv("MCX_SYSFS_ROOT", raising=False)
    if not os.path.isdir("/sys/bus/pci/drivers/virtio-pci") or os.path.exists("/proc/modules"):
        pytest.skip("not the container the reference was probed in")
    buf = ((ctypes.c_char * 128) * 256)()
    if lib.mc_list_candidate_drivers(buf, 256) != 7:
        pytest.skip("different virtual hardware")
    devs = (MCDeviceInfo * 64)()
    n = lib.mc_list_all_devices(devs, 64)
    assert n == 6
    assert devs[0].vidpid == b"0x1af4:0x1045" and devs[0].driver == b"virtio-pci" and devs[0].subsystem == b"pci"
    out = ctypes.create_string_buffer(4096)
    assert lib.mc_list_loaded_modules(out, 4096) == 0
    assert (lib.mc_module_has_holders(b"bridge"), lib.mc_get_module_refcount(b"bridge"),
            lib.mc_driver_is_in_use(b"bridge")) == (0, -1, 0)
