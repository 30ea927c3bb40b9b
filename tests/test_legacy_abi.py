"""The scalar mc_* ABI (include/libmontecarlo.h) on a fake sysfs / procfs tree.

The reference cannot be pointed at a fake tree (its paths are absolute, reference
montecarlo/libmontecarlo.c:449,480,501,592) and cannot be built in this image
(libudev.h is absent), so the expected values below are derived by hand from the
reference source, function by function; the struct layout and the behaviour on
this container's real /sys are those recorded from the reference itself when the
survey ran it here (496-byte record, 7 candidate drivers, 6 virtio PCI devices)."""
import ctypes
import os
import subprocess

import pytest


class MCDeviceInfo(ctypes.Structure):
    # reference heads/libmontecarlo.h:14-20, bound the way desktop/ui.py binds it
    _fields_ = [("syspath", ctypes.c_char * 256), ("vidpid", ctypes.c_char * 32), ("product", ctypes.c_char * 128),
                ("driver", ctypes.c_char * 64), ("subsystem", ctypes.c_char * 16)]


def _declare(lib):
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


@pytest.fixture(scope="module")
def lib():
    """The test build: the same sources with -DMCX_TEST_SEAMS, the only build in
    which MCX_SYSFS_ROOT moves the library's /sys and /proc paths (fixture `tree`)."""
    from montecarlo_amd import binding

    path = os.path.join(os.path.dirname(binding.library_path()), "libmontecarlo_testseams.so")
    return _declare(ctypes.CDLL(path))


@pytest.fixture(scope="module")
def product_lib():
    from montecarlo_amd import binding

    return _declare(binding.load_library())


@pytest.fixture()
def stub_tools(tmp_path, monkeypatch):
    """modprobe / dmesg stand-ins first in PATH: the library runs the reference's own
    command lines (reference montecarlo/libmontecarlo.c:123,133,147), found through PATH"""
    bindir = tmp_path / "stubbin"
    bindir.mkdir()
    log = tmp_path / "modprobe.log"

    def set_modprobe(exit_code):
        f = bindir / "modprobe"
        f.write_text(f"#!/bin/sh\necho \"$@\" >> {log}\nexit {exit_code}\n")
        f.chmod(0o755)

    d = bindir / "dmesg"
    d.write_text("#!/bin/sh\nprintf 'usb 1-1: new device\\ne1000: link up\\n'\n")
    d.chmod(0o755)
    set_modprobe(0)
    monkeypatch.setenv("PATH", f"{bindir}:{os.environ.get('PATH', '')}")
    return set_modprobe, log


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


def test_load_unload_and_dmesg(lib, product_lib, stub_tools, capfd):
    set_modprobe, log = stub_tools
    for L in (lib, product_lib):
        set_modprobe(0)
        assert L.mc_try_load_driver(b"e1000") == 1
        assert L.mc_unload_driver(b"e1000") == 1
        set_modprobe(1)
        assert L.mc_try_load_driver(b"e1000") == 0
        assert L.mc_unload_driver(b"e1000") == -1
        # a name a shell would act on never reaches one
        set_modprobe(0)
        assert L.mc_try_load_driver(b"e1000; touch /tmp/pwned") == 0
        assert L.mc_unload_driver(b"$(reboot)") == -1
        assert L.mc_dmesg_has_activity(b"e1000") == 1
        assert L.mc_dmesg_has_activity(b"iwlwifi") == 0
    # the command lines are the reference's: "modprobe <name>" and "modprobe -r <name>"
    assert log.read_text().split("\n")[:4] == ["e1000", "-r e1000", "e1000", "-r e1000"]
    assert "pwned" not in log.read_text() and "reboot" not in log.read_text()


def test_shipped_library_reads_no_environment(lib, product_lib, tree, tmp_path, monkeypatch):
    """MCX_SYSFS_ROOT moves the test build only; the old command seams are gone from both."""
    assert lib.mc_module_has_holders(b"snd") == 1          # the fake tree, through the seam
    if not os.path.isdir("/sys/module/snd/holders"):
        assert product_lib.mc_module_has_holders(b"snd") == 0   # the real /sys, whatever the variable says
        assert product_lib.mc_get_module_refcount(b"snd") == -1
    marker = tmp_path / "ran"
    monkeypatch.setenv("MCX_MODPROBE", f"touch {marker};")
    monkeypatch.setenv("MCX_DMESG_CMD", f"touch {marker}")
    for L in (lib, product_lib):
        L.mc_try_load_driver(b"no_such_module_xyz")
        L.mc_dmesg_has_activity(b"no_such_module_xyz")
    assert not marker.exists()
    with open(os.path.join(os.path.dirname(__file__), "..", "montecarlo_amd", "lib", "libmontecarlo.so"), "rb") as f:
        blob = f.read()
    assert b"MCX_SYSFS_ROOT" not in blob and b"MCX_MODPROBE" not in blob and b"MCX_DMESG_CMD" not in blob
    # no environment variable at all: the library does not import getenv (or its secure twin)
    nm = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(os.path.dirname(__file__), "..", "montecarlo_amd",
                                                                       "lib", "libmontecarlo.so")],
                        capture_output=True, text=True, check=True).stdout
    assert "getenv" not in nm, [line for line in nm.splitlines() if "getenv" in line]


def test_real_sysfs_matches_reference_probe(product_lib, monkeypatch):
    lib = product_lib
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
