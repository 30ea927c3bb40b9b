import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    """One context on device 0.  GPU tests must run the HIP path: a missing
    library or device is an error, never a skip or a CPU substitute."""
    from montecarlo_amd import Context

    ctx = Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(autouse=True)
def _default_knobs(request):
    """a test that turned a tuning knob of the shared context does not leave it turned"""
    yield
    if "gpu_ctx" in request.fixturenames:
        request.getfixturevalue("gpu_ctx").tune(compile_threads=0, sub_min=0, tiles_min="auto", list_two_pass=0)


# ---- a fake sysfs / procfs tree for the scalar ABI (env MCX_SYSFS_ROOT) ------------

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
