"""The library's name tables as data: the shipped file is the compiled-in set,
a replaced set changes every consumer, a reset brings the defaults back."""
import ctypes
import os

import numpy as np
import pytest

import montecarlo_amd as m
from montecarlo_amd import McxError, safe_module_names

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(HERE), "montecarlo_amd", "data", "montecarlo_tables.txt")


@pytest.fixture(autouse=True)
def defaults_afterwards():
    yield
    m.tables_reset()


@pytest.fixture()
def seams():
    """The test build of the library (the only one whose scalar calls look below
    MCX_SYSFS_ROOT).  It is a second copy of the code with table state of its own:
    its tables are replaced through its own mcx_tables_load_text."""
    from montecarlo_amd import binding

    lib = ctypes.CDLL(os.path.join(os.path.dirname(binding.library_path()), "libmontecarlo_testseams.so"))
    lib.mcx_tables_load_text.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    lib.mcx_tables_reset.restype = None
    lib.mc_is_infrastructure_device.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.mc_driver_is_in_use.argtypes = [ctypes.c_char_p]
    yield lib
    lib.mcx_tables_reset()


def test_shipped_file_is_the_default_set(oracle_mod):
    assert m.tables_info() == dict(format=1, is_default=True, core=113, prefix=56, hw_alias_prefix=6, infra_driver=8,
                                   name_variant=2)
    assert open(DATA).read() == m.tables_dump()
    # the oracle restates the reference's lists literally: same verdict on every name of the
    # table, and on every name with one more character
    text = open(DATA).read()
    names = [ln for ln in text[text.index("[core]"):text.index("[hw_alias_prefix]")].splitlines()
             if ln and not ln.startswith("[")]
    names += [n + "x" for n in names]
    got = safe_module_names(names)
    assert got.tolist() == [oracle_mod.safe_name(n) for n in names]
    m.tables_load(path=DATA)
    assert not m.tables_info()["is_default"] and m.tables_dump() == text
    assert safe_module_names(names).tolist() == got.tolist()


def test_replaced_set_changes_the_consumers(tree, seams):
    assert safe_module_names(["zz_top", "ext4", "e1000"]).tolist() == [1, 0, 1]
    text = """format 1
[core]
e1000          # one name only
[prefix]
zz
[hw_alias_prefix]
acpi
[infra_driver]
ahci
[name_variant]
snd_ e1
"""
    m.tables_load(text=text)
    assert seams.mcx_tables_load_text(text.encode(), len(text.encode())) == 0
    assert m.tables_info() == dict(format=1, is_default=False, core=1, prefix=1, hw_alias_prefix=1, infra_driver=1,
                                   name_variant=1)
    assert safe_module_names(["zz_top", "ext4", "e1000"]).tolist() == [0, 1, 0]
    lib = seams
    dev = lambda p: os.path.join(tree, "sys", "devices", p).encode()   # noqa: E731
    assert lib.mc_is_infrastructure_device(dev("pci0000:00/0000:00:1f.2"), b"pci") == 1    # ahci is infra now
    assert lib.mc_is_infrastructure_device(dev("pci0000:00/0000:04:00.0"), b"pci") == 0    # pcie_pme no longer
    m.tables_reset()
    seams.mcx_tables_reset()
    assert lib.mc_is_infrastructure_device(dev("pci0000:00/0000:00:1f.2"), b"pci") == 0
    assert lib.mc_is_infrastructure_device(dev("pci0000:00/0000:04:00.0"), b"pci") == 1
    assert safe_module_names(["zz_top", "ext4", "e1000"]).tolist() == [1, 0, 1]


def test_name_variant_rules(tree, seams):
    lib = seams
    # the tree binds a device to the usb driver rtw_8821cu: reached through the default rtw88_ rule
    assert lib.mc_driver_is_in_use(b"rtw88_8821cu") == 1
    assert lib.mc_driver_is_in_use(b"rtl8821cu") == 1
    assert lib.mc_driver_is_in_use(b"foo_8821cu") == 0
    text = b"format 1\n[name_variant]\nfoo_ rtw_\n"
    assert seams.mcx_tables_load_text(text, len(text)) == 0
    assert lib.mc_driver_is_in_use(b"foo_8821cu") == 1
    assert lib.mc_driver_is_in_use(b"rtw88_8821cu") == 0


@pytest.mark.parametrize("text", ["", "[core]\nx\n", "format 2\n[core]\nx\n", "format 1\n[nonsense]\nx\n",
                                  "format 1\n[core]\ntwo words\n", "format 1\n[name_variant]\noneword\n"])
def test_malformed_text_is_refused_and_changes_nothing(text):
    with pytest.raises(McxError):
        m.tables_load(text=text)
    assert m.tables_info()["is_default"]
    with pytest.raises(McxError):
        m.tables_load(path="/nonexistent/tables.txt")
