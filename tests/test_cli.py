"""montecarlo_cli: the reference tool's three subcommands keep their output and
exit codes (reference montecarlo/montecarlo.c:7-55); match / roots go through the
batch path and need the GPU."""
import json
import os
import subprocess

import pytest

from test_modtab_cpu import ALIAS_TEXT, DEP_TEXT, PROC_MODULES

HERE = os.path.dirname(os.path.abspath(__file__))
CLI = os.path.join(os.path.dirname(HERE), "montecarlo_amd", "lib", "montecarlo_cli")
# the same tool linked to the test build of the library (the only one that follows MCX_SYSFS_ROOT)
CLI_SEAMS = CLI + "_testseams"


def run(*args, env=None, cli=CLI):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([cli, *args], capture_output=True, text=True, env=e, timeout=120)


def stub_path(tmp_path, exit_code):
    """a directory with a modprobe that exits with exit_code, to go first in PATH"""
    d = tmp_path / f"bin{exit_code}"
    d.mkdir(exist_ok=True)
    f = d / "modprobe"
    f.write_text(f"#!/bin/sh\nexit {exit_code}\n")
    f.chmod(0o755)
    return f"{d}:{os.environ.get('PATH', '')}"


def test_usage_and_unknown_command():
    r = run()
    assert r.returncode == 1 and r.stdout == ""
    assert r.stderr == f"Uso: {CLI} [list|load <driver>|unload <driver>]\n"
    r = run("frobnicate")
    assert r.returncode == 1
    assert r.stderr == f"Comando desconocido: frobnicate\nUso: {CLI} [list|load <driver>|unload <driver>]\n"
    for sub in ("load", "unload"):
        r = run(sub)
        assert r.returncode == 1 and r.stderr == f"Uso: {CLI} {sub} <driver>\n"


def test_list_is_json_of_the_driver_directories(tree):
    r = run("list", env={"MCX_SYSFS_ROOT": tree}, cli=CLI_SEAMS)
    assert r.returncode == 0
    names = json.loads(r.stdout)
    assert sorted(names) == sorted(["usb", "hub", "usbhid", "usb-storage", "dummy_hcd", "rtw_8821cu", "hid-generic",
                                    "pcieport", "ahci", "pcie_pme", "e1000", "dummy", "sd"])
    lines = r.stdout.splitlines()
    assert lines[0] == "[" and lines[-1] == "]" and all(ln.startswith('  "') for ln in lines[1:-1])
    assert all(ln.endswith(",") for ln in lines[1:-2]) and not lines[-2].endswith(",")


def test_load_unload_exit_codes(tmp_path):
    ok = {"PATH": stub_path(tmp_path, 0)}
    bad = {"PATH": stub_path(tmp_path, 1)}
    assert run("load", "e1000", env=ok).returncode == 0
    assert run("load", "e1000", env=bad).returncode == 1
    assert run("unload", "e1000", env=ok).returncode == 0
    r = run("unload", "e1000", env=bad)          # the reference ignores the result and prints the library's message
    assert r.returncode == 0 and "Err en la descarga de drivers." in r.stdout


@pytest.mark.gpu
def test_match_and_roots(tmp_path):
    (tmp_path / "modules.alias").write_text(ALIAS_TEXT)
    (tmp_path / "modules.dep").write_text(DEP_TEXT)
    (tmp_path / "proc_modules").write_text(PROC_MODULES)
    r = run("match", "-d", str(tmp_path), "pci:v00008086d0000100Esv00001AF4sd00001100bc02sc00i00",
            "usb:v0BDAp8153d3000dc00dsc00dp00icFFiscFFipFFin00", "acpi:PNP0C02:", "hid:b0003g0001v0000046Dp0000C52B")
    out = r.stdout.splitlines()
    assert r.returncode == 1, r.stderr                  # one of them has no module
    assert out[0].startswith("pci:v00008086d0000100Esv00001AF4sd00001100bc02sc00i00: e1000 (3 matching aliases")
    assert out[1].startswith("usb:v0BDAp8153d3000dc00dsc00dp00icFFiscFFipFFin00: r8152 (2 matching aliases")
    assert out[2].startswith("acpi:PNP0C02:: button (1 matching alias,")
    assert out[3] == "hid:b0003g0001v0000046Dp0000C52B: no module"
    r = run("match", "-d", str(tmp_path / "nowhere"), "x")
    assert r.returncode == 2 and "modules.alias" in r.stderr
    # loaded: e1000e ptp pps_core rtw88_8821cu; e1000e holds ptp holds pps_core; only modules with a
    # hardware alias and a safe name that nothing holds are listed
    r = run("roots", "-d", str(tmp_path), "-p", str(tmp_path / "proc_modules"))
    assert r.returncode == 0, r.stderr
    assert [ln.split()[0] for ln in r.stdout.splitlines()] == ["e1000e", "rtw88_8821cu"]
    assert all(ln.endswith(" (Idle)") or ln.endswith(" (In Use)") for ln in r.stdout.splitlines())
    assert "1 loaded modules are not in the tables" in r.stderr


@pytest.mark.gpu
def test_roots_status_tag(tmp_path, tree):
    """the status tag of a listed module, on the fake sysfs tree (test build of the tool): a usb
    interface there is bound to `rtw_8821cu`, the name variant of module rtw88_8821cu (reference
    montecarlo/libmontecarlo.c:552-559), nothing is bound to e1000e"""
    (tmp_path / "modules.alias").write_text(ALIAS_TEXT)
    (tmp_path / "modules.dep").write_text(DEP_TEXT)
    (tmp_path / "proc_modules").write_text(PROC_MODULES)
    r = run("roots", "-d", str(tmp_path), "-p", str(tmp_path / "proc_modules"), cli=CLI_SEAMS)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["e1000e (Idle)", "rtw88_8821cu (In Use)"]


def test_bench_usage():
    r = run("bench", "-n", "0")
    assert r.returncode == 1 and "bench [-d <dir>]" in r.stderr
    r = run("bench", "extra-operand")
    assert r.returncode == 1


@pytest.mark.gpu
def test_bench_subcommand(tmp_path):
    """every string is made from a line of modules.alias, so every string resolves; the
    JSON line carries the sizes of the tables it read"""
    (tmp_path / "modules.alias").write_text(ALIAS_TEXT)
    (tmp_path / "modules.dep").write_text(DEP_TEXT)
    r = run("bench", "-d", str(tmp_path), "-n", "5000", "-r", "3")
    assert r.returncode == 0, r.stderr
    rec = json.loads(r.stdout)
    n_alias = sum(1 for ln in ALIAS_TEXT.splitlines() if ln.startswith("alias "))
    assert rec["patterns"] == n_alias and rec["strings"] == 5000 and rec["matched"] == 5000
    assert rec["alias_match_ms"] > 0 and rec["dep_flags_ms"] > 0 and rec["modules"] > 0
    r = run("bench", "-d", str(tmp_path / "nowhere"))
    assert r.returncode == 2
