"""Text-table loaders (modules.alias / modules.dep / /proc/modules): host code of the
library, checked against a plain Python reading of the same text."""
import json
import os

import numpy as np
import pytest

from montecarlo_amd import McxError, ModTab

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ALIAS_TEXT = """# Aliases extracted from modules themselves.
alias pci:v00008086d0000100Esv*sd*bc*sc*i* e1000
alias pci:v00008086d*sv*sd*bc02sc00i* e1000e
alias usb:v0BDAp8153d*dc*dsc*dp*ic*isc*ip*in* r8152

alias usb:v0BDAp*d*dc*dsc*dp*icFFiscFFipFFin* rtw88-8821cu
options e1000 debug=1
alias   of:N*T*Cacme,uart   acme_uart
alias acpi*:PNP0C0[12]:* button
alias pci:v00008086d0000100Esv*sd*bc*sc*i* e1000
"""

DEP_TEXT = """kernel/drivers/net/ethernet/intel/e1000/e1000.ko.xz:
kernel/drivers/net/ethernet/intel/e1000e/e1000e.ko.zst: kernel/drivers/ptp/ptp.ko.zst kernel/drivers/pps/pps_core.ko.zst
kernel/drivers/ptp/ptp.ko.zst: kernel/drivers/pps/pps_core.ko.zst
kernel/drivers/pps/pps_core.ko.zst:
kernel/drivers/net/usb/r8152.ko: kernel/drivers/net/mii.ko kernel/net/core/mdio-devres.ko
kernel/drivers/net/mii.ko:
kernel/net/core/mdio-devres.ko:
kernel/drivers/net/wireless/realtek/rtw88/rtw88-8821cu.ko.gz: kernel/x/rtw88-usb.ko.gz kernel/x/rtw88_core.ko.gz kernel/net/mac80211.ko kernel/net/cfg80211.ko
kernel/x/rtw88-usb.ko.gz: kernel/x/rtw88_core.ko.gz kernel/net/mac80211.ko kernel/net/cfg80211.ko
kernel/x/rtw88_core.ko.gz: kernel/net/mac80211.ko kernel/net/cfg80211.ko
kernel/net/mac80211.ko: kernel/net/cfg80211.ko
kernel/net/cfg80211.ko:
"""

PROC_MODULES = """e1000e 286720 0 - Live 0xffffffffc0a00000
ptp 32768 1 e1000e, Live 0xffffffffc09f0000
pps_core 24576 1 ptp, Live 0xffffffffc09e0000
nvidia 56000000 12 - Live 0xffffffffc1000000 (POE)
rtw88_8821cu 16384 0 - Live 0xffffffffc0900000
"""


def norm(path_or_name):
    return os.path.basename(path_or_name).split(".")[0].replace("-", "_")


def test_alias_text():
    with ModTab() as t:
        t.add_alias_text(ALIAS_TEXT)
        arena, a_off, a_mod = t.alias_arrays()
        want = [ln.split() for ln in ALIAS_TEXT.splitlines() if ln.split() and ln.split()[0] == "alias"]
        assert t.counts()[0] == len(want) == 7
        raw = arena.tobytes()
        assert [raw[a_off[j]:a_off[j + 1]].decode() for j in range(len(want))] == [w[1] for w in want]
        assert [t.module_name(m) for m in a_mod] == [norm(w[2]) for w in want]
        # ids in order of first appearance, names normalised, repeats share an id
        assert t.module_names() == ["e1000", "e1000e", "r8152", "rtw88_8821cu", "acme_uart", "button"]
        assert t.module_id("rtw88-8821cu") == 3 and t.module_id("rtw88_8821cu.ko.xz") == 3
        assert t.module_id("nope") == 0xFFFFFFFF and t.module_name(99) is None
        with pytest.raises(McxError):
            t.add_alias_text("alias onlypattern\n")


def test_dep_text_direct_and_flat():
    with ModTab() as t:
        t.add_alias_text(ALIAS_TEXT)
        t.add_dep_text(DEP_TEXT)
        names = t.module_names()
        ids = {n: i for i, n in enumerate(names)}
        flat = {}
        for ln in DEP_TEXT.splitlines():
            head, _, rest = ln.partition(":")
            flat[norm(head)] = [norm(x) for x in rest.split()]
        assert set(flat) <= set(names)
        rowptr, col = t.dep_arrays(direct=False)
        for n, deps in flat.items():
            u = ids[n]
            assert sorted(col[rowptr[u]:rowptr[u + 1]].tolist()) == sorted(ids[d] for d in deps)
        # direct edges: what the holders directories would show
        rowptr, col = t.dep_arrays(direct=True)
        direct = {n: sorted(names[v] for v in col[rowptr[ids[n]]:rowptr[ids[n] + 1]]) for n in names}
        assert direct["e1000e"] == ["ptp"]
        assert direct["ptp"] == ["pps_core"]
        assert direct["r8152"] == ["mdio_devres", "mii"]
        assert direct["rtw88_8821cu"] == ["rtw88_usb"]
        assert direct["rtw88_usb"] == ["rtw88_core"]
        assert direct["rtw88_core"] == ["mac80211"]
        assert direct["mac80211"] == ["cfg80211"]
        assert direct["e1000"] == [] and direct["button"] == [] and direct["acme_uart"] == []
        # transitive closure of the direct edges gives the file's lists back
        for n, deps in flat.items():
            seen, work = set(), [n]
            while work:
                for d in direct[work.pop()]:
                    if d not in seen:
                        seen.add(d)
                        work.append(d)
            assert seen == set(deps), n
        with pytest.raises(McxError):
            t.add_dep_text("kernel/x/y.ko kernel/x/z.ko\n")


def test_proc_modules():
    with ModTab() as t:
        t.add_alias_text(ALIAS_TEXT)
        t.add_dep_text(DEP_TEXT)
        flags, unknown = t.mark_loaded(PROC_MODULES)
        loaded = {t.module_name(i) for i in np.nonzero(flags & 1)[0]}
        assert loaded == {"e1000e", "ptp", "pps_core", "rtw88_8821cu"}
        assert unknown == 1  # nvidia: not in this kernel's tables


def test_kmod_fixture_as_text_roundtrip():
    fx = json.load(open(os.path.join(GOLDEN, "kmod_resolve.json")))
    text = "".join(f"alias {p} {m}\n" for p, m in fx["aliases"])
    with ModTab() as t:
        t.add_alias_text(text)
        arena, a_off, a_mod = t.alias_arrays()
        assert t.counts()[0] == len(fx["aliases"])
        raw = arena.tobytes()
        for j in (0, 17, len(fx["aliases"]) - 1):
            assert raw[a_off[j]:a_off[j + 1]].decode() == fx["aliases"][j][0]
            assert t.module_name(a_mod[j]) == fx["aliases"][j][1].replace("-", "_")


def test_load_dir(tmp_path):
    (tmp_path / "modules.alias").write_text(ALIAS_TEXT)
    (tmp_path / "modules.dep").write_text(DEP_TEXT)
    with ModTab() as t:
        t.load_dir(str(tmp_path))
        assert t.counts()[:2] == (7, 14)
    with ModTab() as t, pytest.raises(McxError):
        t.load_dir(str(tmp_path / "missing"))


def test_dep_text_fuzz_direct_edges():
    """random dependency DAGs without redundant edges, written the way depmod writes them (every
    line lists all that the module needs, directly or not): the table's direct edges are the DAG's,
    its flat lists are the file's"""
    rng = np.random.default_rng(3)
    for trial in range(30):
        V = int(rng.integers(2, 60))
        need = [set() for _ in range(V)]          # direct, u needs v only for v < u
        reach = [set() for _ in range(V)]
        for u in range(1, V):
            for v in rng.permutation(u)[: int(rng.integers(0, 4))]:
                v = int(v)
                if v in reach[u]:
                    continue                      # would be a redundant edge
                # adding u -> v must not make an earlier direct edge of u redundant
                if any(w in reach[v] for w in need[u]):
                    continue
                need[u].add(v)
                reach[u] |= {v} | reach[v]
        names = [f"mod-{k}" if k % 3 else f"m{k}" for k in range(V)]
        lines = []
        for u in rng.permutation(V):
            deps = sorted(reach[u], reverse=True)
            lines.append(f"kernel/x/{names[u]}.ko.zst:" + "".join(f" kernel/y/{names[d]}.ko.zst" for d in deps))
        with ModTab() as t:
            t.add_dep_text("\n".join(lines) + "\n")
            ids = {n: t.module_id(n) for n in names}
            assert sorted(ids.values()) == list(range(V))
            back = {i: k for k, i in ((k, ids[names[k]]) for k in range(V))}
            rowptr, col = t.dep_arrays(direct=True)
            frow, fcol = t.dep_arrays(direct=False)
            for k in range(V):
                i = ids[names[k]]
                assert {back[int(x)] for x in col[rowptr[i]:rowptr[i + 1]]} == need[k], (trial, k)
                assert {back[int(x)] for x in fcol[frow[i]:frow[i + 1]]} == reach[k], (trial, k)
