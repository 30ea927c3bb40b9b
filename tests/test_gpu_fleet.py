"""GPU parity of the fleet pipeline (mcx_fleet_roots through the C ABI): alias
match -> used drivers -> loaded set -> holders -> root filter on the device,
against the CPU oracle's per-host loop on the same seeded fleets.  Bit-exact."""
import numpy as np
import pytest

from montecarlo_amd import McxError, safe_module_names, synth
from montecarlo_amd.binding import pack_strings

from fleet_cases import HAND_LIMIT, HAND_LISTED, HAND_WANT, hand_case, inuse_lists, labeled_expectation, variant

pytestmark = pytest.mark.gpu
NO_MATCH = 0xFFFFFFFF


def make_fleet(seed, P, H, mean_dev, max_extra, mean_deg=4, V=None):
    A, a_off, a_mod = synth.alias_table(seed, P)
    V = synth.alias_modules(P) if V is None else V  # more modules than the aliases name: extras reach them
    rowptr, col = synth.dep_graph(seed + 1, V, mean_deg)
    dev_cnt, extra_off, extra_ids = synth.fleet_hosts(seed + 2, V, H, mean_dev, max_extra)
    dev_off = np.zeros(H + 1, dtype=np.uint32)
    dev_off[1:] = np.cumsum(dev_cnt)
    N = int(dev_off[H])
    S, s_off = synth.modalias_strings(seed + 3, P, N)
    safe = safe_module_names(synth.module_names(seed + 4, V))
    return dict(A=A, a_off=a_off, a_mod=a_mod, V=V, rowptr=rowptr, col=col, dev_off=dev_off,
                extra_off=extra_off, extra_ids=extra_ids, S=S, s_off=s_off, safe=safe, N=N, H=H)


def oracle_fleet(oracle_mod, ctx, f, use_twin=False):
    first, count = oracle_mod.alias_match(f["A"], f["a_off"], f["S"], f["s_off"], use_twin=use_twin)
    if use_twin:
        # the twin's answers are the expectation: 1000 of them spread over the fleet pinned by glibc
        N = len(f["s_off"]) - 1
        picks = sorted(set(np.random.default_rng(N).choice(N, size=min(N, 1000), replace=False).tolist()))
        g_first, g_count = oracle_mod.alias_match_sample(f["A"], f["a_off"], f["S"], f["s_off"], picks)
        assert np.array_equal(first[picks], g_first) and np.array_equal(count[picks], g_count)
    driver = np.where(first != NO_MATCH, f["a_mod"][np.minimum(first, len(f["a_mod"]) - 1)], NO_MATCH)
    # the hardware-alias stage decided on the host from the table's text (the six tags of
    # reference desktop/ui.py:1540), not taken from the library under test
    hw = oracle_mod.hw_alias_flags(f["A"], f["a_off"], f["a_mod"], f["V"])
    return oracle_mod.fleet_roots(f["rowptr"], f["col"], driver.astype(np.uint32), f["dev_off"], f["extra_off"],
                                  f["extra_ids"], f["safe"], hw)


def run_fleet(ctx, f, cap=None):
    ctx.alias_table_load(f["A"], f["a_off"], f["a_mod"])
    ctx.dep_graph_load(f["rowptr"], f["col"])
    return ctx.fleet_roots(f["S"], f["s_off"], f["dev_off"], f["extra_off"], f["extra_ids"], f["safe"], cap=cap)


@pytest.mark.parametrize("seed,P,H,mean_dev,max_extra", [(21, 3000, 257, 12, 4), (22, 500, 33, 3, 2),
                                                         (23, 2000, 1, 40, 60), (24, 800, 1025, 2, 1)])
def test_fleet_vs_oracle(gpu_ctx, oracle_mod, seed, P, H, mean_dev, max_extra):
    f = make_fleet(seed, P, H, mean_dev, max_extra)
    off, ids = run_fleet(gpu_ctx, f)
    w_off, w_ids = oracle_fleet(oracle_mod, gpu_ctx, f)
    assert np.array_equal(off, w_off)
    assert np.array_equal(ids, w_ids)
    assert len(ids) > 0 or H == 1


@pytest.mark.parametrize("V", [8128, 8192, 10000, 16320, 16384, 20000])
def test_fleet_many_modules(gpu_ctx, oracle_mod, V):
    """module counts around the sizes at which the used-driver bits stop being built by word
    column in LDS (one column of V words, two with bound drivers: 64 KB together with the
    kernel's own 256 bytes) and go back to one atomic per device: both modes against the oracle
    on either side of both limits and right at them"""
    f = make_fleet(91, 3000, 300, 10, 6, V=V)
    assert f["V"] == V and int(f["extra_ids"].max()) > synth.alias_modules(3000)
    off, ids = run_fleet(gpu_ctx, f)
    w_off, w_ids = oracle_fleet(oracle_mod, gpu_ctx, f)
    assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids) and len(ids) > 0
    first, _ = oracle_mod.alias_match(f["A"], f["a_off"], f["S"], f["s_off"])
    resolved = np.where(first != NO_MATCH, f["a_mod"][np.minimum(first, len(f["a_mod"]) - 1)], NO_MATCH).astype(np.uint32)
    bound = resolved.copy()
    bound[::3] = NO_MATCH  # every third device has no driver bound
    hw = oracle_mod.hw_alias_flags(f["A"], f["a_off"], f["a_mod"], f["V"])
    b_off, b_ids = oracle_mod.fleet_roots(f["rowptr"], f["col"], resolved, f["dev_off"], f["extra_off"],
                                          f["extra_ids"], f["safe"], hw, bound_mod=bound)
    off, ids = gpu_ctx.fleet_roots_bound(f["S"], f["s_off"], f["dev_off"], f["extra_off"], f["extra_ids"], f["safe"],
                                         bound)
    assert np.array_equal(off, b_off) and np.array_equal(ids, b_ids) and len(ids) > len(w_ids)


def test_fleet_edge_cases(gpu_ctx, oracle_mod):
    f = make_fleet(31, 1000, 40, 5, 3)
    gpu_ctx.alias_table_load(f["A"], f["a_off"], f["a_mod"])
    gpu_ctx.dep_graph_load(f["rowptr"], f["col"])
    H = f["H"]
    # no hosts
    off, ids = gpu_ctx.fleet_roots(np.zeros(0, np.uint8), np.zeros(1, np.uint32), np.zeros(1, np.uint32),
                                   np.zeros(1, np.uint32), np.zeros(0, np.uint32), f["safe"])
    assert off.tolist() == [0] and len(ids) == 0
    # hosts without devices: extras only
    zero_dev = np.zeros(H + 1, dtype=np.uint32)
    off, ids = gpu_ctx.fleet_roots(np.zeros(0, np.uint8), np.zeros(1, np.uint32), zero_dev, f["extra_off"],
                                   f["extra_ids"], f["safe"])
    hw = gpu_ctx.alias_hw_flags(f["V"])
    w_off, w_ids = oracle_mod.fleet_roots(f["rowptr"], f["col"], np.zeros(0, np.uint32), zero_dev, f["extra_off"],
                                          f["extra_ids"], f["safe"], hw)
    assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids)
    # hosts without extras: every loaded module is a used driver or held by one -> nothing is listed
    zero_x = np.zeros(H + 1, dtype=np.uint32)
    off, ids = gpu_ctx.fleet_roots(f["S"], f["s_off"], f["dev_off"], zero_x, np.zeros(0, np.uint32), f["safe"])
    assert len(ids) == 0 and not off.any()
    # an output list that is too small: error, total and offsets still reported
    w_off, w_ids = oracle_fleet(oracle_mod, gpu_ctx, f)
    assert len(w_ids) > 1
    with pytest.raises(McxError) as e:
        gpu_ctx.fleet_roots(f["S"], f["s_off"], f["dev_off"], f["extra_off"], f["extra_ids"], f["safe"], cap=1)
    assert e.value.rc == -6
    off, ids = gpu_ctx.fleet_roots(f["S"], f["s_off"], f["dev_off"], f["extra_off"], f["extra_ids"], f["safe"],
                                   cap=len(w_ids))
    assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids)
    # a graph with a cycle has no load order
    gpu_ctx.dep_graph_load(np.array([0, 1, 2], np.uint32), np.array([1, 0], np.uint32))
    with pytest.raises(McxError) as e:
        gpu_ctx.fleet_roots(np.zeros(0, np.uint8), np.zeros(1, np.uint32), np.zeros(2, np.uint32),
                            np.zeros(2, np.uint32), np.zeros(0, np.uint32), np.ones(2, np.uint8))
    assert e.value.rc == -2


def test_fleet_hand_case(gpu_ctx, oracle_mod):
    """the hand-made fleet of the CPU suite, with real patterns in front of it"""
    pats = [b"pci:v00008086d00001234*", b"usb:v1D6Bp0002*", b"pci:v00008086d*", b"platform:foo", b"of:Nbar*",
            b"input:b0003*"]
    a_mod = np.array([1, 4, 0, 3, 2, 5], dtype=np.uint32)   # module 5 has no hardware-bus alias
    A, a_off = pack_strings(pats)
    rowptr = np.array([0, 1, 2, 2, 3, 3, 4], dtype=np.uint32)
    col = np.array([1, 2, 2, 4], dtype=np.uint32)
    names = ["e1000", "r8169", "igb", "ahci", "tg3", "snd_x"]
    safe = safe_module_names(names)
    assert safe.tolist() == [1, 1, 1, 1, 1, 0]
    devs = [b"pci:v00008086d00001234sv0sd0bc02sc00i00", b"usb:v1D6Bp0002d0001", b"acpi:PNP0C0A:"]
    S, s_off = pack_strings(devs)
    dev_off = np.array([0, 1, 1, 3, 3, 3], dtype=np.uint32)
    extra_off = np.array([0, 1, 3, 3, 3, 5], dtype=np.uint32)
    extra_ids = np.array([3, 0, 5, 4, 9], dtype=np.uint32)
    gpu_ctx.alias_table_load(A, a_off, a_mod)
    gpu_ctx.dep_graph_load(rowptr, col)
    assert gpu_ctx.alias_hw_flags(6).tolist() == [1, 1, 1, 1, 1, 0]
    off, ids = gpu_ctx.fleet_roots(S, s_off, dev_off, extra_off, extra_ids, safe)
    # host 0: driver 1 (first match wins over the wider pattern of module 0), extra 3 -> 3 listed
    # host 1: extras 0 and 5 -> 0 listed (5: unsafe name and no hardware alias; 4, 1, 2 held)
    # host 2: driver 4; host 3: nothing; host 4: extra 4 listed (hardware alias, safe, nobody holds it)
    assert off.tolist() == [0, 1, 2, 2, 2, 3]
    assert ids.tolist() == [3, 0, 4]


def test_fleet_medium_vs_oracle(gpu_ctx, oracle_mod):
    """20000 hosts x ~20 devices against the 30000-pattern table; alias part of the
    expectation by the CPU twin (itself pinned to glibc fnmatch in the CPU suite)"""
    f = make_fleet(41, 30000, 20000, 20, 6)
    off, ids = run_fleet(gpu_ctx, f)
    w_off, w_ids = oracle_fleet(oracle_mod, gpu_ctx, f, use_twin=True)
    assert np.array_equal(off, w_off)
    assert np.array_equal(ids, w_ids)


def test_fleet_full_size_tiled(gpu_ctx, oracle_mod):
    """full-size fleet of the benchmark definition (100000 hosts, ~100 devices each, 30000 patterns): the
    fleet is 100 copies of 1000 distinct hosts, so the expected lists are the
    oracle's lists of those 1000 hosts, repeated"""
    base = make_fleet(51, 30000, 1000, 100, 8)
    gpu_ctx.alias_table_load(base["A"], base["a_off"], base["a_mod"])   # hw flags come from this table
    w_off, w_ids = oracle_fleet(oracle_mod, gpu_ctx, base, use_twin=True)
    reps = 100
    Hb, Nb, Xb = base["H"], base["N"], int(base["extra_off"][-1])
    sb = int(base["s_off"][-1])
    S = np.tile(base["S"][:sb], reps)
    s_off = np.concatenate([[0], (base["s_off"][1:].astype(np.uint64)[None, :] +
                                  (np.arange(reps, dtype=np.uint64) * sb)[:, None]).ravel()]).astype(np.uint32)
    dev_off = np.concatenate([[0], (base["dev_off"][1:].astype(np.uint64)[None, :] +
                                    (np.arange(reps, dtype=np.uint64) * Nb)[:, None]).ravel()]).astype(np.uint32)
    extra_off = np.concatenate([[0], (base["extra_off"][1:].astype(np.uint64)[None, :] +
                                      (np.arange(reps, dtype=np.uint64) * Xb)[:, None]).ravel()]).astype(np.uint32)
    extra_ids = np.tile(base["extra_ids"], reps)
    assert int(s_off[-1]) == sb * reps and len(dev_off) == Hb * reps + 1
    gpu_ctx.alias_table_load(base["A"], base["a_off"], base["a_mod"])
    gpu_ctx.dep_graph_load(base["rowptr"], base["col"])
    off, ids = gpu_ctx.fleet_roots(S, s_off, dev_off, extra_off, extra_ids, base["safe"])
    T = len(w_ids)
    assert len(ids) == T * reps
    assert np.array_equal(ids.reshape(reps, T), np.tile(w_ids, (reps, 1)))
    want_off = np.concatenate([[0], (w_off[1:].astype(np.uint64)[None, :] +
                                     (np.arange(reps, dtype=np.uint64) * T)[:, None]).ravel()]).astype(np.uint32)
    assert np.array_equal(off, want_off)


@pytest.mark.parametrize("cap_per_host", [0, 5])
def test_inventory_to_lists_in_one_call(gpu_ctx, oracle_mod, cap_per_host):
    """device records -> device list -> alias match -> module lists, one call, against the
    oracle's device filter followed by the oracle's per-host loop over the listed devices"""
    f = make_fleet(61, 3000, 300, 14, 4)
    ctx, N, H = gpu_ctx, f["N"], f["H"]
    rec = synth.device_records(61, f["S"], f["s_off"])
    arena, off = synth.driver_names_packed()
    ctx.alias_table_load(f["A"], f["a_off"], f["a_mod"])
    ctx.dep_graph_load(f["rowptr"], f["col"])
    # expectation
    o_v, o_kept = oracle_mod.device_filter(rec, arena, off, f["dev_off"], cap_per_host)
    keep = np.nonzero(o_v & 1)[0]
    raw = f["S"].tobytes()
    Sk, sk_off = pack_strings([raw[f["s_off"][i]:f["s_off"][i + 1]] for i in keep])
    g = dict(f, S=Sk, s_off=sk_off, dev_off=o_kept, N=len(keep))
    w_off, w_ids = oracle_fleet(oracle_mod, ctx, g)
    # one call
    bufs = [ctx.to_device(x) for x in (rec, f["S"], f["s_off"], arena, off, f["dev_off"], f["extra_off"],
                                       f["extra_ids"], f["safe"])]
    d_rec, d_S, d_soff, d_arena, d_doff, d_dev, d_xoff, d_xid, d_safe = bufs
    d_v, d_roff = ctx.alloc(N), ctx.alloc(4 * (H + 1))
    args = (d_rec.ptr, d_S.ptr, d_soff.ptr, N, d_arena.ptr, d_doff.ptr, len(off) - 1, d_dev.ptr, H, cap_per_host,
            d_xoff.ptr, d_xid.ptr, len(f["extra_ids"]), d_safe.ptr, d_v.ptr, d_roff.ptr)
    total = ctx.fleet_inventory_roots_dev(*args, None, 0, size_only=True)
    assert total == len(w_ids) and total > 0
    d_rid = ctx.alloc(4 * total)
    assert ctx.fleet_inventory_roots_dev(*args, d_rid.ptr, total) == total
    assert np.array_equal(d_v.download(np.uint8, N) & 3, o_v)
    assert np.array_equal(d_roff.download(np.uint32, H + 1), w_off)
    assert np.array_equal(d_rid.download(np.uint32, total), w_ids)
    for buf in bufs + [d_v, d_roff, d_rid]:
        buf.free()


def _bound_case(seed, P, H, mean_dev, max_extra):
    """a fleet with drivers bound by name: most devices are bound to the module their modalias
    resolves to; some resolve and stay unbound; some are bound to a driver whose name is the
    module's with dashes ('e1000-x' for module 'e1000_x': no module of that name), some to a
    module other than the one the modalias names"""
    f = make_fleet(seed, P, H, mean_dev, max_extra)
    names = synth.module_names(seed + 4, f["V"])
    first, _ = __import__("oracle").alias_match(f["A"], f["a_off"], f["S"], f["s_off"], use_twin=True)
    resolved = np.where(first != NO_MATCH, f["a_mod"][np.minimum(first, len(f["a_mod"]) - 1)], NO_MATCH)
    rng = np.random.default_rng(seed)
    drivers = []                      # per device: the bound driver's name, None = unbound
    for i in range(f["N"]):
        r, m = rng.random(), int(resolved[i])
        if m == NO_MATCH or r < 0.2:
            drivers.append(None)
        elif r < 0.3:
            drivers.append(names[m].replace("_", "-"))
        elif r < 0.4:
            drivers.append(names[int(rng.integers(0, f["V"]))])
        else:
            drivers.append(names[m])
    return f, names, resolved.astype(np.uint32), drivers


@pytest.mark.parametrize("seed,P,H,mean_dev,max_extra", [(71, 3000, 200, 12, 4), (72, 600, 33, 5, 0)])
def test_fleet_used_drivers_from_bound_names(gpu_ctx, oracle_mod, seed, P, H, mean_dev, max_extra):
    """the reference's notion of a used driver (desktop/ui.py:1358-1359, 1567): the driver bound
    to a device, compared with the module names as it stands"""
    from montecarlo_amd import driver_modules

    f, names, resolved, drivers = _bound_case(seed, P, H, mean_dev, max_extra)
    by_name = {}
    for v, n in enumerate(names):
        by_name.setdefault(n, v)
    want_bound = np.array([by_name.get(d, NO_MATCH) if d is not None else NO_MATCH for d in drivers], dtype=np.uint32)
    # the library's name lookup, on the distinct driver names
    distinct = sorted({d for d in drivers if d is not None})
    drv_mod = driver_modules(distinct, names)
    at = {d: k for k, d in enumerate(distinct)}
    bound = np.array([drv_mod[at[d]] if d is not None else NO_MATCH for d in drivers], dtype=np.uint32)
    assert np.array_equal(bound, want_bound)
    assert any("-" in d for d in distinct) and (bound == NO_MATCH).sum() > (np.array(drivers) == None).sum()  # noqa: E711
    hw = oracle_mod.hw_alias_flags(f["A"], f["a_off"], f["a_mod"], f["V"])
    w_off, w_ids = oracle_mod.fleet_roots(f["rowptr"], f["col"], resolved, f["dev_off"], f["extra_off"],
                                          f["extra_ids"], f["safe"], hw, bound_mod=want_bound)
    gpu_ctx.alias_table_load(f["A"], f["a_off"], f["a_mod"])
    gpu_ctx.dep_graph_load(f["rowptr"], f["col"])
    off, ids = gpu_ctx.fleet_roots_bound(f["S"], f["s_off"], f["dev_off"], f["extra_off"], f["extra_ids"], f["safe"],
                                         bound)
    assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids)
    # the other mode gives other lists: modules that resolved without binding show up here only
    a_off, a_ids = run_fleet(gpu_ctx, f)
    assert len(ids) > len(a_ids)
    # all devices bound to what they resolve to: the two modes agree
    off2, ids2 = gpu_ctx.fleet_roots_bound(f["S"], f["s_off"], f["dev_off"], f["extra_off"], f["extra_ids"],
                                           f["safe"], resolved)
    assert np.array_equal(off2, a_off) and np.array_equal(ids2, a_ids)


def test_inventory_with_bound_drivers(gpu_ctx, oracle_mod):
    """records -> device list -> module lists with the used drivers read from the records' driver
    field through a name -> module map; the driver table mixes the names the device filter
    looks for with module names, dashed variants and names that are no module"""
    from montecarlo_amd import driver_modules

    f = make_fleet(81, 3000, 250, 14, 4)
    ctx, N, H, V = gpu_ctx, f["N"], f["H"], f["V"]
    names = synth.module_names(85, V)
    rec = synth.device_records(81, f["S"], f["s_off"])
    base = [n for n in synth.DRIVER_NAMES]
    table = base + names[:400] + [n.replace("_", "-") for n in names[400:450]] + ["no-such-module", ""]
    arena, off = pack_strings([n.encode() for n in table])
    rng = np.random.default_rng(81)
    pick = rng.random(N) < 0.7
    rec["driver"] = np.where(pick, rng.integers(0, len(table), size=N), rec["driver"]).astype(np.uint32)
    drv_mod = driver_modules(table, names)
    by_name = {}
    for v, n in enumerate(names):
        by_name.setdefault(n, v)
    assert drv_mod.tolist() == [by_name.get(n, NO_MATCH) for n in table]
    ctx.alias_table_load(f["A"], f["a_off"], f["a_mod"])
    ctx.dep_graph_load(f["rowptr"], f["col"])
    # expectation: the oracle's device list, then its per-host loop over the listed devices
    o_v, o_kept = oracle_mod.device_filter(rec, arena, off, f["dev_off"], 0)
    keep = np.nonzero(o_v & 1)[0]
    raw = f["S"].tobytes()
    Sk, sk_off = pack_strings([raw[f["s_off"][i]:f["s_off"][i + 1]] for i in keep])
    first, _ = oracle_mod.alias_match(f["A"], f["a_off"], Sk, sk_off, use_twin=True)
    resolved = np.where(first != NO_MATCH, f["a_mod"][np.minimum(first, len(f["a_mod"]) - 1)], NO_MATCH)
    d = rec["driver"][keep]
    bound = np.where(d < len(table), drv_mod[np.minimum(d, len(table) - 1)], NO_MATCH).astype(np.uint32)
    hw = oracle_mod.hw_alias_flags(f["A"], f["a_off"], f["a_mod"], V)
    w_off, w_ids = oracle_mod.fleet_roots(f["rowptr"], f["col"], resolved.astype(np.uint32), o_kept, f["extra_off"],
                                          f["extra_ids"], f["safe"], hw, bound_mod=bound)
    bufs = [ctx.to_device(x) for x in (rec, f["S"], f["s_off"], arena, off, drv_mod, f["dev_off"], f["extra_off"],
                                       f["extra_ids"], f["safe"])]
    d_rec, d_S, d_soff, d_arena, d_doff, d_dm, d_dev, d_xoff, d_xid, d_safe = bufs
    d_v, d_roff = ctx.alloc(N), ctx.alloc(4 * (H + 1))
    args = (d_rec.ptr, d_S.ptr, d_soff.ptr, N, d_arena.ptr, d_doff.ptr, len(table), d_dm.ptr, d_dev.ptr, H, 0,
            d_xoff.ptr, d_xid.ptr, len(f["extra_ids"]), d_safe.ptr, d_v.ptr, d_roff.ptr)
    total = ctx.fleet_inventory_roots_bound_dev(*args, None, 0, size_only=True)
    assert total == len(w_ids) and total > 0
    d_rid = ctx.alloc(4 * total)
    for _ in range(2):   # twice: the second call reuses the context's buffers
        assert ctx.fleet_inventory_roots_bound_dev(*args, d_rid.ptr, total) == total
        assert np.array_equal(d_v.download(np.uint8, N) & 3, o_v)
        assert np.array_equal(d_roff.download(np.uint32, H + 1), w_off)
        assert np.array_equal(d_rid.download(np.uint32, total), w_ids)
    for buf in bufs + [d_v, d_roff, d_rid]:
        buf.free()


# ---- "(In Use)" / "(Idle)" labels ---------------------------------------------------------

def _run_inventory_labeled(ctx, f, rec, e, table, names, max_per_host):
    from montecarlo_amd import driver_marks
    mark_off, mark_ids = driver_marks(table, names)
    N, H = f["N"], f["H"]
    bufs = [ctx.to_device(x) for x in (rec, f["S"], f["s_off"], e["arena"], e["off"], e["drv_mod"], mark_off,
                                       mark_ids if len(mark_ids) else np.zeros(1, np.uint32), f["dev_off"],
                                       f["extra_off"], f["extra_ids"], f["safe"])]
    d_rec, d_S, d_soff, d_arena, d_doff, d_dm, d_moff, d_mid, d_dev, d_xoff, d_xid, d_safe = bufs
    d_v, d_roff = ctx.alloc(max(N, 1)), ctx.alloc(4 * (H + 1))
    args = (d_rec.ptr, d_S.ptr, d_soff.ptr, N, d_arena.ptr, d_doff.ptr, len(table), d_dm.ptr, d_moff.ptr, d_mid.ptr,
            d_dev.ptr, H, max_per_host, d_xoff.ptr, d_xid.ptr, len(f["extra_ids"]), d_safe.ptr, d_v.ptr, d_roff.ptr)
    total = ctx.fleet_inventory_roots_labeled_dev(*args, None, None, 0, size_only=True)
    d_rid, d_use = ctx.alloc(4 * max(total, 1)), ctx.alloc(max(total, 1))
    assert ctx.fleet_inventory_roots_labeled_dev(*args, d_rid.ptr, d_use.ptr, total) == total
    out = (d_roff.download(np.uint32, H + 1), d_rid.download(np.uint32, total), d_use.download(np.uint8, total),
           d_v.download(np.uint8, N))
    for buf in bufs + [d_v, d_roff, d_rid, d_use]:
        buf.free()
    return out, (mark_off, mark_ids)


def test_in_use_label_hand_case(gpu_ctx, oracle_mod):
    """the hand-worked host of tests/fleet_cases.py::hand_case: the library's labels are the
    documented ones and the oracle's"""
    f, rec, table, names = hand_case()
    gpu_ctx.alias_table_load(f["A"], f["a_off"], f["a_mod"])
    gpu_ctx.dep_graph_load(f["rowptr"], f["col"])
    e = labeled_expectation(oracle_mod, f, rec, table, names, HAND_LIMIT)
    (off, ids, use, verdict), _ = _run_inventory_labeled(gpu_ctx, f, rec, e, table, names, HAND_LIMIT)
    assert np.array_equal(verdict & 3, e["o_v"])
    assert [int(x) for x in (verdict & 1)] == HAND_LISTED
    got = {names[int(m)]: int(u) for m, u in zip(ids, use)}
    assert got == HAND_WANT
    assert np.array_equal(off, e["w_off"]) and np.array_equal(ids, e["w_ids"]) and np.array_equal(use, e["w_use"])


@pytest.mark.parametrize("seed,P,H,mean_dev,max_per_host,V", [(101, 3000, 300, 14, 0, None), (102, 2000, 130, 30, 8, None),
                                                              (103, 3000, 200, 10, 0, 20000)])
def test_in_use_labels_of_a_seeded_fleet(gpu_ctx, oracle_mod, seed, P, H, mean_dev, max_per_host, V):
    """records -> device list -> module lists -> labels in one call, and the same through the
    entry point that takes per-host lists of marked modules, both against the oracle's restatement
    of mc_driver_is_in_use over the unfiltered records.  Drivers are bound under the module's own
    name, its rtw88 / rtl variant, a dashed spelling, another module's name, or not at all; the
    device filter hides about a fifth of the devices, whose bindings still count."""
    from montecarlo_amd import driver_marks

    f = make_fleet(seed, P, H, mean_dev, 4, V=V)
    ctx, N, Vn = gpu_ctx, f["N"], f["V"]
    names = synth.module_names(seed + 4, Vn)
    rec = synth.device_records(seed, f["S"], f["s_off"])
    variants = sorted({variant(n) for n in names if variant(n)})
    table = list(synth.DRIVER_NAMES) + names[:600] + variants[:200] + [n.replace("_", "-") for n in names[600:650]]
    at = {n: k for k, n in enumerate(table)}
    first, _ = oracle_mod.alias_match(f["A"], f["a_off"], f["S"], f["s_off"], use_twin=True)
    resolved = np.where(first != NO_MATCH, f["a_mod"][np.minimum(first, len(f["a_mod"]) - 1)], NO_MATCH)
    rng = np.random.default_rng(seed)
    drv = rec["driver"].copy()
    for i in range(N):
        m, r = int(resolved[i]), rng.random()
        if m == NO_MATCH or r < 0.25:
            continue                                   # what the generator drew (infra names, none, ...)
        name = names[m]
        if r < 0.55 and name in at:
            drv[i] = at[name]
        elif r < 0.75 and variant(name) in at:
            drv[i] = at[variant(name)]
        elif r < 0.85:
            drv[i] = rng.integers(len(synth.DRIVER_NAMES), len(table))
    rec["driver"] = drv
    ctx.alias_table_load(f["A"], f["a_off"], f["a_mod"])
    ctx.dep_graph_load(f["rowptr"], f["col"])
    e = labeled_expectation(oracle_mod, f, rec, table, names, max_per_host)
    (off, ids, use, verdict), (mark_off, mark_ids) = _run_inventory_labeled(ctx, f, rec, e, table, names, max_per_host)
    assert np.array_equal(verdict & 3, e["o_v"])
    assert np.array_equal(off, e["w_off"]) and np.array_equal(ids, e["w_ids"])
    assert np.array_equal(use, e["w_use"])
    assert 0 < int(use.sum()) < len(use)
    # the marks by hand: name or variant, nothing else
    for d in rng.integers(0, len(table), size=60):
        want = [v for v, n in enumerate(names) if n == table[d] or variant(n) == table[d]]
        assert mark_ids[mark_off[d]:mark_off[d + 1]].tolist() == want
    # the other entry point: listed devices' strings + per-host lists of marked modules, host buffers
    i_off, i_ids = inuse_lists(rec, f["dev_off"], mark_off, mark_ids, len(table))
    off2, ids2, use2 = ctx.fleet_roots_labeled(e["Sk"], e["sk_off"], e["o_kept"], f["extra_off"], f["extra_ids"],
                                               f["safe"], e["bound"], i_off, i_ids)
    assert np.array_equal(off2, e["w_off"]) and np.array_equal(ids2, e["w_ids"]) and np.array_equal(use2, e["w_use"])
    # and without labels nothing else changes
    off3, ids3 = ctx.fleet_roots_bound(e["Sk"], e["sk_off"], e["o_kept"], f["extra_off"], f["extra_ids"], f["safe"],
                                       e["bound"])
    assert np.array_equal(off3, e["w_off"]) and np.array_equal(ids3, e["w_ids"])
