"""GPU parity of the device-list predicates and the string packing
(mcx_device_filter, mcx_device_filter_dev, mcx_fleet_compact_dev through the C
ABI) against the CPU oracle on the same seeded records.  Bit-exact."""
import numpy as np
import pytest

from montecarlo_amd import McxError, synth
from montecarlo_amd import binding as b
from montecarlo_amd.binding import pack_strings

from test_devfilter_cpu import DRV, HAND, hand_records

pytestmark = pytest.mark.gpu


def host_offsets(rng, N, H):
    """H hosts over N devices, some of them empty, one of them large"""
    cuts = np.sort(rng.integers(0, N + 1, size=H - 1))
    if H >= 10:
        cuts[: H // 10] = cuts[H // 10]      # a run of empty hosts
    return np.concatenate([[0], np.sort(cuts), [N]]).astype(np.uint32)


def test_hand_table_with_reasons(gpu_ctx):
    rec, want = hand_records()
    arena, off = pack_strings([d.encode() for d in DRV])
    v, kept = gpu_ctx.device_filter(rec, arena, off, np.array([0, len(rec)], np.uint32))
    assert (v & 3).tolist() == want.tolist()
    assert kept.tolist() == [0, int((want & 1).sum())]
    reasons = (v >> 4).tolist()
    assert all((r == 0) == bool(w & 1) for r, w in zip(reasons, want))
    # PCI rows of the table, in order: why each dropped one was dropped
    assert reasons[:10] == [0, 2, 2, 0, 2, 0, 3, 3, 3, 0]
    assert reasons[10:26] == [0, 5, 5, 6, 6, 5, 0, 7, 7, 0, 0, 0, 0, 0, 5, 5]
    assert reasons[26:34] == [0, 0, 4, 4, 4, 4, 4, 0]
    assert reasons[34:] == [0, 0, 0, 1, 1]


@pytest.mark.parametrize("seed,N,H,cap", [(1, 50000, 700, 0), (2, 50000, 700, 64), (3, 3000, 1, 0), (4, 3000, 3000, 1),
                                           (5, 70000, 10, 64), (6, 100, 400, 2)])
def test_filter_vs_oracle(gpu_ctx, oracle_mod, seed, N, H, cap):
    rng = np.random.default_rng(seed)
    S, s_off = synth.modalias_strings(seed, 2000, N)
    rec = synth.device_records(seed, S, s_off)
    arena, off = synth.driver_names_packed()
    dev_off = host_offsets(rng, N, H)
    v, kept = gpu_ctx.device_filter(rec, arena, off, dev_off, cap)
    o_v, o_kept = oracle_mod.device_filter(rec, arena, off, dev_off, cap)
    assert np.array_equal(v & 3, o_v)
    assert np.array_equal(kept, o_kept)
    assert np.array_equal((v >> 4) == 0, (v & 1) == 1)
    if cap:
        assert (np.diff(kept.astype(np.int64)) <= cap).all() and ((v >> 4) == 8).any()


def test_compact_and_match(gpu_ctx, oracle_mod):
    """listed strings packed on the device, then matched: the same answers as
    matching everything and dropping the unlisted devices afterwards"""
    seed, N, H, P = 11, 40000, 900, 3000
    rng = np.random.default_rng(seed)
    A, a_off, a_mod = synth.alias_table(seed, P)
    S, s_off = synth.modalias_strings(seed + 1, P, N)
    rec = synth.device_records(seed, S, s_off)
    arena, off = synth.driver_names_packed()
    dev_off = host_offsets(rng, N, H)
    ctx = gpu_ctx
    ctx.alias_table_load(A, a_off, a_mod)
    d_rec, d_arena, d_off = ctx.to_device(rec), ctx.to_device(arena), ctx.to_device(off)
    d_dev, d_S, d_soff = ctx.to_device(dev_off), ctx.to_device(S), ctx.to_device(s_off)
    d_v, d_kept = ctx.alloc(N), ctx.alloc(4 * (H + 1))
    K = ctx.device_filter_dev(d_rec.ptr, N, d_arena.ptr, d_off.ptr, len(off) - 1, d_dev.ptr, H, 64, d_v.ptr, d_kept.ptr)
    o_v, o_kept = oracle_mod.device_filter(rec, arena, off, dev_off, 64)
    assert K == o_kept[-1] and np.array_equal(d_v.download(np.uint8, N) & 3, o_v)
    d_S2, d_soff2, d_src = ctx.alloc(len(S)), ctx.alloc(4 * (K + 1)), ctx.alloc(4 * K)
    nbytes = ctx.fleet_compact_dev(d_v.ptr, d_S.ptr, d_soff.ptr, N, d_dev.ptr, H, d_kept.ptr, K, d_S2.ptr,
                                   d_soff2.ptr, d_src.ptr)
    keep = np.nonzero(o_v & 1)[0]
    lens = np.diff(s_off.astype(np.int64))[keep]
    want_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    raw = S.tobytes()
    want_S = np.frombuffer(b"".join(raw[s_off[i]:s_off[i + 1]] for i in keep), dtype=np.uint8)
    assert nbytes == len(want_S)
    assert np.array_equal(d_src.download(np.uint32, K), keep.astype(np.uint32))
    assert np.array_equal(d_soff2.download(np.uint32, K + 1), want_off)
    assert np.array_equal(d_S2.download(np.uint8, len(S))[:nbytes], want_S)
    # without the optional index array, and without waiting for the byte count
    d_S3, d_soff3 = ctx.alloc(len(S)), ctx.alloc(4 * (K + 1))
    assert ctx.fleet_compact_dev(d_v.ptr, d_S.ptr, d_soff.ptr, N, d_dev.ptr, H, d_kept.ptr, K, d_S3.ptr,
                                 d_soff3.ptr, None, want_bytes=False) is None
    assert np.array_equal(d_S3.download(np.uint8, len(S))[:nbytes], want_S)
    assert np.array_equal(d_soff3.download(np.uint32, K + 1), want_off)
    # the packed batch through the alias match
    d_f, d_c = ctx.alloc(4 * K), ctx.alloc(4 * K)
    ctx.alias_match_dev(d_S2.ptr, d_soff2.ptr, K, d_f.ptr, d_c.ptr)
    first, count = ctx.alias_match(S, s_off)
    assert np.array_equal(d_f.download(np.uint32, K), first[keep])
    assert np.array_equal(d_c.download(np.uint32, K), count[keep])
    for buf in (d_rec, d_arena, d_off, d_dev, d_S, d_soff, d_v, d_kept, d_S2, d_soff2, d_src, d_S3, d_soff3, d_f, d_c):
        buf.free()


def test_edge_cases(gpu_ctx):
    arena, off = pack_strings([d.encode() for d in DRV])
    empty = np.zeros(0, dtype=b.DEV_REC)
    v, kept = gpu_ctx.device_filter(empty, arena, off, np.zeros(1, np.uint32))
    assert len(v) == 0 and kept.tolist() == [0]
    v, kept = gpu_ctx.device_filter(empty, arena, off, np.zeros(5, np.uint32), 3)
    assert kept.tolist() == [0] * 5
    rec, want = hand_records()
    # no driver table at all: names cannot match, links still count
    no_names = rec.copy()
    no_names["driver"][no_names["driver"] < b.DRV_NOSLASH] = b.DRV_NONE
    v, _ = gpu_ctx.device_filter(no_names, np.zeros(0, np.uint8), np.zeros(1, np.uint32),
                                 np.array([0, len(rec)], np.uint32))
    assert (v[0] & 1) == 1 and (v[6] & 1) == 1 and (v[8] & 1) == 0
    with pytest.raises(McxError) as e:
        gpu_ctx.device_filter(rec, arena, off, np.array([0, 3], np.uint32))       # dev_off does not end at N
    assert e.value.rc == -2
    bad = rec.copy()
    bad["driver"][4] = len(DRV)
    with pytest.raises(McxError) as e:
        gpu_ctx.device_filter(bad, arena, off, np.array([0, len(rec)], np.uint32))
    assert e.value.rc == -2
    d = gpu_ctx.alloc(64)
    with pytest.raises(McxError):
        gpu_ctx.device_filter_dev(d.ptr + 4, 1, None, None, 0, d.ptr, 1, 0, d.ptr, None)   # misaligned records
    d.free()


def test_replaced_tables_reach_the_device(gpu_ctx):
    """driver names and bus tags come from the active table set, also on the GPU"""
    import montecarlo_amd as m

    rec = np.zeros(3, dtype=b.DEV_REC)
    rec["subsystem"] = b.SUB_PCI
    rec["driver"] = [0, 1, 2]
    arena, off = pack_strings([b"pcieport", b"ahci", b"nvme"])
    one_host = np.array([0, 3], np.uint32)
    A, a_off = pack_strings([b"pci:v00008086*", b"acpi*:PNP0C09:*", b"usb:v1234*"])
    a_mod = np.array([0, 1, 2], dtype=np.uint32)
    try:
        v, _ = gpu_ctx.device_filter(rec, arena, off, one_host)
        assert (v & 1).tolist() == [0, 1, 1]
        gpu_ctx.alias_table_load(A, a_off, a_mod)
        assert gpu_ctx.alias_hw_flags(3).tolist() == [1, 0, 1]
        m.tables_load(text="format 1\n[infra_driver]\nahci\nnvme\n[hw_alias_prefix]\nacpi\n")
        v, _ = gpu_ctx.device_filter(rec, arena, off, one_host)
        assert (v & 1).tolist() == [1, 0, 0] and (v >> 4).tolist() == [0, 3, 3]
        gpu_ctx.alias_table_load(A, a_off, a_mod)
        assert gpu_ctx.alias_hw_flags(3).tolist() == [0, 1, 0]
    finally:
        m.tables_reset()
    v, _ = gpu_ctx.device_filter(rec, arena, off, one_host)
    assert (v & 1).tolist() == [0, 1, 1]


@pytest.mark.parametrize("shift_in,shift_out", [(0, 0), (1, 0), (0, 2), (3, 1)])
def test_compact_arena_alignment(gpu_ctx, oracle_mod, shift_in, shift_out):
    """word-wide packing when both arenas are 4-byte aligned, byte-wide otherwise; strings of every
    length from 0 to 40 at every offset, so heads, tails and straddled words all occur"""
    rng = np.random.default_rng(5)
    strings = [bytes(rng.integers(33, 127, size=int(n), dtype=np.uint8)) for n in rng.integers(0, 41, size=3000)]
    S, s_off = pack_strings(strings)
    N, H = len(strings), 30
    rec = np.zeros(N, dtype=b.DEV_REC)
    rec["subsystem"] = np.where(rng.random(N) < 0.6, b.SUB_HID, b.SUB_OTHER)      # listed / not listed
    rec["driver"] = b.DRV_NONE
    dev_off = np.linspace(0, N, H + 1).astype(np.uint32)
    ctx = gpu_ctx
    d_rec, d_dev, d_soff = ctx.to_device(rec), ctx.to_device(dev_off), ctx.to_device(s_off)
    d_Sbuf = ctx.alloc(len(S) + 8)
    ctx.lib.mcx_h2d(ctx.h, d_Sbuf.ptr + shift_in, S.ctypes.data, len(S))
    d_v, d_kept = ctx.alloc(N), ctx.alloc(4 * (H + 1))
    K = ctx.device_filter_dev(d_rec.ptr, N, None, None, 0, d_dev.ptr, H, 0, d_v.ptr, d_kept.ptr)
    keep = np.nonzero(rec["subsystem"] == b.SUB_HID)[0]
    assert K == len(keep)
    d_out, d_soff2 = ctx.alloc(len(S) + 8), ctx.alloc(4 * (K + 1))
    nbytes = ctx.fleet_compact_dev(d_v.ptr, d_Sbuf.ptr + shift_in, d_soff.ptr, N, d_dev.ptr, H, d_kept.ptr, K,
                                   d_out.ptr + shift_out, d_soff2.ptr)
    want = b"".join(strings[i] for i in keep)
    assert nbytes == len(want)
    got = d_out.download(np.uint8, len(S) + 8)[shift_out:shift_out + nbytes].tobytes()
    assert got == want
    for buf in (d_rec, d_dev, d_soff, d_Sbuf, d_v, d_kept, d_out, d_soff2):
        buf.free()


def numpy_listed(rec, names):
    """the predicates once more, vectorised (no per-host limit): a third statement of
    montecarlo/libmontecarlo.c:233-440, 729-823 beside the kernel and the C oracle"""
    infra = np.array([n in ("pcieport", "pci_bridge", "pciehp", "pcie_aspm", "pcie_pme", "pcie_edr", "shpchp",
                            "piix4_smbus") for n in names] + [False])
    host = np.array([("hcd" in n) or n == "hub" for n in names] + [False])
    drv = rec["driver"].astype(np.int64)
    known = drv < len(names)
    didx = np.where(known, drv, len(names))
    sub, dt, at = rec["subsystem"], rec["devtype"], rec["attrs"]
    cls = rec["pci_class"]
    base, subc = (cls >> 16) & 0xFF, (cls >> 8) & 0xFF
    pci_bad = ((at & b.DA_PCI_CLASS) != 0) & ((base == 6) | ((base == 0xC) & (subc == 5)) | (base == 8))
    pci_bad |= (rec["driver"] == b.DRV_NOSLASH) | infra[didx]
    scsi_bad = np.isin(dt, [b.DT_NONE, b.DT_SCSI_HOST, b.DT_SCSI_TARGET, b.DT_SCSI_GENERIC]) | \
        ((at & (b.DA_SCSI_MODEL | b.DA_SCSI_VENDOR)) == 0)
    usb_bad = (dt != b.DT_USB_INTERFACE) | (((at & b.DA_IFACE_CLASS) != 0) & (rec["iface_class"] == 9)) | \
        ((at & b.DA_PARENT) == 0) | (((at & b.DA_PARENT_CLASS) != 0) & (rec["parent_class"] == 9)) | host[didx]
    listed = np.where(sub == b.SUB_PCI, ~pci_bad, np.where(sub == b.SUB_SCSI, ~scsi_bad, np.where(
        sub == b.SUB_USB, ~usb_bad, (sub == b.SUB_HID) | (sub == b.SUB_PCMCIA))))
    excl = (((at & b.DA_OWN_CLASS) != 0) & (rec["own_class"] == 8)) | \
        (((at & b.DA_IFACE_CLASS) != 0) & (rec["iface_class"] == 8)) | \
        (((at & b.DA_PARENT) != 0) & ((at & b.DA_PARENT_CLASS) != 0) & (rec["parent_class"] == 8))
    return listed, excl


def test_filter_full_size_properties(gpu_ctx, oracle_mod):
    """the fleet of the benchmark definition (10^5 hosts, 10^7 device records): verdicts against the
    vectorised statement of the predicates, the per-host limit and the packed strings through
    properties that hold at any size, a slice of hosts against the C oracle"""
    H, P = 100_000, 30_000
    dev_cnt, _, _ = synth.fleet_hosts(7, 6000, H, 100, 8)
    dev_off = np.zeros(H + 1, dtype=np.uint32)
    dev_off[1:] = np.cumsum(dev_cnt)
    N = int(dev_off[H])
    S, s_off = synth.modalias_strings(8, P, N)
    rec = synth.device_records(9, S, s_off)
    arena, off = synth.driver_names_packed()
    ctx = gpu_ctx
    d_rec, d_arena, d_off = ctx.to_device(rec), ctx.to_device(arena), ctx.to_device(off)
    d_dev, d_S, d_soff = ctx.to_device(dev_off), ctx.to_device(S), ctx.to_device(s_off)
    d_v, d_kept = ctx.alloc(N), ctx.alloc(4 * (H + 1))
    D = len(off) - 1
    # no limit: every verdict
    K0 = ctx.device_filter_dev(d_rec.ptr, N, d_arena.ptr, d_off.ptr, D, d_dev.ptr, H, 0, d_v.ptr, d_kept.ptr)
    v0 = d_v.download(np.uint8, N)
    listed, excl = numpy_listed(rec, synth.DRIVER_NAMES)
    assert np.array_equal((v0 & 1) != 0, listed) and np.array_equal((v0 & 2) != 0, excl)
    assert K0 == int(listed.sum())
    per_host = np.add.reduceat(listed.astype(np.int64), dev_off[:-1].astype(np.int64)) * (dev_cnt > 0)
    assert np.array_equal(np.diff(d_kept.download(np.uint32, H + 1).astype(np.int64)), per_host)
    # limit 64: the first 64 listed devices of every host, nothing else changes
    K = ctx.device_filter_dev(d_rec.ptr, N, d_arena.ptr, d_off.ptr, D, d_dev.ptr, H, 64, d_v.ptr, d_kept.ptr)
    v = d_v.download(np.uint8, N)
    kept = d_kept.download(np.uint32, H + 1)
    assert np.array_equal(np.diff(kept.astype(np.int64)), np.minimum(per_host, 64)) and K == kept[-1]
    rank = np.cumsum(listed) - listed                       # listed devices before i, fleet-wide
    host_of = np.repeat(np.arange(H), dev_cnt)
    first_rank = np.concatenate([[0], np.cumsum(per_host)])[:-1]
    in_host_rank = rank - first_rank[host_of]
    assert np.array_equal((v & 1) != 0, listed & (in_host_rank < 64))
    assert np.array_equal(v & 2, v0 & 2) and ((v >> 4) == 8).sum() == K0 - K
    lo, hi = 5000, 5300
    a, z = int(dev_off[lo]), int(dev_off[hi])
    o_v, o_kept = oracle_mod.device_filter(rec[a:z], arena, off, dev_off[lo:hi + 1] - dev_off[lo], 64)
    assert np.array_equal(v[a:z] & 3, o_v) and np.array_equal(kept[lo:hi + 1] - kept[lo], o_kept)
    # packing: lengths, order and bytes of the kept strings
    d_S2, d_soff2, d_src = ctx.alloc(len(S)), ctx.alloc(4 * (K + 1)), ctx.alloc(4 * K)
    nbytes = ctx.fleet_compact_dev(d_v.ptr, d_S.ptr, d_soff.ptr, N, d_dev.ptr, H, d_kept.ptr, K, d_S2.ptr,
                                   d_soff2.ptr, d_src.ptr)
    keep = np.nonzero(v & 1)[0]
    assert np.array_equal(d_src.download(np.uint32, K), keep.astype(np.uint32))
    lens = np.diff(s_off.astype(np.int64))[keep]
    off2 = d_soff2.download(np.uint32, K + 1)
    assert np.array_equal(np.diff(off2.astype(np.int64)), lens) and nbytes == lens.sum()
    S2 = d_S2.download(np.uint8, len(S))[:nbytes]
    # gather the same bytes on the host: position p of the output comes from string keep[k] at p - off2[k]
    k_of = np.repeat(np.arange(K), lens)
    src_pos = s_off[keep][k_of].astype(np.int64) + (np.arange(nbytes) - off2[:-1][k_of].astype(np.int64))
    assert np.array_equal(S2, S[src_pos])
    for buf in (d_rec, d_arena, d_off, d_dev, d_S, d_soff, d_v, d_kept, d_S2, d_soff2, d_src):
        buf.free()
