"""N>1 host logic on CPU: two gloo ranks.  The per-rank match results come from the
CPU oracle here (there is no GPU on this side, and the product has no CPU path);
what is under test is the sharding arithmetic and the MIN/SUM combine that the
pattern-sharded mode runs over RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from montecarlo_amd import shard, synth

SEED, P, N = 21, 1500, 240


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # pattern-sharded: this rank's slice of the table, all strings
    j0, j1 = shard.shard_range(P, rank, world)
    A, a_off, a_mod = synth.alias_table(SEED, P, j0, j1)
    S, s_off = synth.modalias_strings(SEED, P, N)
    first, count = oracle.alias_match(A, a_off, S, s_off)
    f32, c32 = shard.as_i32(first), shard.as_i32(count)
    shard.combine_pattern_shards(f32, c32, j0)
    np.save(os.path.join(out_dir, f"pat_first_{rank}.npy"), shard.as_u32(f32))
    np.save(os.path.join(out_dir, f"pat_count_{rank}.npy"), shard.as_u32(c32))
    # string-sharded: the whole table, this rank's slice of the strings, no collective
    i0, i1 = shard.shard_range(N, rank, world)
    A, a_off, a_mod = synth.alias_table(SEED, P)
    Sr, sr_off = synth.modalias_strings(SEED, P, N, i0, i1)
    first, count = oracle.alias_match(A, a_off, Sr, sr_off)
    np.save(os.path.join(out_dir, f"str_first_{rank}.npy"), first)
    np.save(os.path.join(out_dir, f"str_count_{rank}.npy"), count)
    # match lists of the pattern-sharded mode, merged by all-gather
    j0, j1 = shard.shard_range(P, rank, world)
    A, a_off, a_mod = synth.alias_table(SEED, P, j0, j1)
    off, idx = oracle.alias_match_list(A, a_off, S, s_off)
    g_off, g_idx = shard.gather_match_lists(shard.as_i32(off), shard.as_i32(idx), j0)
    np.save(os.path.join(out_dir, f"list_off_{rank}.npy"), g_off.numpy())
    np.save(os.path.join(out_dir, f"list_idx_{rank}.npy"), g_idx.numpy())
    # per-host lists of a host-sharded fleet pass: ragged CSRs, rank 1 may own no host
    H = 7
    h0, h1 = shard.shard_range(H, rank, world)
    sizes = [(3 * h) % 5 for h in range(H)]
    ids = [np.arange(sizes[h], dtype=np.uint32) + 100 * h for h in range(h0, h1)]
    r_off = np.zeros(h1 - h0 + 1, dtype=np.uint32)
    r_off[1:] = np.cumsum([len(x) for x in ids])
    r_ids = np.concatenate(ids) if ids else np.zeros(0, dtype=np.uint32)
    f_off, f_ids = shard.gather_host_lists(shard.as_i32(r_off), shard.as_i32(r_ids))
    np.save(os.path.join(out_dir, f"fleet_off_{rank}.npy"), f_off.numpy())
    np.save(os.path.join(out_dir, f"fleet_ids_{rank}.npy"), f_ids.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            cuts = [shard.shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


def test_first_index_order_mapping():
    f = shard.as_i32(np.array([0, 5, 0xFFFFFFFF, 7], dtype=np.uint32))
    shard.globalize_first(f, 100)
    g = shard.as_i32(np.array([3, 0xFFFFFFFF, 0xFFFFFFFF, 2], dtype=np.uint32))
    shard.globalize_first(g, 0)
    m = torch.minimum(f, g)
    m.bitwise_xor_(-2147483648)
    assert shard.as_u32(m).tolist() == [3, 105, 0xFFFFFFFF, 2]


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_combine(tmp_path, oracle_mod, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    A, a_off, a_mod = synth.alias_table(SEED, P)
    S, s_off = synth.modalias_strings(SEED, P, N)
    first, count = oracle_mod.alias_match(A, a_off, S, s_off)
    assert (count > 0).sum() > N // 2
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"pat_first_{r}.npy"), first)
        assert np.array_equal(np.load(tmp_path / f"pat_count_{r}.npy"), count)
    sf = np.concatenate([np.load(tmp_path / f"str_first_{r}.npy") for r in range(world)])
    sc = np.concatenate([np.load(tmp_path / f"str_count_{r}.npy") for r in range(world)])
    assert np.array_equal(sf, first) and np.array_equal(sc, count)
    o_off, o_idx = oracle_mod.alias_match_list(A, a_off, S, s_off)
    sizes = [(3 * h) % 5 for h in range(7)]
    want_ids = np.concatenate([np.arange(sizes[h]) + 100 * h for h in range(7)])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"list_off_{r}.npy"), o_off.astype(np.int64))
        assert np.array_equal(np.load(tmp_path / f"list_idx_{r}.npy"), o_idx.astype(np.int64))
        assert np.array_equal(np.load(tmp_path / f"fleet_off_{r}.npy"), np.concatenate([[0], np.cumsum(sizes)]))
        assert np.array_equal(np.load(tmp_path / f"fleet_ids_{r}.npy"), want_ids)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher starts N ranks as child processes (before
    anything touches the GPU) and passes rank 0's line through; under torch.distributed.run it
    is a rank itself.  MCX_BENCH_DRYRUN=1 stops each rank after the rendezvous, so this runs
    without a GPU."""
    import json
    import subprocess
    import sys

    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = dict(os.environ, MCX_BENCH_DRYRUN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "5", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 1
    # one GPU: no children, same process
    r = subprocess.run([sys.executable, bench, "--gpus", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
