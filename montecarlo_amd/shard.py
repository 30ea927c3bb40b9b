"""Multi-GPU sharding of the alias match (one process per GPU, torch.distributed).

Two ways to cut the work, both exact:

* by string (the default): every rank holds the whole alias table and matches its
  own slice of the batch.  Nothing is exchanged on the data path.
* by pattern (configs[3] of the benchmark definition, "alias table sharded across 2/4/8 GPUs"): every
  rank holds a contiguous slice [j0, j1) of the table and matches all strings
  against it.  A string's global first match is the smallest of the per-rank
  first matches (shifted by j0) and its match count is the sum of the per-rank
  counts: one MIN and one SUM all-reduce of N words each (RCCL over xGMI on the
  GPUs, gloo on CPU tensors in the tests).

Match lists of a pattern-sharded match are merged by all-gather (gather_match_lists);
the fleet pipeline shards by host, and only its per-host result lists are gathered at
the end (gather_host_lists).

The reference has no multi-process path at all; this module is host logic of this
build only.
"""
import torch

NO_MATCH = 0xFFFFFFFF
_FLIP = -2147483648  # 0x80000000 as int32: u32 order -> i32 order


def shard_range(n, rank, world):
    """contiguous, balanced slice of range(n) owned by `rank`"""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def globalize_first(first_i32, j0):
    """in place: local first_idx (u32 bits in an int32 tensor) -> global index,
    NO_MATCH kept, then mapped so that signed MIN is unsigned MIN"""
    hit = first_i32 != -1
    first_i32.add_(hit.to(first_i32.dtype) * int(j0))
    first_i32.bitwise_xor_(_FLIP)
    return first_i32


def combine_pattern_shards(first_i32, count_i32, j0, group=None):
    """All-reduce the per-rank results of a pattern-sharded match, in place.

    first_i32 / count_i32: int32 tensors holding the u32 outputs of
    mcx_alias_match for this rank's table slice that starts at global pattern j0.
    Afterwards every rank holds the global first_idx (u32 bits) and count.
    """
    import torch.distributed as dist

    globalize_first(first_i32, j0)
    dist.all_reduce(first_i32, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(count_i32, op=dist.ReduceOp.SUM, group=group)
    first_i32.bitwise_xor_(_FLIP)
    return first_i32, count_i32


def as_i32(u32_array):
    """numpy uint32 -> torch int32 with the same bits"""
    import numpy as np

    return torch.from_numpy(np.ascontiguousarray(u32_array).view(np.int32).copy())


def as_u32(i32_tensor):
    import numpy as np

    return i32_tensor.cpu().numpy().view(np.uint32)


def _all_gather_ragged(t, group=None):
    """all-gather of 1-D tensors of different lengths: the lengths first, then the
    tensors padded to the longest (one collective each); returns the per-rank pieces"""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(x.item()) for x in sizes]
    width = max(max(sizes), 1)
    padded = torch.zeros(width, dtype=t.dtype, device=t.device)
    padded[: t.numel()] = t
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return [p[:k] for p, k in zip(parts, sizes)]


def gather_match_lists(list_off, list_idx, j0, group=None):
    """Merge the match lists of a pattern-sharded match on every rank.

    list_off (N+1) / list_idx: this rank's CSR from mcx_alias_match_list over its table
    slice starting at global pattern j0, as int32/int64 tensors holding the u32 values
    (CPU tensors with gloo, device tensors with RCCL).  Table slices are contiguous and
    ascend with the rank, so a string's global list is the ranks' lists one after the
    other, each shifted by its j0.  Returns (off int64[N+1], idx int64[total]).
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    off = list_off.to(torch.int64)
    cnt = off[1:] - off[:-1]
    counts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt.contiguous(), group=group)
    glob = (list_idx.to(torch.int64) & 0xFFFFFFFF) + int(j0)
    parts = _all_gather_ragged(glob.contiguous(), group)
    total = torch.stack(counts).sum(dim=0)
    out_off = torch.zeros(off.numel(), dtype=torch.int64, device=off.device)
    out_off[1:] = torch.cumsum(total, dim=0)
    out = torch.empty(int(out_off[-1].item()), dtype=torch.int64, device=off.device)
    before = torch.zeros_like(total)  # entries of lower ranks, per string
    for r in range(world):
        c = counts[r]
        n = int(c.sum().item())
        if n:
            src_off = torch.cumsum(c, dim=0) - c
            owner = torch.repeat_interleave(torch.arange(c.numel(), device=c.device), c)
            within = torch.arange(n, device=c.device) - src_off[owner]
            out[out_off[:-1][owner] + before[owner] + within] = parts[r]
        before += c
    return out_off, out


def gather_host_lists(root_off, root_ids, group=None):
    """Concatenate the per-host result lists of a host-sharded fleet pass on every rank.

    Ranks own consecutive runs of hosts (shard_range over the hosts), so the fleet's
    CSR is the ranks' CSRs one after the other.  root_off (H_r+1) / root_ids: this rank's
    output of mcx_fleet_roots as integer tensors.  Returns (off int64[H+1], ids int64[total]).
    """
    off = root_off.to(torch.int64)
    cnt = (off[1:] - off[:-1]).contiguous()
    cnts = _all_gather_ragged(cnt, group)
    ids = _all_gather_ragged((root_ids.to(torch.int64) & 0xFFFFFFFF).contiguous(), group)
    allc = torch.cat(cnts)
    out_off = torch.zeros(allc.numel() + 1, dtype=torch.int64, device=off.device)
    out_off[1:] = torch.cumsum(allc, dim=0)
    return out_off, torch.cat(ids)
