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
