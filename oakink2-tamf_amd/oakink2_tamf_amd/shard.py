"""Clip sharding across ranks (one process per GPU) and the result gather.

The reference parallelises sampling as independent worker processes over contiguous dataset-index ranges with
no communication (launch/sample.py:198-199,264-292).  Here the same contiguous split is kept, Philox noise is
keyed by the *global* clip id so the split does not change any clip's result, and the only exchange is one
all_gather of the sampled poses (RCCL over xGMI on GPUs; gloo in the CPU tests)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def worker_range(n_items: int, worker_id: int, num_worker: int) -> Tuple[int, int]:
    """[start, stop) of worker `worker_id` - the formula of launch/sample.py:198-199."""
    return int(n_items * worker_id / num_worker), int(n_items * (worker_id + 1) / num_worker)


def clip_id_base(rank: int, clips_per_rank: int) -> int:
    """Global id of a rank's first clip when every rank holds `clips_per_rank` clips (weak scaling)."""
    return rank * clips_per_rank


def gather_clips(local: torch.Tensor, out: Optional[torch.Tensor] = None, group=None,
                 force_collective: bool = False) -> torch.Tensor:
    """all_gather equal-sized per-rank results along dim 0, in rank (= global clip) order.
    force_collective: issue the collective even in a group of one rank (the GPU test that puts the RCCL call itself
    on a device; a single rank otherwise just copies)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        if out is not None:
            out.copy_(local)
            return out
        return local
    if out is None:
        out = local.new_empty((world * local.shape[0],) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def gather_ragged(local: torch.Tensor, counts: List[int], group=None) -> torch.Tensor:
    """all_gather of per-rank results whose leading sizes differ (contiguous dataset split): pad to the
    maximum, gather, strip the padding."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    mx = max(counts)
    pad = local.new_zeros((mx,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    buf = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx : r * mx + counts[r]] for r in range(world)], dim=0)
