"""View-parallel use of the operator: one process per GPU, independent camera views dealt to
ranks, point-cloud parameters replicated, ONE sum all-reduce of the point gradients per step
(SURVEY 8e).  The reference has no distributed code; this is the only collective the path
has.  Backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.

The operator returns grad_pointcloud (N,3) and grad_pointcloud_features (N,56) as two views
of one flat 59*N float buffer, so the all-reduce is a single RCCL call on 236*N bytes.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> (int, int, int):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank); a no-op single-process result when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def views_of_rank(n_views: int, rank: int, world_size: int) -> List[int]:
    """Round-robin deal of view indices: rank r renders views r, r+W, r+2W, ..."""
    return list(range(rank, n_views, world_size))


def _flat_base(grad_pc: torch.Tensor, grad_feat: torch.Tensor) -> Optional[torch.Tensor]:
    """The flat 59*N buffer if the two gradients are adjacent views of one allocation, else None."""
    n = grad_pc.shape[0]
    if not (grad_pc.dtype == grad_feat.dtype == torch.float32 and grad_pc.is_contiguous() and grad_feat.is_contiguous()
            and grad_pc.device == grad_feat.device and n == grad_feat.shape[0]
            and grad_pc.untyped_storage().data_ptr() == grad_feat.untyped_storage().data_ptr()):
        return None
    if grad_pc.storage_offset() == grad_feat.storage_offset() + 56 * n:       # [features | positions], the operator's layout
        return torch.as_strided(grad_feat, (59 * n,), (1,), grad_feat.storage_offset())
    if grad_feat.storage_offset() == grad_pc.storage_offset() + 3 * n:        # [positions | features]
        return torch.as_strided(grad_pc, (59 * n,), (1,), grad_pc.storage_offset())
    return None


def all_reduce_point_gradients(grad_pc: torch.Tensor, grad_feat: torch.Tensor, group=None, average: bool = False):
    """Sum (or mean) the point gradients over all ranks, in place.  Returns the number of collectives issued."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    flat = _flat_base(grad_pc, grad_feat)
    bufs = [flat] if flat is not None else [grad_pc, grad_feat]
    for b in bufs:
        dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
        if average:
            b.div_(dist.get_world_size(group))
    return len(bufs)
