"""Multi-GPU use of the operator: one process per GPU.  The reference has no distributed code.
Backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.  Two schemes:

* view-parallel: parameters replicated, independent camera views dealt to ranks, the point gradients summed with an
  all-reduce.  The operator returns grad_pointcloud (N,3) and grad_pointcloud_features (N,56) as two views of one flat
  59*N float buffer, so the all-reduce is a single RCCL call on 236*N bytes (all_reduce_point_gradients).  With several
  views per rank per step the collective is either issued once per step on the locally accumulated gradient, or once
  per view asynchronously so that it runs beside the next view's forward + backward (OverlappedGradientReducer).

* Gaussian-parallel: every rank OWNS a contiguous shard of the Gaussians (parameters, gradients, optimiser state) and
  RENDERS one view.  Owners project their shard for every view, an all-to-all hands each renderer the projected splat
  records of its view (16 floats per in-camera point), the renderer blends and back-propagates to per-splat sums
  (12 floats), a second all-to-all returns them, owners finish the Jacobian chain.  No gradient is replicated and no
  all-reduce exists: per rank and step 112 B per in-camera point instead of 2 x 236 B per Gaussian of the scene
  (gaussian_parallel_step).  The four compute stages are libgsrast's staged entry points (stages.StagedRasteriser).
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


class OverlappedGradientReducer:
    """Several views per rank per step, one asynchronous sum all-reduce PER VIEW: submit() is called right after a view's
    backward and returns at once; the collective runs on the process group's own stream (RCCL) beside the next view's
    forward + backward on the compute stream, and finish() waits for all of them and adds the per-view results up.
    Each view's gradient stays in its own buffer while it is in flight (the operator allocates a fresh 59*N buffer per
    backward), so nothing is written under a running collective.  Costs V all-reduces of 236*N bytes per step instead of
    one; what it buys is that only the LAST one is exposed."""

    def __init__(self, group=None):
        self.group = group
        self._pending = []
        self.collectives = 0

    def submit(self, grad_pc: torch.Tensor, grad_feat: torch.Tensor):
        works = []
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            flat = _flat_base(grad_pc, grad_feat)
            for b in ([flat] if flat is not None else [grad_pc, grad_feat]):
                works.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self.collectives += 1
        self._pending.append((works, grad_pc, grad_feat))

    def finish(self):
        """Waits for every submitted view and returns (grad_pointcloud, grad_pointcloud_features) summed over this rank's
        views and over all ranks."""
        if not self._pending:
            raise RuntimeError("finish() without submit()")
        total_pc = total_feat = None
        for works, gpc, gft in self._pending:
            for w in works:
                w.wait()                      # orders the current stream after the collective; the host does not block on RCCL
            if total_pc is None:
                total_pc, total_feat = gpc, gft
            else:
                total_pc += gpc
                total_feat += gft
        self._pending = []
        return total_pc, total_feat


def shard_bounds(n_points: int, world_size: int):
    """Contiguous, near-equal id ranges: shard r owns [bounds[r], bounds[r+1]).  Contiguity matters: concatenating the
    shards' in-camera records in rank order is ascending point-id order, which is the tie order of the depth sort."""
    base, rem = divmod(n_points, world_size)
    b = [0]
    for r in range(world_size):
        b.append(b[-1] + base + (1 if r < rem else 0))
    return b


def _all_to_all_rows(send: torch.Tensor, send_rows, recv_rows, group=None) -> torch.Tensor:
    """Variable-size all-to-all of the rows of a 2-D tensor (rank j gets send_rows[j] rows, gives recv_rows[j])."""
    recv = send.new_empty((int(sum(recv_rows)), send.shape[1]))
    if dist.get_backend(group) == "gloo":        # ProcessGroupGloo has no all_to_all: point-to-point with the same splits
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        s_chunks = list(torch.split(send, [int(x) for x in send_rows]))
        r_chunks = list(torch.split(recv, [int(x) for x in recv_rows]))
        r_chunks[rank].copy_(s_chunks[rank])
        reqs = []
        for peer in range(world):
            if peer == rank:
                continue
            if send_rows[peer]:
                reqs.append(dist.isend(s_chunks[peer].contiguous(), peer, group=group))
            if recv_rows[peer]:
                reqs.append(dist.irecv(r_chunks[peer], peer, group=group))
        for r in reqs:
            r.wait()
    else:
        dist.all_to_all_single(recv, send.contiguous(), [int(x) for x in recv_rows], [int(x) for x in send_rows], group=group)
    return recv


def gaussian_parallel_step(backend, grad_of_image, group=None):
    """One Gaussian-parallel step: world_size views, view v rendered by rank v, every rank owning one shard.

    `backend` supplies the four compute stages for THIS rank's shard (stages.StagedRasteriser on the GPU through
    HipStageBackend below; the CPU tests plug the oracle's staged halves in):
        project(view)                      -> (records (M,16) tensor, handle)     per-point half, forward, shard x view
        render(records)                    -> (image tensor, handle)              per-pixel half, forward, my view
        backward_render(handle, g_image)   -> sums (M,12) tensor                  per-pixel half, backward
        backward_project(handle, sums)     -> (grad_pointcloud, grad_features)    per-point half, backward, shard x view
    `grad_of_image(image)` returns dL/dimage of this rank's view.
    Returns (image of this rank's view, grad_pointcloud, grad_pointcloud_features of the OWN shard summed over all views,
    stats).  Collectives per step: one small all-gather of the M counts and two all-to-alls; no all-reduce."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    recs, handles = [], []
    if hasattr(backend, "project_all"):          # queue all views' projections first, read their counts afterwards: one wait, not W
        for r, h in backend.project_all(range(world)):
            recs.append(r)
            handles.append(h)
    else:
        for v in range(world):
            r, h = backend.project(v)
            recs.append(r)
            handles.append(h)
    send_rows = [int(r.shape[0]) for r in recs]
    counts = torch.tensor(send_rows, dtype=torch.int64, device=recs[0].device)
    table = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(table, counts, group=group)                    # table[r'][v] = in-camera points of shard r' in view v
    recv_rows = torch.stack(table)[:, rank].tolist()       # ONE device->host read of the count table's column
    records = _all_to_all_rows(torch.cat(recs), send_rows, recv_rows, group)
    image, rhandle = backend.render(records)
    sums = backend.backward_render(rhandle, grad_of_image(image))
    sums_back = _all_to_all_rows(sums, recv_rows, send_rows, group)     # the same splits, reversed
    grad_pc = grad_feat = None
    off = 0
    for v in range(world):
        gpc, gft = backend.backward_project(handles[v], sums_back[off:off + send_rows[v]])
        off += send_rows[v]
        if grad_pc is None:
            grad_pc, grad_feat = gpc, gft
        else:
            grad_pc += gpc
            grad_feat += gft
    stats = {"records_rows_sent": sum(send_rows) - send_rows[rank], "records_rows_received": sum(recv_rows) - recv_rows[rank],
             "bytes_sent": (sum(send_rows) - send_rows[rank]) * 64 + (sum(recv_rows) - recv_rows[rank]) * 48,
             "collectives": 3}
    return image, grad_pc, grad_feat, stats


class HipStageBackend:
    """The four stages on this rank's GPU through libgsrast's staged entry points, for a shard given as a
    GaussianPointCloudRasterisationInput template (its camera pose fields are overwritten per view)."""

    def __init__(self, shard_input, view_poses, config=None):
        from .stages import StagedRasteriser
        self.st = StagedRasteriser(config)
        self.inp = shard_input
        self.poses = view_poses                     # list of (q (Kobj,4) tensor, t (Kobj,3) tensor), one per view
        self._outs = None

    def _input_for(self, view):
        import copy
        i = copy.copy(self.inp)
        i.q_pointcloud_camera, i.t_pointcloud_camera = self.poses[view]
        return i

    def project(self, view):
        rec, _ids, frame = self.st.project_shard(self._input_for(view))
        return rec, (view, frame)

    def project_all(self, views):
        frames = [(v, self.st.project_shard_begin(self._input_for(v))) for v in views]
        return [(self.st.project_shard_finish(f, want_ids=False)[0], (v, f)) for v, f in frames]

    def render(self, records):
        outs, frame = self.st.forward_projected(records.contiguous(), self.inp.camera_info)
        return outs.rasterized_image, (outs, frame)

    def backward_render(self, handle, grad_image):
        outs, frame = handle
        sums, _ = self.st.backward_projected(frame, outs, grad_image)
        return sums

    def backward_project(self, handle, sums):
        view, frame = handle
        g = self.st.backward_shard(frame, self._input_for(view), sums.contiguous())
        return g.grad_pointcloud, g.grad_pointcloud_features
