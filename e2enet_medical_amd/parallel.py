"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

* Sliding-window inference (BASELINE config 4): the x->y->z tile list is dealt round-robin over the ranks; weights
  are replicated; one all-gather moves every rank's Gaussian-weighted probability patches to all ranks, which then
  overlap-add them in the reference's tile order (neural_network.py:373-393), so fp32 summation order -- and the
  result -- is identical to the single-GPU run.  xGMI is point-to-point: an all-gather lets every peer push its
  shard over its own link instead of funnelling partial volumes through a ring all-reduce.
* Data-parallel training (config 5): replicas with identical weights; gradients are averaged with one flat
  all-reduce before the fused clip+SGD step (the clip norm must see the reduced gradients); the DSFF kernel maps
  are broadcast from rank 0 after every prune/grow so that the random growth stays consistent.
"""
from typing import Dict, List

import torch
import torch.distributed as dist


def partition_tiles(num_tiles: int, rank: int, world: int) -> List[int]:
    return list(range(rank, num_tiles, world))


def tile_slot(tile_index: int, world: int):
    """(owner rank, slot in the owner's buffer) of a tile."""
    return tile_index % world, tile_index // world


def slots_per_rank(num_tiles: int, world: int) -> int:
    return (num_tiles + world - 1) // world


def gather_patches(mine: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """mine: [slots, ...] -> [world, slots, ...] on every rank."""
    out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    if mine.is_cuda:
        dist.all_gather_into_tensor(out.view(-1), mine.contiguous().view(-1), group=group)
    else:                                   # gloo (CPU tests)
        parts = [out[r] for r in range(world)]
        dist.all_gather(parts, mine.contiguous(), group=group)
    return out


def allreduce_mean_gradients(grads: Dict[str, torch.Tensor], names: List[str], group=None, flat: torch.Tensor = None,
                             force: bool = False):
    """Average gradients over the ranks with ONE flat all-reduce (67 MB at 32 ch, 95.5 MB at 48 ch: latency-bound on
    per-link xGMI rings, so no bucketing below that size)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:            # force: exercise the collective path on a single rank (self-test)
        return flat
    total = sum(grads[n].numel() for n in names)
    if flat is None or flat.numel() != total:
        flat = torch.empty(total, dtype=torch.float32, device=grads[names[0]].device)
    off = 0
    for n in names:
        k = grads[n].numel()
        flat[off:off + k].copy_(grads[n].reshape(-1))
        off += k
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / world)
    off = 0
    for n in names:
        k = grads[n].numel()
        grads[n].copy_(flat[off:off + k].view_as(grads[n]))
        off += k
    return flat


def allreduce_mean_flat(flat: torch.Tensor, group=None, force: bool = False):
    """Average the engine's flat gradient buffer (Engine.grad_flat: every parameter gradient is a view into it) over
    the ranks in place: one RCCL all-reduce of 67 MB at 32 ch, no staging copies."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if world > 1:
        flat.mul_(1.0 / world)
    return flat


class OverlappedGradAllReduce:
    """Data-parallel gradient averaging overlapped with the backward pass.  The engine lays its flat gradient buffer
    out in backward-completion order and calls ``grad_bucket_hook(lo, hi)`` when a slice is final; each slice goes out
    as an asynchronous RCCL all-reduce (the process group's stream waits for the work issued so far, later backward
    kernels keep running on the compute stream), finish() joins them and applies the 1/world scale."""

    def __init__(self, engine, group=None, force: bool = False):
        self.engine, self.group, self.force = engine, group, force
        self.world = dist.get_world_size(group)
        self.handles = []
        self.active = self.world > 1 or force
        if self.active:
            engine.grad_bucket_hook = self._bucket

    def _bucket(self, lo, hi):
        self.handles.append(dist.all_reduce(self.engine.grad_flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                            async_op=True))

    def finish(self):
        if not self.active:
            return
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.world > 1:
            self.engine.grad_flat.mul_(1.0 / self.world)


def broadcast_kernel_masks(kmasks: Dict[str, torch.Tensor], src: int = 0, group=None):
    """Broadcast the uint8 kernel maps (1.39 M kernels at 32 ch = 1.4 MB) from ``src`` as one tensor."""
    names = list(kmasks.keys())
    flat = torch.cat([kmasks[n].reshape(-1) for n in names])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for n in names:
        k = kmasks[n].numel()
        kmasks[n].copy_(flat[off:off + k].view_as(kmasks[n]))
        off += k
    return kmasks
