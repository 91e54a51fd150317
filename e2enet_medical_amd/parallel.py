"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

* Sliding-window inference (BASELINE config 4), two exchanges (``SegmentationNetwork.tile_exchange``): the default
  "allgather" below, bit-identical to one GPU, and "allreduce" (``run_tiles_partial``: partial volumes, one all-reduce at the
  end, <= 1e-6 from the single-process result).  All-gather: the x->y->z tile list is dealt round-robin over the ranks; weights
  are replicated; per group of `world` consecutive tiles one asynchronous all-gather moves the ranks' mirror-averaged
  probability patches to all ranks while the next group's tiles are computed; every rank overlap-adds the groups in
  the reference's tile order (neural_network.py:373-393), so fp32 summation order -- and the result -- is identical
  to the single-GPU run.  xGMI is point-to-point: an all-gather lets every peer push its shard over its own link
  instead of funnelling partial volumes through a ring all-reduce.
* Data-parallel training (config 5): replicas with identical weights; gradients are averaged with one flat
  all-reduce before the fused clip+SGD step (the clip norm must see the reduced gradients); after every prune/grow
  ``Masking.sync_kernel_maps`` broadcasts rank 0's kernel maps (host mirror, device maps, element masks and liveness
  tables are all rebuilt from them), so the random growth stays consistent whatever the ranks' RNG states are; with
  ``batch_dice`` the per-class tp/fp/fn sums are all-reduced between the two loss kernels (``batch_dice_allreduce``).
  ``nnUNetTrainer_simple.run_iteration`` wires all three when ``torch.distributed`` is initialised.
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


def _all_gather_async(out: torch.Tensor, mine: torch.Tensor, world: int, group=None):
    """out [world, ...] <- every rank's `mine`; returns the work handle (the collective runs on the process group's stream)"""
    if mine.is_cuda:
        return dist.all_gather_into_tensor(out.view(-1), mine.view(-1), group=group, async_op=True)
    return dist.all_gather([out[r] for r in range(world)], mine, group=group, async_op=True)      # gloo (CPU tests)


def run_tiles_sharded(num_tiles: int, rank: int, world: int, group, predict_tile, accumulate, patch_shape, device,
                      dtype=torch.float32, pipelined: bool = True, stats: dict = None):
    """The sharded tile loop of sliding-window inference (BASELINE config 4), shared by ``predict_3D`` and the tests.

    The x -> y -> z tile list is cut into groups of ``world`` consecutive tiles; in group g this rank evaluates tile
    ``g * world + rank`` (``predict_tile(ti) -> [K, px, py, pz]`` mirror-averaged probabilities).  Pipelined form: the group's
    patches are exchanged by ONE asynchronous all-gather on the process group's stream while the next group's tile is being
    computed; when a group has landed its ``world`` tiles are handed to ``accumulate(ti, patch)`` in tile order.  Groups are
    drained in order, so every rank overlap-adds all tiles in the reference's x -> y -> z order (neural_network.py:373-393): the
    fp32 sums -- and the result -- are bit-identical to the single-process loop.  Two group buffers are live (2 x world
    patches: 2.1 GB at K = 16, 128^3, 8 ranks) instead of every tile of the volume (14.5 GB for the AMOS-shaped benchmark volume).
    ``pipelined=False``: the former blocking form (all local tiles, one all-gather of everything, then the overlap-add);
    kept for A/B timing.  ``stats`` (dict) receives tile counts; with ``stats["time"]`` set (the benchmark does) it also receives
    the time the compute stream spent waiting for collectives -- that costs two events per group and ONE device synchronisation
    at the end, which an ordinary inference call does not pay.  Returns None in both forms (the result is what ``accumulate``
    has been handed)."""
    if stats is None:
        stats = {}
    timed = bool(stats.get("time"))
    mine_tiles = partition_tiles(num_tiles, rank, world)
    stats.update(tiles_total=num_tiles, tiles_local=len(mine_tiles), world=world, pipelined=bool(pipelined),
                 mode="allgather_patches")
    if not pipelined:
        per = slots_per_rank(num_tiles, world)
        mine = torch.zeros((per,) + tuple(patch_shape), dtype=dtype, device=device)
        for slot, ti in enumerate(mine_tiles):
            mine[slot].copy_(predict_tile(ti))
        gathered = gather_patches(mine, world, group)
        for ti in range(num_tiles):
            owner, slot = tile_slot(ti, world)
            accumulate(ti, gathered[owner, slot])
        return None
    ngroups = slots_per_rank(num_tiles, world)
    send = [torch.zeros(tuple(patch_shape), dtype=dtype, device=device) for _ in range(2)]
    recv = [torch.empty((world,) + tuple(patch_shape), dtype=dtype, device=device) for _ in range(2)]
    cuda = torch.device(device).type == "cuda"
    waits = []

    def drain(g, handle):
        if cuda and timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            handle.wait()                       # the compute stream waits for the collective; the host does not
            e1.record()
            waits.append((e0, e1))
        else:
            handle.wait()
        for r in range(world):
            ti = g * world + r
            if ti < num_tiles:
                accumulate(ti, recv[g & 1][r])
    pending = None
    for g in range(ngroups):
        ti = g * world + rank
        if ti < num_tiles:
            send[g & 1].copy_(predict_tile(ti))
        # (buffer g & 1 was last used by group g - 2, drained one iteration ago: its accumulate kernels are ahead of this
        #  collective on the compute stream, and its own collective has been waited for)
        handle = _all_gather_async(recv[g & 1], send[g & 1], world, group)
        if pending is not None:
            drain(*pending)
        pending = (g, handle)
    if pending is not None:
        drain(*pending)
    if waits:
        torch.cuda.synchronize()
        stats["collective_wait_ms"] = sum(a.elapsed_time(b) for a, b in waits)
    stats["groups"] = ngroups
    stats["exchange_buffer_bytes"] = 2 * (world + 1) * int(torch.tensor(patch_shape).prod()) * 4
    return None


def run_tiles_partial(num_tiles: int, rank: int, world: int, group, predict_tile, accumulate, count_only, agg: torch.Tensor,
                      stats: dict = None):
    """The cheaper exchange SURVEY section 8e names beside the all-gather: every rank overlap-adds ITS tiles
    (``rank::world`` of the x -> y -> z list) into its own partial ``[K, X, Y, Z]`` volume ``agg`` -- ``accumulate(ti, patch)``
    -- and ONE all-reduce (sum) of the partial volumes at the end replaces the per-group all-gathers.  The weight map is
    not exchanged: it does not depend on the data, so every rank adds the Gaussian of ALL tiles to its own copy in tile order
    (``count_only(ti)``: e2e_sw_accumulate with a null patch), bit-identical to the single-process map.

    Bytes: 2 (N-1)/N x K X Y Z x 4 once (3.9 GB at K = 16, 220 x 400 x 400, N = 8) instead of (N-1) x K x 128^3 x 4 per rank and
    group (0.94 GB x 14 groups).  It cannot be overlapped with compute (it needs every tile), and it changes the fp32 summation
    order of overlapping tiles -- per voxel the tiles of one rank are added first, then the ranks' partial sums in the
    collective's order -- so the probabilities differ from the single-process result in the last bits (<= 1e-6; the all-gather
    form is bit-identical).  Independent-tile structure: reference neural_network.py:373-393."""
    if stats is None:
        stats = {}
    timed = bool(stats.get("time"))
    mine = set(partition_tiles(num_tiles, rank, world))
    stats.update(tiles_total=num_tiles, tiles_local=len(mine), world=world, mode="allreduce_partial_volumes", pipelined=False)
    for ti in range(num_tiles):
        if ti in mine:
            accumulate(ti, predict_tile(ti))
        else:
            count_only(ti)
    cuda = agg.is_cuda
    if cuda and timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if world > 1 or stats.get("force"):
        dist.all_reduce(agg, op=dist.ReduceOp.SUM, group=group)
    if cuda and timed:
        e1.record()
        torch.cuda.synchronize()
        stats["collective_wait_ms"] = e0.elapsed_time(e1)
    stats["exchange_buffer_bytes"] = 0                       # in place on the partial volume
    stats["allreduce_bytes"] = int(agg.numel()) * 4
    return None


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


def batch_dice_allreduce(group=None):
    """Engine.batch_dice_hook for data-parallel batch dice (reference nnUNetTrainerV2_DDP.py:263-268): sums the folded
    per-class (tp, fp, fn) doubles over the ranks in place."""
    def hook(t: torch.Tensor):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return hook
