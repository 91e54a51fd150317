#!/usr/bin/env python
"""Benchmark of the hot path on MI355X: one training iteration (forward, deep-supervision Dice+CE, backward,
clip_grad_norm_, SGD-Nesterov, DSFF mask step) of the shiftConvPP network on synthetic BraTS-shaped patches.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 4-modal 128^3 patches, base width 32, 4 classes, pools [[2,2,2]]*5, DSFF density
0.2, batch 2 per GPU (the nnU-Net BraTS 3d_fullres plan batch), fp32.  N > 1: data-parallel replicas (weak scaling), one
flat RCCL all-reduce of the gradients per step.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PATCH = (128, 128, 128)
BASE, CIN, K, DENSITY, BATCH = 32, 4, 4, 0.2, 2
POOLS = [(2, 2, 2)] * 5
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3       # fp32-input MFMA = fp32 vector peak
FWD_BYTES_PER_VOXEL = 5393.0   # BASELINE.md §3 / SURVEY §8(d): algorithmic fwd bytes per voxel at 32 ch
TRAIN_BYTES_PER_VOXEL = 3 * FWD_BYTES_PER_VOXEL


def build(device, patch=PATCH, batch=BATCH, seed=0):
    from torch import nn
    from e2enet_medical_amd.network_architecture.unetpp_d import Generic_UNetPlusPlus
    from e2enet_medical_amd.network_architecture.initialization import InitWeights_He
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    from e2enet_medical_amd.training.fused_optim import FusedClipSGD
    torch.manual_seed(seed)
    net = Generic_UNetPlusPlus(patch, CIN, BASE, K, 5, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True},
                               nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU,
                               {'negative_slope': 1e-2, 'inplace': True}, True, False, lambda x: x, InitWeights_He(1e-2),
                               POOLS, None, False, True, True).to(device)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        update_frequency = 1200          # BASELINE configs[2]
        final_density = 0.05
    random.seed(seed)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 250 * 1000),
                   growth_mode='random', redistribution_mode='none', args=A())
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        mask.add_module(net, sparse_init='uniform', density=DENSITY)
    fused = FusedClipSGD(opt, list(net.named_parameters()), 12.0)
    return net, opt, mask, fused


def synthetic_batch(device, patch, batch, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((batch, CIN) + tuple(patch), generator=g)
    full = torch.randint(0, K, (batch, 1) + tuple(patch), generator=g).float()
    targets = [full[:, :, ::s, ::s, ::s].contiguous() for s in (1, 2, 4, 8)]
    return x.to(device), [t.to(device) for t in targets]


class KernelTimer:
    """HIP-event timing of one C-ABI entry point on the stream it is launched on (torch's current stream)."""

    def __init__(self, libobj, name):
        self.lib, self.name, self.orig = libobj, name, getattr(libobj, name)
        self.events, self.enabled = [], False
        setattr(libobj, name, self._call)

    def _call(self, *a):
        if not self.enabled:
            return self.orig(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.orig(*a)
        e1.record()
        self.events.append((e0, e1, a))

    def total_ms(self):
        return sum(e0.elapsed_time(e1) for e0, e1, _ in self.events)


TRAFFIC_FILE = "r01h_pmc_traffic.json"


def pmc_traffic(prefix):
    """HBM bytes per launch of the kernels whose name starts with `prefix`, from the committed PMC summary
    (measured by rocprofv3 outside this process: counters cannot be read from inside the benchmark)."""
    path = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    n = sum(v["launches"] for k, v in d.items() if k.startswith(prefix))
    if not n:
        return None
    return sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in d.items() if k.startswith(prefix)) / n


def wgrad_flops(args):
    # e2e_conv133_wgrad(chans, dy, dw, ws, B, Cin, Cout, Di, Hi, Wi, sd, sh, sw, stream): dense 2*9*Cin*Cout*voxels_out
    b, cin, cout, di, hi, wi, sd, sh, sw = args[4:13]
    vox = ((di - 1) // sd + 1) * ((hi - 1) // sh + 1) * ((wi - 1) // sw + 1)
    return 2.0 * 9 * cin * cout * vox * b


def cpu_baseline(seconds_budget=25.0):
    """The oracle (CPU restatement, kind "port") timed on the host cores on a bounded sample of the same workload:
    fwd + loss + bwd of one 64^3 patch (same net, same density).  Threads are capped (E2E_CPU_THREADS, default 16):
    torch-CPU on 3D convs of this size stops scaling there, and the box may expose far more logical CPUs than its
    cgroup lets us use (256 threads took minutes per step).  Hard wall-clock guard: after the first step the loop only
    continues while the projected time stays inside the budget."""
    import oracle
    from oracle import network as onet
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:                                       # cgroup v2 CPU quota (the GPU box: 256 logical CPUs visible, 16 granted)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            avail = min(avail, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    threads = max(1, min(avail, int(os.environ.get("E2E_CPU_THREADS", "16"))))
    torch.set_num_threads(threads)
    spec = oracle.make_spec(CIN, BASE, K, POOLS)
    params = oracle.init_params(spec, seed=0)
    names = oracle.masked_names(spec)
    random.seed(0)
    masks = oracle.uniform_kernel_masks(onet.param_shapes(spec), names, DENSITY)
    for n in names:
        params[n] = params[n] * masks[n]
    patch = (64, 64, 64)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, CIN) + patch, generator=g)
    full = torch.randint(0, K, (1, 1) + patch, generator=g).float()
    targets = [full[:, :, ::s, ::s, ::s].contiguous() for s in (1, 2, 4, 8)]
    w = oracle.ds_weights(5)

    def step():
        leaves = {n: p.detach().clone().requires_grad_(True) for n, p in params.items()}
        outs = oracle.forward(spec, leaves, x)
        loss = oracle.deep_supervision_loss(outs, targets, w)
        loss.backward()
        return float(loss.detach())
    t0 = time.time()
    step()                                     # first step doubles as warm-up and as the time probe
    first = time.time() - t0
    n, dt = 1, first
    if first < seconds_budget / 3:
        t1, n = time.time(), 0
        while n < 2 or ((time.time() - t1) + first < seconds_budget and n < 8):
            step()
            n += 1
        dt = time.time() - t1
    return {"value": n * 64 ** 3 / dt, "unit": "voxels/s", "cores": threads, "kind": "port",
            "sample": "%d fwd+loss+bwd steps of one 64^3 patch (B=1, 32 ch, density 0.2), torch-CPU oracle, %d threads, "
                      "%.1f s (+ %.1f s first step)" % (n, threads, dt, first)}


def main():
    # stdout carries exactly ONE line, the result JSON of rank 0.  Libraries write there too (RCCL prints a version
    # banner through C stdio at init, flushed at exit): from here on file descriptor 1 is stderr, and the JSON line goes
    # to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--forward-only", action="store_true", help="inference forward (reported under 'extra', never as value)")
    ap.add_argument("--op-profile", action="store_true", help="print per-entry-point GPU time (diagnostic)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus must equal WORLD_SIZE (launch N>1 with torch.distributed.run)"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    import torch.distributed as dist
    force_dist = os.environ.get("E2E_FORCE_DIST") == "1"      # self-test: run the RCCL path with a 1-rank group
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world)      # "nccl" is RCCL on ROCm

    from e2enet_medical_amd._lib import lib
    from e2enet_medical_amd import parallel
    patch = (args.patch,) * 3
    net, opt, mask, fused = build(device, patch, args.batch)
    names = [n for n, _ in net.named_parameters()]
    x, targets = synthetic_batch(device, patch, args.batch, seed=100 + rank)
    eng = net.engine(x)
    ds_w = np.array([8, 4, 2, 1, 0], dtype=np.float64) / 15.0
    overlap = None

    def train_step():
        nonlocal overlap
        eng.forward(x, True)
        if use_dist and overlap is None:
            eng.prepare_backward()
            overlap = parallel.OverlappedGradAllReduce(eng, force=force_dist)   # bucketed all-reduce under the backward pass
        loss = eng.loss_backward(targets, ds_w, batch_dice=False)
        if use_dist:
            overlap.finish()
        fused.step(eng.grads, mask.masks)
        mask.step(masks_already_applied=True)
        return loss

    def fwd_step():
        with torch.no_grad():
            eng.forward(x, False)

    step = fwd_step if args.forward_only else train_step
    L = lib()
    timers = {}
    if args.op_profile:
        for name in ["conv133_fwd", "conv133_dgrad", "conv133_wgrad", "in_stats_finalize", "in_lrelu_bwd", "convT_fwd",
                     "convT_dgrad", "convT_wgrad", "maxpool_fwd", "maxpool_bwd", "head1x1_fwd", "head1x1_dgrad",
                     "head1x1_wgrad", "dc_ce_reduce", "dc_ce_grad", "grad_sqnorm", "sgd_clip_mask_step"]:
            timers[name] = KernelTimer(L, name)
    else:
        timers["conv133_wgrad"] = KernelTimer(L, "conv133_wgrad")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    for t in timers.values():
        t.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for t in timers.values():
        t.enabled = False
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        vox_per_step = world * args.batch * patch[0] * patch[1] * patch[2]
        value = vox_per_step * args.steps / dt
        out = {
            "metric": "voxels/sec (train step fwd+bwd+update), 128^3 patch 32ch density=0.2",
            "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BraTS-shaped 4-modal %d^3 patches, shiftConvPP base 32, K=4, DSFF density 0.2, "
                                   "batch %d per GPU, %s" % (patch[0], args.batch, "inference forward (deep supervision heads on)"
                                                             if args.forward_only else
                                                             "fwd+loss+bwd+clip+SGD+mask step, dense (parity) wgrad"),
                       "parallelism": "dp%d" % world},
            "per_gpu_voxels_per_s": value / world,
            "hbm_roofline_frac_whole_step": (value / world) * TRAIN_BYTES_PER_VOXEL / (HBM_PEAK_GBS * 1e9),
        }
        if args.forward_only:
            out["metric"] = "voxels/sec (inference forward only), 128^3 patch 32ch density=0.2"
            out["hbm_roofline_frac_whole_step"] = (value / world) * FWD_BYTES_PER_VOXEL / (HBM_PEAK_GBS * 1e9)
        wt = timers.get("conv133_wgrad")
        if wt is not None and wt.events:
            ms = wt.total_ms()
            flops = sum(wgrad_flops(a) for _, _, a in wt.events)
            achieved = flops / (ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "conv133_wgrad_v3/v2/s2_kernel (+ slab reduce): every launch of e2e_conv133_wgrad",
                               "achieved": achieved, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                               "frac": achieved / MFMA_F32_PEAK_TF, "traffic": pmc_traffic("conv133_wgrad"),
                               "traffic_source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)" % TRAFFIC_FILE,
                               "launches": len(wt.events), "avg_ms": ms / len(wt.events),
                               "share_of_step": ms / (dt * 1e3)}
        if args.op_profile:
            prof = {k: round(t.total_ms() / args.steps, 3) for k, t in timers.items() if t.events}
            out["op_ms_per_step"] = dict(sorted(prof.items(), key=lambda kv: -kv[1]))
            launches = []
            for k, t in timers.items():
                per = len(t.events) // args.steps
                for i in range(per):                      # same launch index across steps = same layer
                    ms = sum(t.events[s * per + i][0].elapsed_time(t.events[s * per + i][1]) for s in range(args.steps)) / args.steps
                    ints = [a for a in t.events[i][2] if isinstance(a, int) and 0 < a < 100000]
                    launches.append((round(ms, 3), k, ints[:12]))
            launches.sort(key=lambda v: -v[0])
            out["top_launches"] = launches[:40]
        if not args.no_cpu_baseline and world == 1:      # the CPU port is timed on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
