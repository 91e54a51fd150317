#!/usr/bin/env python
"""Benchmark of the hot path on MI355X: one training iteration (forward, deep-supervision Dice+CE, backward,
clip_grad_norm_, SGD-Nesterov, DSFF mask step) of the shiftConvPP network on synthetic BraTS-shaped patches.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W        (no WORLD_SIZE in the environment: this process touches no GPU,
                                                          starts N children -- one rank per GPU -- and relays rank 0's line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 4-modal 128^3 patches, base width 32, 4 classes, pools [[2,2,2]]*5, DSFF density
0.2, batch 2 per GPU (the nnU-Net BraTS 3d_fullres plan batch), fp32.  N > 1: data-parallel replicas (weak scaling),
bucketed RCCL all-reduce of the gradients overlapped with the backward pass.  Prints ONE JSON line (rank 0):

  value / ms_per_step    K timed steps, uninstrumented
  roofline               the step's dominant kernel, the 1x3x3 conv walk (e2e_conv133_fwd + e2e_conv133_dgrad):
                         HIP-event time of every launch over extra instrumented steps, algorithmic bytes against the
                         HBM peak and live FLOPs (from the DSFF kernel maps) against the fp32 FMA peak
  roofline_secondary     the dense weight gradient (fp32 MFMA)
  forward_only           inference forward of the same batch
  sliding_window         BASELINE config 4 shape: predict_3D of a [1,220,400,400] volume, 16 classes, patch 128^3,
                         step 0.5, 8 mirrors, everything device resident
  dsff_update            BASELINE config 3: one Masking.truncate_weights() (prune + grow of all 35 masked tensors)
  config3                BASELINE config 3 at its own shape ([2,1,48,192,192], anisotropic pools, 14 classes): ms per training step
                         and the cost of one DSFF update measured inside the running loop
  width48                the headline configuration at base width 48 (the width the reference trainer hard-codes): ms per step,
                         whole-step HBM fraction, the conv kernels dispatched
  config5                BASELINE config 5's per-rank workload (AMOS-shaped, 1 modality, 16 classes) at DSFF density 0.1 and 0.5
  cpu_baseline           the CPU oracle on the same 128^3 patch (B = 1: fwd + loss + bwd), 1 warm-up + up to 3 timed
  parity                 the metric's "Dice vs CPU ref" half: engine vs that oracle on the IDENTICAL patch (the GPU network's
                         weights and masks): Dice of the argmax maps, max |dlogit| per head, loss difference
"""
import argparse
import contextlib
import io
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
FP32_PEAK_TF = 157.3           # fp32 vector FMA peak = fp32-input MFMA peak (MI355X_MICROARCH.md)
BF16_PEAK_TF = 2500.0          # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF headline figure includes 2:1 sparsity)
FWD_BYTES_PER_VOXEL = 5393.0   # BASELINE.md section 3 / SURVEY section 8(d): algorithmic fwd bytes per voxel at 32 ch
TRAIN_BYTES_PER_VOXEL = 3 * FWD_BYTES_PER_VOXEL
TRAFFIC_FILE = "r06_pmc_traffic.json"
DTYPE_NOTE = "f32 (fp16x2 split products, f32 accumulate)"   # inputs, outputs, statistics fp32 / fp64; conv products = three fp16 MFMAs on 11+11-bit pieces
PURE_FP32_ENV = {"E2E_CONV_MM": "0", "E2E_CONV_DENSE": "0", "E2E_WG_BF3": "0", "E2E_CT_BF3": "0"}   # every product an fp32 FMA / fp32-input MFMA
NOMINAL_MHZ = 2400.0           # the shader clock the guide's compute peaks are quoted at
CLOCK_NOTE = ("peaks are the 2.4 GHz figures of the guide; measured_clock_mhz = shader-clock cycles / wall time of workgroup 0 of "
              "every launch of this family inside the timed steps (s_memtime / s_memrealtime, e2e_diag_kernel_clock); "
              "frac_at_measured_clock prices the compute peak at that clock (profiles/r05_power_clock.txt: the same launches run at "
              "~2.35 GHz when confined to 64 CUs -- the chip is power-limited under them)")


def kernel_clock(libobj, family, reset=True):
    """(MHz, ms of workgroup-0 time) the launches of a hot-kernel family ran at since the last reset."""
    import ctypes
    mhz, ms = ctypes.c_double(0.0), ctypes.c_double(0.0)
    libobj.diag_kernel_clock(family, ctypes.byref(mhz), ctypes.byref(ms), 1 if reset else 0)
    return mhz.value, ms.value


def build(device, patch=PATCH, cin=CIN, k=K, seed=0, density=DENSITY, pools=None, update_frequency=1200, base=None):
    from torch import nn
    from e2enet_medical_amd.network_architecture.unetpp_d import Generic_UNetPlusPlus
    from e2enet_medical_amd.network_architecture.initialization import InitWeights_He
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    from e2enet_medical_amd.training.fused_optim import FusedClipSGD
    torch.manual_seed(seed)
    net = Generic_UNetPlusPlus(patch, cin, BASE if base is None else base, k, 5, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True},
                               nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU,
                               {'negative_slope': 1e-2, 'inplace': True}, True, False, lambda x: x, InitWeights_He(1e-2),
                               pools or POOLS, None, False, True, True).to(device)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        final_density = 0.05
    A.update_frequency = update_frequency          # 1200: BASELINE configs[2]
    random.seed(seed)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 250 * 1000),
                   growth_mode='random', redistribution_mode='none', args=A())
    with contextlib.redirect_stdout(io.StringIO()):
        mask.add_module(net, sparse_init='uniform', density=density)
    fused = FusedClipSGD(opt, list(net.named_parameters()), 12.0)
    return net, opt, mask, fused


def synthetic_batch(device, patch, batch, seed, cin=CIN, k=K):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((batch, cin) + tuple(patch), generator=g)
    full = torch.randint(0, k, (batch, 1) + tuple(patch), generator=g).float()
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


LIVE_TRAFFIC = None          # --live-traffic: the summary collected by this very invocation (collect_live_traffic)


def collect_live_traffic():
    """--live-traffic: run the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, counters only, one stream -- the
    recipe of tools/prof_bench.sh and of the guide's HBM section) over two steps of this benchmark as CHILD processes, before this
    process touches a GPU, and summarise them per kernel (tools/traffic_summary.py).  `roofline.traffic` then comes from the code
    that is being benchmarked instead of from the committed summary."""
    import shutil
    import subprocess
    import tempfile
    import importlib.util
    global LIVE_TRAFFIC
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    root = tempfile.mkdtemp(prefix="e2e_live_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", E2E_LANES="0", E2E_WGRAD_STREAM="0")
    env.pop("WORLD_SIZE", None)
    for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        cmd = [prof, "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(root, sub), "--", sys.executable,
               os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
        r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            sys.stderr.write("bench.py --live-traffic: %s pass failed (rc %d): %s\n" % (ctr, r.returncode, r.stderr[-400:]))
            return
    spec = importlib.util.spec_from_file_location("traffic_summary", os.path.join(ROOT, "tools", "traffic_summary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    LIVE_TRAFFIC = mod.summarise(root)
    shutil.rmtree(root, ignore_errors=True)


def traffic_source():
    if LIVE_TRAFFIC is not None:
        return ("live: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) and WRITE_SIZE passes run by this invocation over two steps "
                "of this benchmark (child processes, one stream), bytes per launch")
    return ("profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch; null when the summary was collected from "
            "another ABI version of the library; `bench.py --live-traffic` collects it in the run)" % TRAFFIC_FILE)


def pmc_traffic(*prefixes):
    """HBM bytes per launch of the kernels whose name starts with one of `prefixes`: from this invocation's own counter passes
    (--live-traffic) or from the committed PMC summary (measured by rocprofv3 outside this process, separate --pmc passes:
    tools/prof_bench.sh)."""
    path = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if LIVE_TRAFFIC is None and not os.path.exists(path):
        return None
    d = dict(LIVE_TRAFFIC) if LIVE_TRAFFIC is not None else json.load(open(path))
    meta = d.pop("_meta", None)
    from e2enet_medical_amd._lib import ABI_VERSION
    if meta is None or meta.get("abi_version") != ABI_VERSION:
        return None                          # collected from another revision of the library: stale, not reported
    sel = {k: v for k, v in d.items() if any(k.startswith(pf) for pf in prefixes)}
    n = sum(v["launches"] for v in sel.values())
    if not n:
        return None
    return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel.values()) / n


def conv_work(eng, mask):
    """Per conv op: algorithmic bytes of one fwd (= one dgrad) launch and its live / dense FLOPs, keyed by the device
    pointers the launches carry (chans table for e2e_conv133_fwd / wgrad, outs table for e2e_conv133_dgrad)."""
    by_ptr = {}
    for op in eng.conv_ops.values():
        b = op.out.shape[0]
        di, hi, wi = op.in_dims
        vin = b * op.cin * di * hi * wi
        vout = op.out.data.numel()
        km = mask.kmasks.get(op.w_name) if mask is not None else None
        live = int(km.sum().item()) if km is not None else op.cin * op.cout
        per_kernel = 2.0 * 9 * (vout / op.cout)
        rec = {"bytes": 4.0 * (vin + vout), "flops_live": per_kernel * live, "flops_dense": per_kernel * op.cin * op.cout}
        by_ptr[op.chans.data_ptr()] = rec
        if op.outs is not None:
            by_ptr[op.outs.data_ptr()] = rec
        if op.sp_fwd is not None:                        # load-balanced kernel: plane / destination tables in plan order
            by_ptr[op.sp_fwd.table.data_ptr()] = rec
        if op.sp_bwd is not None and op.out_structs is not None:
            by_ptr[op._bwd_table().data_ptr()] = rec
    return by_ptr


def host_cpu_info():
    """(threads to use, logical CPUs visible, cgroup CPU quota in cores or None)."""
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    quota_cores = None
    try:                                       # cgroup v2 CPU quota (the GPU box: 256 logical CPUs visible, 16 granted)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            quota_cores = int(quota) / int(period)
    except (OSError, ValueError):
        pass
    avail = visible if quota_cores is None else min(visible, max(1, int(quota_cores)))
    return max(1, min(avail, int(os.environ.get("E2E_CPU_THREADS", "16")))), visible, quota_cores


def host_threads():
    return host_cpu_info()[0]


def parity_sample(net, device, patch, ds_w):
    """GPU side of the metric's "Dice vs CPU ref" half: ONE patch (B = 1, seed 0) through the engine with
    the network's current weights and DSFF masks -- logits of the four heads and the deep-supervision loss -- plus everything
    the CPU oracle needs to evaluate the identical patch (cpu_baseline)."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, CIN) + tuple(patch), generator=g)
    full = torch.randint(0, K, (1, 1) + tuple(patch), generator=g).float()
    targets = [full[:, :, ::s, ::s, ::s].contiguous() for s in (1, 2, 4, 8)]
    xd = x.to(device)
    eng1 = net.engine(xd)
    outs = eng1.forward(xd, True)
    loss = eng1.loss_backward([t.to(device) for t in targets], ds_w, batch_dice=False)
    return {"x": x, "targets": targets, "logits": [o.detach().cpu() for o in outs], "loss": float(loss.item()),
            "params": {n: p.detach().cpu().clone() for n, p in net.named_parameters()}}


def cpu_baseline(sample, patch_edge=128, budget_s=150.0):
    """The oracle (CPU restatement, kind "port") on the host cores: fwd + loss + bwd of ONE patch of the benchmarked
    network (B = 1, 4 x 128^3, 32 ch, density 0.2; dense masked weights like the reference) with the GPU network's own
    weights, masks, input and targets (`sample`), 1 warm-up + 3 timed steps (the loop stops early only
    when the projected time leaves the budget), then a forward-only leg (no_grad, 1 warm-up + 3 timed).  The warm-up step's
    logits and loss are compared with the engine's on the identical patch: returns (cpu_baseline record, parity record)."""
    import oracle
    threads, visible, quota = host_cpu_info()
    torch.set_num_threads(threads)
    spec = oracle.make_spec(CIN, BASE, K, POOLS)
    params, x, targets = sample["params"], sample["x"], sample["targets"]
    w = oracle.ds_weights(5)
    keep = {}

    def step():
        leaves = {n: p.detach().clone().requires_grad_(True) for n, p in params.items()}
        outs = oracle.forward(spec, leaves, x)
        loss = oracle.deep_supervision_loss(outs, targets, w)
        loss.backward()
        if not keep:
            keep["logits"], keep["loss"] = [o.detach() for o in outs], float(loss.detach())
        return float(loss.detach())

    def fwd():
        with torch.no_grad():
            oracle.forward(spec, params, x, do_ds=False)
    t0 = time.time()
    step()                                     # warm-up
    first = time.time() - t0
    n, t1 = 0, time.time()
    while n < 3 and (n == 0 or (time.time() - t1) / n * (n + 1) + first < budget_s):
        step()
        n += 1
    dt = time.time() - t1
    fwd()
    nf, t2 = 0, time.time()
    while nf < 3 and (nf == 0 or time.time() - t0 < budget_s):
        fwd()
        nf += 1
    fdt = time.time() - t2
    base = {"value": n * patch_edge ** 3 / dt, "unit": "voxels/s", "cores": threads, "kind": "port",
            "sample": "%d timed fwd+loss+bwd steps (+1 warm-up of %.1f s) of one 4 x %d^3 patch (B=1, 32 ch, density 0.2, dense "
                      "masked weights: the GPU network's own), torch-CPU oracle, %d threads, %.1f s" % (n, first, patch_edge, threads, dt),
            "seconds_per_step": dt / n,
            "forward_only": {"value": nf * patch_edge ** 3 / fdt, "unit": "voxels/s", "seconds_per_patch": fdt / nf,
                             "sample": "%d timed inference forwards (+1 warm-up) of the same patch, no_grad" % nf},
            "host_cores_total": visible, "cgroup_cpu_quota_cores": quota, "torch_threads": threads,
            "reference_in_container": "the reference itself (imported in the build container, 8 threads, 64^3 patch, same net): "
                                      "2.43 s fwd+bwd against 2.08 s for this oracle there (DESIGN.md section 6); it cannot "
                                      "travel to the GPU box"}
    # ---- Dice(GPU argmax, CPU argmax) and max |dlogit| on the identical patch (metrics.py:106-121)
    dl = [float((a - b).abs().max()) for a, b in zip(sample["logits"], keep["logits"])]
    seg_g, seg_c = sample["logits"][0].argmax(1).numpy(), keep["logits"][0].argmax(1).numpy()
    dices = [oracle.hard_dice(seg_g, seg_c, label) for label in range(K)]
    parity = {"what": "engine vs the fp32 CPU oracle on the identical patch (B=1, 4 x %d^3), same weights, masks and targets" % patch_edge,
              "max_abs_dlogit": max(dl), "max_abs_dlogit_per_head": dl, "dice_vs_cpu": min(dices), "dice_vs_cpu_per_label": dices,
              "argmax_mismatch_voxels": int((seg_g != seg_c).sum()), "loss_gpu": sample["loss"], "loss_cpu": keep["loss"],
              "loss_abs_diff": abs(sample["loss"] - keep["loss"]), "tolerance": "1e-4 logit / 1e-3 Dice (BASELINE.json north_star)"}
    return base, parity


def pure_fp32_record(steps=6, warmup=2):
    """The same step with every split-operand path switched off (library knobs, read once per process: a child process): the vector
    walk for the convs, the fp32-input MFMA kernels for the weight gradients and the transposed convs.  Reported beside the headline so
    that what the fp16 two-piece products buy -- and that they are what the headline runs on -- is always visible."""
    import subprocess
    env = dict(os.environ, **PURE_FP32_ENV)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup), "--no-extras", "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        return {"env": PURE_FP32_ENV, "ms_per_step": d["ms_per_step"], "voxels_per_s": d["value"],
                "roofline_frac": d.get("roofline", {}).get("frac"), "roofline_ms_per_step": d.get("roofline", {}).get("ms_per_step"),
                "wgrad_ms_per_step": d.get("roofline_secondary", {}).get("ms_per_step"), "steps": steps,
                "what": "the headline workload with E2E_CONV_MM=0 E2E_CONV_DENSE=0 (convs on the fp32 vector walk), E2E_WG_BF3=0 "
                        "(weight gradients on fp32-input MFMA), E2E_CT_BF3=0 (transposed convs on the fp32 kernels), child process"}
    except Exception as e:                               # noqa: BLE001 -- a sub-record must not take the headline down
        return {"env": PURE_FP32_ENV, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def config1_parity_record(device):
    """BASELINE config 1 (Hippocampus-shaped plumbing case: [1,1,40,56,40], pools [[2,2,2]]*3+[[1,1,1]]*2, K 3, density 1.0, base 32):
    engine vs the fp32 CPU oracle vs an fp64 evaluation of the same graph, max |dlogit| per head.  Stated plainly in the record: on
    this configuration the engine is NOT within the literal 1e-4 of the CPU path on every head -- nor is the CPU path within 1e-4 of
    exact arithmetic (InstanceNorms over 8..175 voxels at the deep levels): the tests assert the triangle bound instead."""
    import oracle
    from torch import nn
    from e2enet_medical_amd.network_architecture.unetpp_d import Generic_UNetPlusPlus
    from e2enet_medical_amd.network_architecture.initialization import InitWeights_He
    pools = [(2, 2, 2)] * 3 + [(1, 1, 1)] * 2
    patch, cin, k = (40, 56, 40), 1, 3
    torch.manual_seed(11)
    net = Generic_UNetPlusPlus(patch, cin, BASE, k, 5, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                               {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                               lambda x: x, InitWeights_He(1e-2), pools, None, False, True, True).to(device)
    x = torch.randn((1, cin) + patch, generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        outs = [o.cpu() for o in net(x.to(device))]
    spec = oracle.make_spec(cin, BASE, k, pools)
    params = {n: p.detach().cpu().clone() for n, p in net.named_parameters()}
    with torch.no_grad():
        r32 = oracle.forward(spec, params, x)
        r64 = oracle.forward(spec, {n: p.double() for n, p in params.items()}, x.double())
    rec = {"what": "config 1 (Hippocampus-shaped, density 1.0, B=1, He init): max |dlogit| per head [full, 1/2, 1/4, 1/8]",
           "engine_vs_cpu32": [float((a - b).abs().max()) for a, b in zip(outs, r32)],
           "engine_vs_fp64": [float((a.double() - b).abs().max()) for a, b in zip(outs, r64)],
           "cpu32_vs_fp64": [float((a.double() - b).abs().max()) for a, b in zip(r32, r64)]}
    worst = max(rec["engine_vs_cpu32"])
    rec["config1_engine_vs_cpu"] = worst
    rec["note"] = ("engine vs the reference CPU path on config 1: %.2e max |dlogit| -- %s the literal 1e-4 bar of the north_star; the fp32 "
                   "CPU path itself is %.2e from an fp64 evaluation of the same graph here and the engine %.2e (two fp32 evaluations of "
                   "InstanceNorms over 8..175 voxels cannot agree to 1e-4); tests assert engine <= 1e-4 + the CPU path's own distance "
                   "from fp64 (tests/test_gpu_configs.py:_logit_bars).  The headline configuration (config 2) meets the literal bar: "
                   "see max_abs_dlogit above." % (worst, "EXCEEDS" if worst > 1e-4 else "within", max(rec["cpu32_vs_fp64"]),
                                                 max(rec["engine_vs_fp64"])))
    del net
    torch.cuda.empty_cache()
    return rec


def sliding_window_record(device, rank=0, world=1):
    """BASELINE config 4 shape: AMOS-like volume [1,220,400,400], 16 classes, base 32, patch 128^3,
    step 0.5, 8 mirrors.  Wall time of predict_3D with the volume already on the host as float32 (the call uploads it
    once, 141 MB; aggregation, softmax, flips, Gaussian weighting and argmax stay on the device).
    world > 1 (every rank calls this): the north_star's multi-GPU split -- tiles dealt round-robin over the ranks
    (net.shard_tiles), per group of `world` tiles one asynchronous RCCL all-gather of the probability patches under the next
    group's compute, every rank overlap-adds in the reference's order; time = max over ranks between two barriers."""
    import torch.distributed as dist
    from e2enet_medical_amd.utilities.nd_softmax import softmax_helper
    net, _, mask, _ = build(device, PATCH, cin=1, k=16, seed=1)
    net.inference_apply_nonlin = softmax_helper
    net.eval()
    net.do_ds = False
    if world > 1 or os.environ.get("E2E_FORCE_DIST") == "1":
        net.shard_tiles(rank, world, None, force=world == 1)
        net.time_sharding = True                     # collective_wait_ms in the record (one extra device sync per volume)
    vol = torch.randn((1, 220, 400, 400), generator=torch.Generator().manual_seed(7)).numpy()
    kw = dict(do_mirroring=True, mirror_axes=(0, 1, 2), use_sliding_window=True, step_size=0.5, patch_size=PATCH,
              use_gaussian=True, verbose=False)
    steps = net._compute_steps_for_sliding_window(PATCH, vol.shape[1:], 0.5)
    tiles = len(steps[0]) * len(steps[1]) * len(steps[2])
    net.predict_3D(vol[:, :128, :160, :160], **kw)          # warm-up: plan allocation for the 8-mirror batch (and RCCL channels)
    sharded = world > 1 or os.environ.get("E2E_FORCE_DIST") == "1"

    def timed_predict():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        seg, probs = net.predict_3D(vol, **kw)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, seg, probs
    dt, seg, probs = timed_predict()
    vox = float(np.prod(vol.shape[1:]))
    rec = {"workload": "predict_3D [1,220,400,400] N(0,1), K=16, base 32, density 0.2, patch 128^3, step 0.5, 8 mirrors "
                       "(%d tiles x 8 forwards, mirrors of a tile as one batch-8 forward)" % tiles,
           "seconds": dt, "volume_voxels_per_s": vox / dt, "patch_voxels_per_s": tiles * 8 * 128 ** 3 / dt,
           "includes": "host->device upload of the volume, device->host copy of seg + probs (1.4 GB)", "n_gpus": world}
    st = getattr(net, "last_shard_stats", None)
    if st and sharded:
        rec["sharding"] = dict(st)
        rec["sharding"]["mode"] = st.get("mode", "allgather_patches")
        counts = torch.zeros(world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(counts, torch.tensor([st["tiles_local"]], dtype=torch.int64, device=device))
        rec["sharding"]["tiles_per_rank"] = [int(v) for v in counts.tolist()]
        assert sum(rec["sharding"]["tiles_per_rank"]) == tiles, (rec["sharding"]["tiles_per_rank"], tiles)
        # calibration: the same all-gather (one group: `world` patches of [16,128^3] fp32), blocking, 5 times
        mine = torch.zeros((16,) + PATCH, dtype=torch.float32, device=device)
        outb = torch.empty((world, 16) + PATCH, dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(outb.view(-1), mine.view(-1))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            dist.all_gather_into_tensor(outb.view(-1), mine.view(-1))
        torch.cuda.synchronize()
        rec["sharding"]["allgather_ms_per_group_blocking"] = (time.perf_counter() - t1) / 5 * 1e3
        rec["sharding"]["allgather_bytes_per_group_per_rank_inbound"] = (world - 1) * mine.numel() * 4
        del mine, outb
        # the other exchange of SURVEY section 8e, beside it: partial volumes, ONE all-reduce at the end (fewer bytes, not
        # overlappable, <= 1e-6 from the all-gather form, which is bit-identical to one GPU)
        net.shard_tiles(rank, world, None, force=world == 1, exchange="allreduce")
        dt2, seg2, probs2 = timed_predict()
        st2 = dict(getattr(net, "last_shard_stats", {}))
        st2.pop("force", None)
        st2.update(seconds=dt2, volume_voxels_per_s=vox / dt2, max_abs_dprob_vs_allgather=float(np.abs(probs2 - probs).max()),
                   argmax_mismatch_voxels_vs_allgather=int((seg2 != seg).sum()))
        rec["sharding_allreduce"] = st2
        net.shard_tiles(rank, world, None, force=world == 1, exchange="allgather")
    return rec


def config3_record(device):
    """BASELINE config 3 at its SURVEY section 8d shape: BTCV-like [2, 1, 48, 192, 192], pools [[1,2,2],[2,2,2]x3,[1,2,2]], 14 classes, base
    32, DSFF density 0.2: ms per training step, and the measured cost of one DSFF update (Masking.truncate_weights inside
    mask.step, reference core_channel.py:290-317 / :556-611) INSIDE the running loop: the update fires every 6th step here
    (update_frequency 1200 in the reference; the step that carries it is timed against the median of the others)."""
    pools = [(1, 2, 2), (2, 2, 2), (2, 2, 2), (2, 2, 2), (1, 2, 2)]
    patch, k, b = (48, 192, 192), 14, 2
    net, opt, mask, fused = build(device, patch, cin=1, k=k, seed=3, pools=pools, update_frequency=6)
    g = torch.Generator().manual_seed(33)
    x = torch.randn((b, 1) + patch, generator=g).to(device)
    eng = net.engine(x)
    outs = eng.forward(x, True)
    targets = [torch.randint(0, k, (b, 1) + tuple(o.shape[2:]), generator=g).float().to(device) for o in outs]
    ds_w = np.array([8, 4, 2, 1, 0], dtype=np.float64) / 15.0
    times, updated = [], []
    for it in range(6 + 18):                       # 6 warm-up steps (they end with the first update: allocations), 18 timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.forward(x, True)
        eng.loss_backward(targets, ds_w, batch_dice=False)
        fused.step(eng.grads, mask.masks)
        up = mask.step(masks_already_applied=True)
        torch.cuda.synchronize()
        if it >= 6:
            times.append((time.perf_counter() - t0) * 1e3)
            updated.append(bool(up))
    plain = sorted(t for t, u in zip(times, updated) if not u)
    with_up = [t for t, u in zip(times, updated) if u]
    med = plain[len(plain) // 2]
    vox = b * patch[0] * patch[1] * patch[2]
    return {"workload": "BTCV-shaped [2,1,48,192,192], pools [[1,2,2],[2,2,2]x3,[1,2,2]], K=14, base 32, density 0.2, full training step",
            "ms_per_step": med, "voxels_per_s": vox / (med * 1e-3), "steps_timed": len(times), "updates_timed": len(with_up),
            "ms_step_with_dsff_update": sum(with_up) / max(1, len(with_up)),
            "dsff_update_ms_inside_loop": sum(with_up) / max(1, len(with_up)) - med,
            "amortised_ms_per_step_at_update_frequency_1200": (sum(with_up) / max(1, len(with_up)) - med) / 1200.0}


def step_record(device, what, patch, cin, k, base, density, batch, seed, bytes_per_voxel=None):
    """ms per full training step (forward, loss, backward, clip, SGD, mask step) of another configuration of the same network, and the
    conv kernels it dispatched to (e2e_last_kernel behind every conv entry point of one step)."""
    net, opt, mask, fused = build(device, patch, cin=cin, k=k, seed=seed, density=density, base=base)
    g = torch.Generator().manual_seed(seed + 100)
    x = torch.randn((batch, cin) + tuple(patch), generator=g).to(device)
    eng = net.engine(x)
    outs = eng.forward(x, True)
    targets = [torch.randint(0, k, (batch, 1) + tuple(o.shape[2:]), generator=g).float().to(device) for o in outs]
    ds_w = np.array([8, 4, 2, 1, 0], dtype=np.float64) / 15.0

    def one():
        eng.forward(x, True)
        eng.loss_backward(targets, ds_w, batch_dice=False)
        fused.step(eng.grads, mask.masks)
        mask.step(masks_already_applied=True)
    from e2enet_medical_amd._lib import lib
    L = lib()
    names = ["conv133_fwd", "conv133_fwd_splitk", "conv133_fwd_dense", "conv133_fwd_sparse", "conv133_fwd_mm", "conv133_dgrad",
             "conv133_dgrad_splitk", "conv133_dgrad_dense", "conv133_dgrad_sparse", "conv133_dgrad_mm", "conv133_wgrad"]
    seen, orig = {}, {}
    for nme in names:
        fn = getattr(L, nme)
        orig[nme] = fn

        def wrapped(*a, _fn=fn):
            _fn(*a)
            kname = (L.last_kernel() or b"").decode().split(" ")[0]
            seen[kname] = seen.get(kname, 0) + 1
        setattr(L, nme, wrapped)
    try:
        one()                                       # warm-up + kernel census
    finally:
        for nme, fn in orig.items():
            setattr(L, nme, fn)
    one()
    torch.cuda.synchronize()
    times = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    med = sorted(times)[len(times) // 2]
    vox = batch * patch[0] * patch[1] * patch[2]
    rec = {"workload": what, "ms_per_step": med, "voxels_per_s": vox / (med * 1e-3), "steps_timed": len(times),
           "conv_kernels_per_step": dict(sorted(seen.items()))}
    if bytes_per_voxel is not None:
        rec["hbm_roofline_frac_whole_step"] = vox / (med * 1e-3) * bytes_per_voxel / (HBM_PEAK_GBS * 1e9)
        rec["bytes_per_voxel"] = bytes_per_voxel
    del eng
    net._engines.clear()
    torch.cuda.empty_cache()
    return rec


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: this parent has made no GPU call (importing torch initialises nothing);
    it starts N fresh children of this same script, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would), relays rank 0's single JSON line on stdout (the other ranks' stdout goes to stderr), and
    returns the first non-zero exit code, stopping the remaining ranks (by their exact PIDs) as soon as one has failed."""
    import socket
    import subprocess
    import threading
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL between processes on this image)
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    deadline = float("inf")
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write("bench.py: rank %d exited with code %d, stopping the other ranks\n" % (r, code))
                for o in alive:
                    procs[o].terminate()
                deadline = time.time() + 15.0                # a rank stuck inside a collective or a kernel may not honour SIGTERM
        if rc != 0 and alive and time.time() > deadline:
            for o in alive:
                procs[o].kill()
            for o in alive:
                procs[o].wait()
            alive.clear()
        time.sleep(0.05)
    reader.join(timeout=10)
    if chunks and chunks[0]:
        sys.stdout.write(chunks[0].decode())
        sys.stdout.flush()
    return rc


def dry_rank(args, rank, world, json_fd):
    """E2E_BENCH_DRY=1 (CPU test of the launcher, tests/test_host_cpu.py): no GPU, gloo instead of RCCL; every rank joins the
    group, checks the all-reduce and rank 0 prints a line with the same `rccl` / per-rank fields as the real run."""
    import torch.distributed as dist
    if os.environ.get("E2E_BENCH_DRY_FAIL_RANK") == str(rank):      # launcher test: one rank dies before the rendezvous
        sys.exit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ones = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(ones)
    per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(per_rank, torch.tensor([float(rank + 1)], dtype=torch.float64))
    assert float(ones.item()) == float(world), "all-reduce of ones gave %r on %d ranks" % (float(ones.item()), world)
    if rank == 0:
        rec = {"metric": "dry run of the launcher (no GPU work)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "rccl": {"backend": "gloo", "world": dist.get_world_size(), "allreduce_of_ones": float(ones.item()),
                        "device_per_rank": list(range(world)),
                        "ms_per_step_without_allreduce": 0.0, "allreduce_exposed_ms": 0.0, "allreduce_bytes_per_step": 0},
               "ms_per_step_per_rank": [float(t.item()) for t in per_rank]}
        os.write(json_fd, (json.dumps(rec) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()


def main():
    # stdout carries exactly ONE line, the result JSON of rank 0.  Libraries write there too (RCCL prints a version
    # banner through C stdio at init, flushed at exit): from here on file descriptor 1 is stderr, and the JSON line goes
    # to the saved descriptor.
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the forward_only / sliding_window / dsff_update sub-records")
    ap.add_argument("--forward-only", action="store_true", help="time the inference forward as the main loop (diagnostic)")
    ap.add_argument("--op-profile", action="store_true", help="print per-entry-point GPU time (diagnostic)")
    ap.add_argument("--live-traffic", action="store_true",
                    help="collect roofline.traffic in this run (two rocprofv3 --pmc child passes first; N = 1 only; adds ~2 minutes)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))           # before anything touches a GPU
    # RCCL between processes on this image needs dmabuf IPC; the variable is read when the HIP runtime initialises (under
    # torch.distributed.run nothing has touched a GPU yet at this point)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N` or under torch.distributed.run "
                 "with --nproc-per-node N)" % (args.gpus, world))
    if args.live_traffic and world == 1:
        collect_live_traffic()                                     # child processes; nothing has touched a GPU here yet
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("E2E_BENCH_DRY") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        return dry_rank(args, rank, world, json_fd)
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
    net, opt, mask, fused = build(device, patch)
    x, targets = synthetic_batch(device, patch, args.batch, seed=100 + rank)
    eng = net.engine(x)
    ds_w = np.array([8, 4, 2, 1, 0], dtype=np.float64) / 15.0
    overlap = None
    sample = None

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

    def timed(fn, n):
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for _ in range(args.warmup):
        step()
    kernel_clock(L, 0), kernel_clock(L, 1)                       # reset: the clocks below are those of the timed steps
    dt = timed(step, args.steps)
    clk_mm, clk_wg = kernel_clock(L, 0), kernel_clock(L, 1)       # (outside the timed region)
    rccl = per_rank_ms = None
    if use_dist:
        mine = torch.tensor([dt], dtype=torch.float64, device=device)
        every = torch.zeros(world, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(every, mine)
        per_rank_ms = [float(v) / args.steps * 1e3 for v in every.tolist()]
        dt = max(float(v) for v in every.tolist())               # MAX over ranks
        ones = torch.ones(1, dtype=torch.float32, device=device)
        dist.all_reduce(ones)
        devs = torch.zeros(world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(devs, torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device=device))
        rccl = {"backend": dist.get_backend(), "world": dist.get_world_size(), "allreduce_of_ones": float(ones.item()),
                "devices": torch.cuda.device_count(), "device_per_rank": [int(v) for v in devs.tolist()]}
        # the first multi-GPU run must not be a debugging session: a wrong world size or two ranks on one GPU stop the run here
        assert rccl["allreduce_of_ones"] == float(world), "RCCL all-reduce of ones gave %r on %d ranks" % (rccl["allreduce_of_ones"], world)
        if torch.cuda.device_count() >= world:
            assert len(set(rccl["device_per_rank"])) == world, "ranks share a GPU: %r" % (rccl["device_per_rank"],)
        if not args.forward_only:
            # what the gradient all-reduce costs the step: the same steps with the bucket hook off (every rank trains alone; no
            # collective inside the step), MAX over ranks like the headline; exposed = step with the overlapped all-reduce - that
            hook, eng.grad_bucket_hook = eng.grad_bucket_hook, None

            def local_step():
                eng.forward(x, True)
                eng.loss_backward(targets, ds_w, batch_dice=False)
                fused.step(eng.grads, mask.masks)
                mask.step(masks_already_applied=True)
            n2 = max(3, min(args.steps, 10))
            local_step()
            dt2 = timed(local_step, n2)
            eng.grad_bucket_hook = hook
            mine2 = torch.tensor([dt2], dtype=torch.float64, device=device)
            every2 = torch.zeros(world, dtype=torch.float64, device=device)
            dist.all_gather_into_tensor(every2, mine2)
            ms_local = max(float(v) for v in every2.tolist()) / n2 * 1e3
            rccl["ms_per_step_without_allreduce"] = ms_local
            rccl["allreduce_exposed_ms"] = dt / args.steps * 1e3 - ms_local
            rccl["allreduce_bytes_per_step"] = int(eng.grad_flat.numel()) * 4

    # ---- instrumented steps (outside the timed region): HIP events around the conv entry points --------------------
    # (_splitk: the deep levels' forward / data gradient, same kernel + a sum kernel; _dense: the unmasked layers on the matrix cores)
    names = ["conv133_fwd", "conv133_fwd_splitk", "conv133_fwd_dense", "conv133_fwd_sparse", "conv133_fwd_mm", "conv133_dgrad", "conv133_dgrad_splitk",
             "conv133_dgrad_dense", "conv133_dgrad_sparse", "conv133_dgrad_mm", "conv133_sparse_pack", "conv133_mm_pack",
             "conv133_input_ranges", "conv133_wgrad"]
    if args.op_profile:
        names += ["in_stats_finalize", "in_lrelu_bwd", "convT_fwd", "convT_dgrad", "convT_wgrad", "maxpool_fwd", "maxpool_bwd",
                  "head1x1_fwd", "head1x1_dgrad", "head1x1_wgrad", "dc_ce_reduce", "dc_ce_grad", "grad_sqnorm", "sgd_clip_mask_step"]
    timers = {n: KernelTimer(L, n) for n in names}
    isteps = max(1, min(args.steps, 5))
    # the timed steps above overlap the deep levels and the weight gradients with the full-resolution kernels on extra HIP
    # streams; a pair of events around a launch only brackets THAT kernel when nothing else shares the device, so the
    # instrumented steps issue everything on one stream (the kernels and their inputs are the same)
    from e2enet_medical_amd import engine as _engine
    conc = (_engine.LANES, _engine.WGRAD_STREAM)
    _engine.LANES = _engine.WGRAD_STREAM = False
    for t in timers.values():
        t.enabled = True
    for _ in range(isteps):
        step()
    torch.cuda.synchronize()
    for t in timers.values():
        t.enabled = False
    _engine.LANES, _engine.WGRAD_STREAM = conc

    sw_multi = None
    if world > 1 and not args.no_extras and not args.forward_only:
        # the north_star's multi-GPU path: tile-sharded sliding-window inference with RCCL all-gather (every rank takes part)
        sw_multi = sliding_window_record(device, rank, world)

    if rank == 0:
        vox_per_step = world * args.batch * patch[0] * patch[1] * patch[2]
        value = vox_per_step * args.steps / dt
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "voxels/sec (train step fwd+bwd+update), 128^3 patch 32ch density=0.2",
            "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_NOTE, "data": "synthetic",
            "config": {"workload": "BraTS-shaped 4-modal %d^3 patches, shiftConvPP base 32, K=4, DSFF density 0.2, "
                                   "batch %d per GPU, %s" % (patch[0], args.batch, "inference forward (deep supervision heads off)"
                                                             if args.forward_only else
                                                             "fwd+loss+bwd+clip+SGD+mask step, dense (parity) wgrad"),
                       "parallelism": "dp%d" % world},
            "per_gpu_voxels_per_s": value / world,
            "rccl": rccl, "ms_per_step_per_rank": per_rank_ms,
            "hbm_roofline_frac_whole_step": (value / world) * (FWD_BYTES_PER_VOXEL if args.forward_only else TRAIN_BYTES_PER_VOXEL)
            / (HBM_PEAK_GBS * 1e9),
        }
        if args.forward_only:
            out["metric"] = "voxels/sec (inference forward only), 128^3 patch 32ch density=0.2"
        work = conv_work(eng, mask)
        ev = [(e0.elapsed_time(e1), work[a[0]]) for e0, e1, a in timers["conv133_fwd"].events + timers["conv133_fwd_splitk"].events +
              timers["conv133_fwd_dense"].events + timers["conv133_fwd_sparse"].events + timers["conv133_fwd_mm"].events]
        ev += [(e0.elapsed_time(e1), work[a[4]]) for e0, e1, a in timers["conv133_dgrad_mm"].events]
        ev += [(e0.elapsed_time(e1), work[a[6]]) for e0, e1, a in timers["conv133_dgrad_sparse"].events]
        # weight packing (once per optimizer step for all K1m layers and both directions) and the operand-range words: counted with the family
        pack_ms = timers["conv133_sparse_pack"].total_ms() + timers["conv133_mm_pack"].total_ms() + timers["conv133_input_ranges"].total_ms()
        ev += [(e0.elapsed_time(e1), work[a[3]]) for e0, e1, a in timers["conv133_dgrad"].events + timers["conv133_dgrad_splitk"].events]
        ev += [(e0.elapsed_time(e1), work[a[3]]) for e0, e1, a in timers["conv133_dgrad_dense"].events]
        if ev:
            ms = sum(t for t, _ in ev) + pack_ms
            byt = sum(w["bytes"] for _, w in ev)
            fl = sum(w["flops_live"] for _, w in ev)
            gbs = byt / (ms * 1e-3) / 1e9
            out["roofline"] = {
                "bound": "hbm", "kernel": "conv133_mm_kernel + conv133_kernel (+ conv133_sparse_kernel / conv133_dense_kernel where "
                                          "E2E_CONV_MM=0 or a shape is not served): every launch of e2e_conv133_fwd* and e2e_conv133_dgrad* "
                                          "(depth shift + concat + 1x3x3 conv, forward and data gradient; stride-1 layers of the 16x32 tile "
                                          "class, DSFF-masked or not, as a persistent GEMM on the fp16 matrix pipe with fp32-exact two-piece "
                                          "operands incl. the per-step weight-packing and operand-range launches; strided convs and planes <= 16 wide on the vector walk)",
                "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "traffic": pmc_traffic("conv133_kernel", "conv133_dense_kernel", "conv133_sparse_kernel", "conv133_mm_kernel"),
                "traffic_source": traffic_source(),
                "algorithmic_bytes_per_launch": byt / len(ev), "launches_per_step": len(ev) // isteps,
                "avg_ms": ms / len(ev), "ms_per_step": ms / isteps, "share_of_step": (ms / isteps) / ms_step,
                "timing": "HIP events around each launch in %d instrumented steps issued on ONE stream; the timed steps run the "
                          "deep levels and the weight gradients on two more streams beside these kernels (share_of_step compares "
                          "the serial kernel time with the overlapped step)" % isteps,
                "fma": {"achieved": fl / (ms * 1e-3) / 1e12, "peak": FP32_PEAK_TF, "unit": "TFLOP/s",
                        "frac": fl / (ms * 1e-3) / 1e12 / FP32_PEAK_TF, "flops": "live (DSFF-masked) FLOPs from the kernel maps"},
                "measured_clock_mhz": clk_mm[0] or None,
                # an HBM peak does not move with the shader clock: the same number as frac, kept so that both records carry
                # the key; the compute sub-record (fma) and roofline_secondary are the ones the clock derates
                "frac_at_measured_clock": gbs / HBM_PEAK_GBS,
                "clock_note": CLOCK_NOTE,
            }
            if clk_mm[0]:
                out["roofline"]["fma"]["frac_at_measured_clock"] = out["roofline"]["fma"]["frac"] * NOMINAL_MHZ / clk_mm[0]
        wt = timers["conv133_wgrad"]
        if wt.events:
            ms = wt.total_ms()
            fl = sum(work[a[0]]["flops_dense"] for _, _, a in wt.events)
            tf = fl / (ms * 1e-3) / 1e12
            out["roofline_secondary"] = {
                "bound": "mfma", "kernel": "conv133_wgrad_bf3v5<G, 2> (fp16 MFMA, three fp16 products per fp32 product) / v2 / s2 / smallc "
                                           "(fp32 MFMA) + slab reduce: every launch of e2e_conv133_wgrad (dense weight gradient)",
                "achieved": tf, "peak": BF16_PEAK_TF / 3.0, "unit": "TFLOP/s", "frac": tf / (BF16_PEAK_TF / 3.0),
                "note": "fp32-equivalent dense FLOPs / time; peak = the dense fp16 / bf16 MFMA peak (2500 TFLOP/s) / 3, the rate an fp32 "
                        "product rebuilt from three fp16 products can reach (rounds 3-4: six bf16 products, peak / 6 = 416.7 -- against "
                        "that denominator this is %.3f); against the fp32 MFMA / vector peak (157.3) the fraction is %.3f"
                        % (tf / (BF16_PEAK_TF / 6.0), tf / FP32_PEAK_TF),
                "traffic": pmc_traffic("conv133_wgrad"),
                "algorithmic_bytes_per_launch": sum(work[a[0]]["bytes"] for _, _, a in wt.events) / len(wt.events),
                "launches_per_step": len(wt.events) // isteps,
                "avg_ms": ms / len(wt.events), "ms_per_step": ms / isteps, "share_of_step": (ms / isteps) / ms_step,
                "measured_clock_mhz": clk_wg[0] or None,
                "frac_at_measured_clock": (tf / (BF16_PEAK_TF / 3.0) * NOMINAL_MHZ / clk_wg[0]) if clk_wg[0] else None,
                "clock_note": CLOCK_NOTE}
        if args.op_profile:
            prof = {k: round(t.total_ms() / isteps, 3) for k, t in timers.items() if t.events}
            out["op_ms_per_step"] = dict(sorted(prof.items(), key=lambda kv: -kv[1]))
            launches = []
            for k, t in timers.items():
                per = len(t.events) // isteps
                for i in range(per):                      # same launch index across steps = same layer
                    ms = sum(t.events[s * per + i][0].elapsed_time(t.events[s * per + i][1]) for s in range(isteps)) / isteps
                    ints = [a for a in t.events[i][2] if isinstance(a, int) and 0 < a < 100000]
                    launches.append((round(ms, 3), k, ints[:12]))
            launches.sort(key=lambda v: -v[0])
            out["top_launches"] = launches[:40]
            if os.environ.get("E2E_BENCH_ALL_LAUNCHES"):
                out["all_launches"] = launches
        if world == 1 and not args.no_extras and not args.forward_only:
            for _ in range(2):
                fwd_step()
            fdt = timed(fwd_step, 10)
            fvox = args.batch * patch[0] * patch[1] * patch[2] * 10 / fdt
            out["forward_only"] = {"ms_per_batch": fdt / 10 * 1e3, "voxels_per_s": fvox,
                                   "hbm_roofline_frac": fvox * FWD_BYTES_PER_VOXEL / (HBM_PEAK_GBS * 1e9)}
            mask.truncate_weights()                       # warm-up (allocations)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                mask.truncate_weights()
            torch.cuda.synchronize()
            out["dsff_update"] = {"ms": (time.perf_counter() - t0) / 3 * 1e3, "tensors": len(mask.names),
                                  "what": "Masking.truncate_weights(): kernel L1 + exact k-th value + death on device, one packed D2H, "
                                          "growth draws on the host (Python random), masks and liveness tables re-expanded",
                                  "amortised_ms_per_step_at_update_frequency_1200": (time.perf_counter() - t0) / 3 * 1e3 / 1200}
            sample = parity_sample(net, device, patch, ds_w) if not args.no_cpu_baseline else None
            del eng
            net._engines.clear()
            torch.cuda.empty_cache()
            out["sliding_window"] = sliding_window_record(device)
            out["config3"] = config3_record(device)
            # the width the reference trainer hard-codes (nnUNetTrainer_simple.py:296); algorithmic bytes scale with the channel
            # counts: 16 179 B/voxel at base 32 -> x 48 / 32 (every term of BASELINE.md section 3 is linear in the width except the
            # 4-channel input and the K-channel heads, < 1 %)
            out["width48"] = step_record(device, "BraTS-shaped 4-modal 128^3 patches, shiftConvPP base 48 (the reference trainer's width), "
                                                 "K=4, DSFF density 0.2, batch 2, full training step", (128, 128, 128), 4, 4, 48, 0.2, 2, 5,
                                         bytes_per_voxel=TRAIN_BYTES_PER_VOXEL * 48.0 / 32.0)
            out["config5"] = {"d%s" % dens: step_record(device, "AMOS-shaped 1-modal 128^3 patches, 16 classes, base 32, DSFF density %s, "
                                                                "batch 2, full training step (per-rank workload of BASELINE config 5)" % dens,
                                                        (128, 128, 128), 1, 16, 32, dens, 2, 7) for dens in (0.1, 0.5)}
        if sw_multi is not None:
            out["sliding_window"] = sw_multi
        if not args.no_cpu_baseline and world == 1:      # the CPU port is timed on rank 0 of the single-GPU run only
            if sample is None:
                sample = parity_sample(net, device, patch, ds_w)
            out["cpu_baseline"], out["parity"] = cpu_baseline(sample, args.patch)
            if not args.no_extras:
                out["parity"]["config1"] = config1_parity_record(device)
                out["parity"]["config1_engine_vs_cpu"] = out["parity"]["config1"]["config1_engine_vs_cpu"]
        if world == 1 and not args.no_extras and not args.forward_only and not any(os.environ.get(k) == v for k, v in PURE_FP32_ENV.items()):
            out["pure_fp32"] = pure_fp32_record()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
