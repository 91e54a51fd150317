#!/usr/bin/env python
"""Measured parity of the engine on BASELINE configs 1 and 5 (and the whole 128^3 network of config 2): for every output
head the largest and the RMS |dlogit| of the engine against (a) the reference's own golden logits, (b) an fp64 evaluation
of the same graph, next to the fp32 CPU oracle's own distance from fp64; loss difference; per-tensor gradient noise.

    python tools/parity_report.py [--out profiles/r05_parity.json] [--tag default] [--net128]

Test infrastructure: imports `oracle` (checker).  Run once per library build (E2E_LIB_PATH selects a diagnostic build);
records are merged into the output file under their tag."""
import argparse
import json
import os
import random
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                                     # noqa: E402
from tests.helpers import golden, seeded_input, seeded_labels, check_grads_same_branches    # noqa: E402
from tests.test_gpu_net import build_net, load_closed_form, HIPPO  # noqa: E402


def head_stats(outs, ref32, ref64, gold=None):
    """per head: engine vs fp64 / vs fp32 oracle / vs golden (where the golden holds that head), cpu32 vs fp64"""
    rows = []
    for i, (o, r32, r64) in enumerate(zip(outs, ref32, ref64)):
        o64 = o.cpu().double()
        d64, d32, c64 = (o64 - r64).abs(), (o64 - r32.double()).abs(), (r32.double() - r64).abs()
        row = {"head": i, "shape": list(o.shape),
               "engine_vs_fp64_max": d64.max().item(), "engine_vs_fp64_rms": d64.pow(2).mean().sqrt().item(),
               "engine_vs_cpu32_max": d32.max().item(), "engine_vs_cpu32_rms": d32.pow(2).mean().sqrt().item(),
               "cpu32_vs_fp64_max": c64.max().item(), "cpu32_vs_fp64_rms": c64.pow(2).mean().sqrt().item()}
        if gold is not None and gold[i] is not None:
            sel, ref = gold[i]
            dg = np.abs(sel(o.cpu().numpy()).astype(np.float64) - ref.astype(np.float64))
            row["engine_vs_golden_max"] = float(dg.max())
            row["engine_vs_golden_rms"] = float(np.sqrt((dg ** 2).mean()))
        rows.append(row)
    return rows


def grad_stats(eng, shapes, leaves32, leaves64):
    l2g, l2c = {}, {}
    for n in shapes:
        r64 = leaves64[n].grad
        nrm = r64.norm().item()
        if nrm <= 1e-6:
            continue
        l2g[n] = (eng.grads[n].cpu().double() - r64).norm().item() / nrm
        l2c[n] = (leaves32[n].grad.double() - r64).norm().item() / nrm
    names = list(l2g)
    num_g = sum((eng.grads[n].cpu().double() - leaves64[n].grad).pow(2).sum().item() for n in names)
    num_c = sum((leaves32[n].grad.double() - leaves64[n].grad).pow(2).sum().item() for n in names)
    den = sum(leaves64[n].grad.pow(2).sum().item() for n in names)
    wg, wc = max(l2g, key=l2g.get), max(l2c, key=l2c.get)
    return {"global_rel_l2_engine": (num_g / den) ** 0.5, "global_rel_l2_cpu32": (num_c / den) ** 0.5,
            "median_rel_l2_engine": statistics.median(l2g.values()), "median_rel_l2_cpu32": statistics.median(l2c.values()),
            "worst_tensor_engine": [wg, l2g[wg]], "worst_tensor_cpu32": [wc, l2c[wc]],
            "tensors_engine_above_3x_cpu_worst": sum(1 for n in names if l2g[n] > 3 * l2c[wc])}


def same_branch_stats(eng, spec, params, x, targets, w, shapes):
    """gradients under the engine's own LeakyReLU / pooling decisions (tests/helpers.py): engine and fp32 CPU oracle vs fp64"""
    glob, worst = check_grads_same_branches(eng, spec, params, x, targets, w, shapes, tol_global=1.0, tol_tensor=1.0)
    return {"global_rel_l2_engine": glob["engine"], "global_rel_l2_cpu32": glob["cpu32"],
            "worst_tensor_engine": [worst["engine"][1], worst["engine"][0]], "worst_tensor_cpu32": [worst["cpu32"][1], worst["cpu32"][0]]}


def oracle_grads(spec, params, x, targets, w, dtype):
    leaves = {n: p.detach().to(dtype).clone().requires_grad_(True) for n, p in params.items()}
    ref = oracle.forward(spec, leaves, x.to(dtype))
    loss = oracle.deep_supervision_loss(ref, targets, w, False)
    loss.backward()
    return leaves, loss, [r.detach() for r in ref]


def config1(B=1):
    g = golden("net_hippo.npz")
    net = build_net(HIPPO["patch"], HIPPO["cin"], 32, HIPPO["k"], HIPPO["pools"])
    shapes, params = load_closed_form(net)
    spec = oracle.make_spec(HIPPO["cin"], 32, HIPPO["k"], HIPPO["pools"])
    x = seeded_input((1, HIPPO["cin"]) + HIPPO["patch"], seed=81)
    if B == 2:
        x = torch.cat([x, seeded_input((1, HIPPO["cin"]) + HIPPO["patch"], seed=82)], 0)
    eng = net.engine(x.cuda())
    outs = eng.forward(x.cuda(), True)
    targets = [seeded_labels((1, 1) + tuple(o.shape[2:]), HIPPO["k"], seed=90 + i) for i, o in enumerate(outs)]
    if B == 2:
        targets = [torch.cat([t, seeded_labels(tuple(t.shape), HIPPO["k"], seed=95 + i)], 0) for i, t in enumerate(targets)]
    w = oracle.ds_weights(5)
    loss = eng.loss_backward([t.cuda() for t in targets], w, batch_dice=False)
    l32, loss32, ref32 = oracle_grads(spec, params, x, targets, w, torch.float32)
    l64, loss64, ref64 = oracle_grads(spec, params, x, targets, w, torch.float64)
    gold = None
    if B == 1:
        gold = [((lambda a: a[:, :, ::2, ::2, ::2]) if i == 0 else (lambda a: a), g["b32_logits%d" % i]) for i in range(len(outs))]
    rec = {"heads": head_stats(outs, ref32, ref64, gold), "loss_engine": loss.item(), "loss_cpu32": loss32.item(),
           "loss_fp64": loss64.item(), "grads": grad_stats(eng, shapes, l32, l64),
           "grads_same_branches": same_branch_stats(eng, spec, params, x, targets, w, shapes)}
    if B == 1:
        rec["loss_golden"] = float(g["loss"])
    return rec


def config5(dens):
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    g = golden("net_amos.npz")
    tag = "d%s" % dens
    net = build_net((64, 64, 64), 1, 32, 16, [(2, 2, 2)] * 5)
    shapes, params = load_closed_form(net)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        update_frequency = 1200
        final_density = 0.05
    random.seed(0)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 10), growth_mode='random',
                   redistribution_mode='none', args=A())
    mask.add_module(net, sparse_init='uniform', density=dens)
    x = seeded_input((1, 1, 64, 64, 64), seed=141)
    eng = net.engine(x.cuda())
    outs = eng.forward(x.cuda(), True)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), 16, seed=150 + i) for i, o in enumerate(outs)]
    w = oracle.ds_weights(5)
    loss = eng.loss_backward([t.cuda() for t in targets], w, batch_dice=False)
    spec = oracle.make_spec(1, 32, 16)
    mp = {n: p.detach().cpu().clone() for n, p in net.named_parameters()}
    l32, loss32, ref32 = oracle_grads(spec, mp, x, targets, w, torch.float32)
    l64, loss64, ref64 = oracle_grads(spec, mp, x, targets, w, torch.float64)
    gold = [((lambda a: a[0, :, 31, ::2, ::2]), g[tag + "_slice_d31"]), None, None, ((lambda a: a), g[tag + "_logits3"])]
    return {"heads": head_stats(outs, ref32, ref64, gold), "loss_engine": loss.item(), "loss_cpu32": loss32.item(),
            "loss_fp64": loss64.item(), "loss_golden": float(g[tag + "_loss"]), "grads": grad_stats(eng, shapes, l32, l64),
            "grads_same_branches": same_branch_stats(eng, spec, mp, x, targets, w, shapes)}


def width48():
    """the reference trainer's width at 64^3 (tests/test_gpu_configs.py::test_width48_whole_net_vs_reference_golden_and_oracle)"""
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    g = golden("net_w48.npz")
    net = build_net((64, 64, 64), 4, 48, 4, [(2, 2, 2)] * 5)
    shapes, params = load_closed_form(net)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        update_frequency = 1200
        final_density = 0.05
    random.seed(0)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 10), growth_mode='random',
                   redistribution_mode='none', args=A())
    mask.add_module(net, sparse_init='uniform', density=0.2)
    x = seeded_input((1, 4, 64, 64, 64), seed=241)
    eng = net.engine(x.cuda())
    outs = eng.forward(x.cuda(), True)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), 4, seed=250 + i) for i, o in enumerate(outs)]
    w = oracle.ds_weights(5)
    loss = eng.loss_backward([t.cuda() for t in targets], w, batch_dice=False)
    spec = oracle.make_spec(4, 48, 4)
    mp = {n: p.detach().cpu().clone() for n, p in net.named_parameters()}
    l32, loss32, ref32 = oracle_grads(spec, mp, x, targets, w, torch.float32)
    l64, loss64, ref64 = oracle_grads(spec, mp, x, targets, w, torch.float64)
    gold = [((lambda a: a[0, :, 31, ::2, ::2]), g["slice_d31"]), None, ((lambda a: a), g["logits2"]), ((lambda a: a), g["logits3"])]
    return {"heads": head_stats(outs, ref32, ref64, gold), "loss_engine": loss.item(), "loss_cpu32": loss32.item(),
            "loss_fp64": loss64.item(), "loss_golden": float(g["loss"]), "grads": grad_stats(eng, shapes, l32, l64),
            "grads_same_branches": same_branch_stats(eng, spec, mp, x, targets, w, shapes)}


def net128():
    """BASELINE config 2 at the benchmarked size, B = 1: forward logits of all four heads and the loss against the fp32
    oracle (the fp64 oracle at this size costs ~1 min on 16 threads; done for the forward only)."""
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda")
    net, opt, mask, fused = bench.build(dev)
    x, targets = bench.synthetic_batch(dev, bench.PATCH, 1, seed=100)
    eng = net.engine(x)
    outs = eng.forward(x, True)
    w = oracle.ds_weights(5)
    loss = eng.loss_backward(targets, w, batch_dice=False)
    spec = oracle.make_spec(bench.CIN, bench.BASE, bench.K, bench.POOLS)
    params = {n: p.detach().cpu().clone() for n, p in net.named_parameters()}
    with torch.no_grad():
        ref32 = oracle.forward(spec, params, x.cpu())
        loss32 = oracle.deep_supervision_loss(ref32, [t.cpu() for t in targets], w, False)
        ref64 = oracle.forward(spec, {n: p.double() for n, p in params.items()}, x.cpu().double())
    shapes = {n: tuple(p.shape) for n, p in net.named_parameters()}
    return {"heads": head_stats(outs, ref32, ref64), "loss_engine": loss.item(), "loss_cpu32": loss32.item(),
            "grads_same_branches": same_branch_stats(eng, spec, params, x.cpu(), [t.cpu() for t in targets], w, shapes)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_parity.json"))
    ap.add_argument("--tag", default="default")
    ap.add_argument("--net128", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(max(1, min(torch.get_num_threads(), 16)))
    from e2enet_medical_amd._lib import lib
    rec = {"library": os.path.basename(lib().path)}
    rec["config1_B1"] = config1(1)
    rec["config1_B2"] = config1(2)
    rec["config5_d0.1"] = config5(0.1)
    rec["config5_d0.5"] = config5(0.5)
    rec["width48"] = width48()
    if args.net128:
        rec["config2_net128_B1"] = net128()
    allrec = {}
    if os.path.exists(args.out):
        allrec = json.load(open(args.out))
    allrec[args.tag] = rec
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(allrec, open(args.out, "w"), indent=1)
    for k, v in rec.items():
        if not isinstance(v, dict):
            continue
        for h in v["heads"]:
            print("%-18s head %d  eng-fp64 max %.2e rms %.2e | eng-cpu32 max %.2e | cpu32-fp64 max %.2e rms %.2e | eng-golden max %s" % (
                k, h["head"], h["engine_vs_fp64_max"], h["engine_vs_fp64_rms"], h["engine_vs_cpu32_max"], h["cpu32_vs_fp64_max"], h["cpu32_vs_fp64_rms"],
                ("%.2e" % h["engine_vs_golden_max"]) if "engine_vs_golden_max" in h else "-"))
        if "grads" in v:
            gr = v["grads"]
            print("%-18s grads: global %.4f (cpu %.4f) median %.4f (cpu %.4f) worst %s %.4f (cpu worst %.4f)" % (
                k, gr["global_rel_l2_engine"], gr["global_rel_l2_cpu32"], gr["median_rel_l2_engine"], gr["median_rel_l2_cpu32"],
                gr["worst_tensor_engine"][0], gr["worst_tensor_engine"][1], gr["worst_tensor_cpu32"][1]))
        if "grads_same_branches" in v:
            gr = v["grads_same_branches"]
            print("%-18s grads under the engine's branch decisions, vs fp64: engine global %.3e worst %.3e (%s) | cpu32 global %.3e worst %.3e" % (
                k, gr["global_rel_l2_engine"], gr["worst_tensor_engine"][1], gr["worst_tensor_engine"][0], gr["global_rel_l2_cpu32"],
                gr["worst_tensor_cpu32"][1]))


if __name__ == "__main__":
    main()
