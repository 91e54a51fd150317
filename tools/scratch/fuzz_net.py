#!/usr/bin/env python
"""Randomised whole-network parity sweep (engine vs CPU oracle: logits, loss, every parameter gradient) on the GPU box.
   python tools/scratch/fuzz_net.py [n_cases] [seed]"""
import os, sys, random
import numpy as np
import torch
from torch import nn
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle
from tests.helpers import closed_form_params, seeded_input, seeded_labels
from e2enet_medical_amd.network_architecture.unetpp_d import Generic_UNetPlusPlus
from e2enet_medical_amd.network_architecture.initialization import InitWeights_He


def one(rng, idx):
    pools = [rng.choice([(2, 2, 2), (2, 2, 2), (1, 2, 2)]) for _ in range(5)]
    stride = [int(np.prod([p[a] for p in pools])) for a in range(3)]
    mult = [rng.choice([1, 1, 2]) if stride[a] >= 16 else rng.choice([1, 2, 3]) for a in range(3)]
    patch = tuple(stride[a] * mult[a] for a in range(3))
    if np.prod(patch) > 160 * 160 * 16 or min(patch[a] // stride[a] for a in range(3)) * 0 + np.prod([patch[a] // stride[a] for a in range(3)]) < 2:
        return None
    cin, base, k = rng.choice([1, 2, 4]), rng.choice([4, 8]), rng.choice([2, 3, 5])
    maxf = rng.choice([16, 24, 32])
    B = rng.choice([1, 2])
    net = Generic_UNetPlusPlus(patch, cin, base, k, 5, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True},
                               nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True},
                               True, False, lambda x: x, InitWeights_He(1e-2), [list(p) for p in pools], None, False, True, True,
                               max_num_features=maxf).cuda()
    shapes = {n: tuple(p.shape) for n, p in net.named_parameters()}
    params = closed_form_params(shapes)
    with torch.no_grad():
        for n, p in net.named_parameters():
            p.copy_(params[n])
    spec = oracle.make_spec(cin, base, k, pools, 2, maxf)
    x = seeded_input((B, cin) + patch, seed=300 + idx)
    eng = net.engine(x.cuda())
    outs = eng.forward(x.cuda(), True)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), k, seed=400 + i) for i, o in enumerate(outs)]
    w = oracle.ds_weights(5)
    loss = eng.loss_backward([t.cuda() for t in targets], w, batch_dice=False)
    leaves = {n: p.clone().requires_grad_(True) for n, p in params.items()}
    ref = oracle.forward(spec, leaves, x)
    ref_loss = oracle.deep_supervision_loss(ref, targets, w, False)
    ref_loss.backward()
    msgs = []
    with torch.no_grad():
        ref64 = oracle.forward(spec, {n: p.double() for n, p in params.items()}, x.double())
    for i, (o, r, r64) in enumerate(zip(outs, ref, ref64)):
        err = float((o.cpu() - r.detach()).abs().max())
        if err > 1e-4:
            # two fp32 evaluations further apart than 1e-4: a kernel problem only if the ENGINE is the one far from exact arithmetic
            e64, c64 = float((o.cpu().double() - r64).abs().max()), float((r.detach().double() - r64).abs().max())
            tag = "noise class" if e64 <= 3.0 * c64 + 1e-5 else "ENGINE FAR FROM fp64"
            msgs.append("logits[%d] %.2e (engine-fp64 %.2e, cpu32-fp64 %.2e: %s)" % (i, err, e64, c64, tag))
    if abs(loss.item() - ref_loss.item()) > 5e-5:
        msgs.append("loss %.2e" % abs(loss.item() - ref_loss.item()))
    worst = (0.0, None)
    for n in shapes:
        rg = leaves[n].grad
        err = (eng.grads[n].cpu() - rg).abs().max().item() / max(1.0, rg.abs().max().item())
        if err > worst[0]:
            worst = (err, n)
    if worst[0] > 2e-3:
        msgs.append("grad %s rel %.2e" % (worst[1], worst[0]))
    print("case %d patch %s pools %s cin %d base %d k %d maxf %d B %d -> %s" % (idx, patch, pools, cin, base, k, maxf, B, "ok (worst grad %.1e)" % worst[0] if not msgs else "FAIL " + "; ".join(msgs)))
    return not msgs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = done = 0
    i = 0
    while done < n:
        i += 1
        try:
            r = one(rng, i)
        except Exception as e:
            print("case %d EXC %r" % (i, e))
            r = False
        if r is None:
            continue
        done += 1
        bad += (not r)
    print("net fuzz done: %d failures of %d" % (bad, done))


if __name__ == "__main__":
    main()
