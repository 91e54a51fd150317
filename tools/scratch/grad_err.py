#!/usr/bin/env python
"""Diagnostic: per-tensor gradient error (relative L2 vs an fp64 evaluation) of the HIP engine and the fp32 CPU oracle."""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from tests.helpers import seeded_input, seeded_labels
from tests.test_gpu_net import build_net, load_closed_form, HIPPO
torch.set_num_threads(16)
tag = sys.argv[1] if len(sys.argv) > 1 else "hippo"
if tag == "hippo":
    patch, cin, base, k, pools = HIPPO["patch"], 1, 32, 3, HIPPO["pools"]
    x = seeded_input((1, cin) + patch, seed=81)
else:
    patch, cin, base, k, pools = (64, 64, 64), 1, 32, 16, [(2, 2, 2)] * 5
    x = seeded_input((1, cin) + patch, seed=141)
net = build_net(patch, cin, base, k, pools)
shapes, params = load_closed_form(net)
spec = oracle.make_spec(cin, base, k, pools)
eng = net.engine(x.cuda())
outs = eng.forward(x.cuda(), True)
targets = [seeded_labels((1, 1) + tuple(o.shape[2:]), k, seed=90 + i) for i, o in enumerate(outs)]
w = oracle.ds_weights(5)
eng.loss_backward([t.cuda() for t in targets], w, batch_dice=False)
def og(dtype):
    leaves = {n: p.detach().to(dtype).clone().requires_grad_(True) for n, p in params.items()}
    ref = oracle.forward(spec, leaves, x.to(dtype))
    oracle.deep_supervision_loss(ref, targets, w, False).backward()
    return leaves
l32, l64 = og(torch.float32), og(torch.float64)
rows = []
for n in shapes:
    g64 = l64[n].grad
    nrm = g64.norm().item() + 1e-30
    rows.append((n, (eng.grads[n].cpu().double() - g64).norm().item() / nrm, (l32[n].grad.double() - g64).norm().item() / nrm, nrm))
rows = [r for r in rows if r[3] > 1e-6]
rows.sort(key=lambda r: -r[1])
for n, eg, ec, nrm in rows[:25]:
    print("%-48s gpu relL2 %.2e | cpu32 relL2 %.2e | ratio %5.1f | norm %.2e" % (n, eg, ec, eg / max(ec, 1e-30), nrm))
import statistics
print("median ratio %.2f" % statistics.median(r[1] / max(r[2], 1e-30) for r in rows if r[3] > 1e-6))
