#!/usr/bin/env python
"""Diagnostic: per-node error (vs an fp64 evaluation) of the HIP engine and of the fp32 CPU oracle."""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
import test_gpu_ops as T
from tests.helpers import seeded_input
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
eng.forward(x.cuda(), True)
with torch.no_grad():
    _, n32 = oracle.forward(spec, params, x, return_nodes=True)
    _, n64 = oracle.forward(spec, {n: p.double() for n, p in params.items()}, x.double(), return_nodes=True)
rms = lambda t: t.double().pow(2).mean().sqrt().item()
for key in sorted(n64.keys(), key=lambda kj: (kj[0] + kj[1], -kj[0])):
    g = T._act_value(eng.nodes[key]).double()
    print("node L%d j%d %-18s gpu rms %.2e max %.2e | cpu32 rms %.2e max %.2e | ratio %.1f" % (
        key[0], key[1], tuple(n64[key].shape[1:]), rms(g - n64[key]), (g - n64[key]).abs().max().item(),
        rms(n32[key].double() - n64[key]), (n32[key].double() - n64[key]).abs().max().item(),
        rms(g - n64[key]) / max(rms(n32[key].double() - n64[key]), 1e-30)))
