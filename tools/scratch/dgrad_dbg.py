#!/usr/bin/env python
"""Diagnostic: where does the conv data gradient differ from torch at a given shape?"""
import os, sys, math
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
import test_gpu_ops as T
from e2enet_medical_amd.engine import ConvOp
from e2enet_medical_amd._lib import lib
from tests.helpers import seeded_input
torch.set_num_threads(16)
cases = {"l1": (2, [(64, True), (64, False), (32, False)], 64, (64, 64, 64), (1, 1, 1), 0.2),
         "l1d": (2, [(64, True), (64, False), (32, False)], 64, (64, 64, 64), (1, 1, 1), 1.0),
         "l1b1": (1, [(64, True), (64, False), (32, False)], 64, (64, 64, 64), (1, 1, 1), 0.2),
         "l1s": (2, [(64, True), (64, False), (32, False)], 64, (8, 64, 64), (1, 1, 1), 0.2),
         "l0": (2, [(32, True), (32, False)], 32, (128, 128, 128), (1, 1, 1), 0.2)}
B, src_desc, cout, dims, stride, density = cases[sys.argv[1] if len(sys.argv) > 1 else "l1"]
srcs = [T._make_act((B, c) + dims, normed, 10 + i) for i, (c, normed) in enumerate(src_desc)]
cin = sum(c for c, _ in src_desc)
w = seeded_input((cout, cin, 1, 3, 3), seed=3) * (1.0 / math.sqrt(cin * 9))
km = T._kmask(cout, cin, density, 5)
if km is not None:
    w = w * km.view(cout, cin, 1, 1, 1)
params = {"blk.conv.weight": w, "blk.conv.bias": torch.zeros(cout), "blk.instnorm.weight": torch.ones(cout), "blk.instnorm.bias": torch.zeros(cout)}
e = T._eng_stub(params); e.batch = B
op = ConvOp(e, "blk", srcs, cout, stride)
if km is not None:
    rows = torch.empty(((cout + 3) // 4) * ((cin + 7) // 8), dtype=torch.int32, device=e.device)
    cols = torch.empty(((cin + 3) // 4) * ((cout + 7) // 8), dtype=torch.int32, device=e.device)
    lib().dsff_expand_quads(km.cuda().data_ptr(), rows.data_ptr(), cols.data_ptr(), cout, cin, 0)
    op.live, op.live_t = rows, cols
dy = seeded_input((B, cout) + dims, seed=8)
leaf = [T._act_value(a).requires_grad_(True) for a in srcs]
y = F.conv3d(oracle.depth_shift(torch.cat(leaf, 1)), w, None, stride=stride, padding=(0, 1, 1))
y.backward(dy)
op.out.alloc_grad(); op.plan_backward()
op.out.grad.copy_(dy)
for s in srcs:
    s.grad.fill_(float("nan"))
L = lib()
L.conv133_dgrad(op.out.grad.data_ptr(), e.params["blk.conv.weight"].data_ptr(), op.live_t.data_ptr() if op.live_t is not None else None,
                op.outs.data_ptr(), B, cin, cout, *dims, *stride, 0)
torch.cuda.synchronize()
print("kernel:", L.last_kernel().decode())
c0 = 0
for s, lf in zip(srcs, leaf):
    got = s.grad.cpu()
    err = (got - lf.grad).abs()
    bad = err > 2e-4 * max(1.0, lf.grad.abs().max().item())
    print("source C=%d: max err %.3e (max|g| %.3f) nan %d bad %d of %d" % (got.shape[1], err.nan_to_num(1e9).max().item(), lf.grad.abs().max().item(),
                                                                    int(torch.isnan(got).sum()), int(bad.sum()), bad.numel()))
    if bad.any():
        idx = torch.nonzero(bad)
        for dim, name in enumerate("ncdhw"):
            vals, cnt = torch.unique(idx[:, dim], return_counts=True)
            print("   %s: %d distinct, e.g. %s" % (name, len(vals), list(zip(vals[:12].tolist(), cnt[:12].tolist()))))
        print("   shifts of bad channels:", sorted(set(op.shifts[c0 + int(c)] for c in torch.unique(idx[:, 1]))))
    c0 += got.shape[1]
