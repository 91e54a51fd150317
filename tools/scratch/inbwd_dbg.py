#!/usr/bin/env python
"""Diagnostic: InstanceNorm+LeakyReLU backward (dz -> dy) of the engine vs fp64 on the engine's own y, at full size."""
import os, sys, math
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
import test_gpu_ops as T
from e2enet_medical_amd.engine import ConvOp, LRELU_SLOPE
from e2enet_medical_amd._lib import lib
from tests.helpers import seeded_input
torch.set_num_threads(16)
cases = {"l1": (2, [(64, True), (64, False), (32, False)], 64, (64, 64, 64)),
         "l0": (2, [(32, True), (32, False)], 32, (128, 128, 128)),
         "s": (2, [(32, True), (32, False)], 32, (16, 32, 32))}
B, src_desc, cout, dims = cases[sys.argv[1] if len(sys.argv) > 1 else "l1"]
srcs = [T._make_act((B, c) + dims, normed, 10 + i) for i, (c, normed) in enumerate(src_desc)]
cin = sum(c for c, _ in src_desc)
w = seeded_input((cout, cin, 1, 3, 3), seed=3) * (1.0 / math.sqrt(cin * 9))
params = {"blk.conv.weight": w, "blk.conv.bias": seeded_input((cout,), seed=4) * 0.1,
          "blk.instnorm.weight": 1 + 0.2 * seeded_input((cout,), seed=6), "blk.instnorm.bias": 0.2 * seeded_input((cout,), seed=7)}
e = T._eng_stub(params); e.batch = B
op = ConvOp(e, "blk", srcs, cout, (1, 1, 1))
op.forward()
torch.cuda.synchronize()
yg = op.out.data.cpu()
dz = seeded_input(tuple(yg.shape), seed=8)
g, b = params["blk.instnorm.weight"], params["blk.instnorm.bias"]
y64 = yg.double().requires_grad_(True)
z64 = F.leaky_relu(F.instance_norm(y64, weight=g.double(), bias=b.double(), eps=1e-5), 0.01)
(dy64,) = torch.autograd.grad(z64, y64, dz.double())
y32 = yg.clone().requires_grad_(True)
z32 = F.leaky_relu(F.instance_norm(y32, weight=g, bias=b, eps=1e-5), 0.01)
(dy32,) = torch.autograd.grad(z32, y32, dz)
op.out.alloc_grad()
op.out.grad.copy_(dz)
o = op.out
p, gr = e.params, e.grads
lib().in_lrelu_bwd(o.grad.data_ptr(), o.data.data_ptr(), o.mean.data_ptr(), o.rstd.data_ptr(), p["blk.instnorm.weight"].data_ptr(),
                   p["blk.instnorm.bias"].data_ptr(), LRELU_SLOPE, gr["blk.instnorm.weight"].data_ptr(), gr["blk.instnorm.bias"].data_ptr(),
                   gr["blk.conv.bias"].data_ptr(), e.in_sums.data_ptr(), B, cout, o.spatial, 0)
torch.cuda.synchronize()
dyg = o.grad.cpu()
eg = (dyg.double() - dy64).abs()
ec = (dy32.double() - dy64).abs()
print("dy: gpu max %.3e rms %.3e | cpu32 max %.3e rms %.3e | max|dy| %.3f" % (eg.max().item(), eg.pow(2).mean().sqrt().item(), ec.max().item(), ec.pow(2).mean().sqrt().item(), dy64.abs().max().item()))
per = eg.flatten(2).max(2)[0]
print("per (n,c) max gpu err:", ["%.1e" % v for v in per.flatten()[:16].tolist()])
mean64 = yg.double().flatten(2).mean(2); var64 = yg.double().flatten(2).var(2, unbiased=False)
print("mean err %.3e  rstd rel err %.3e" % ((o.mean.cpu().double().view(B, cout) - mean64).abs().max().item(),
      ((o.rstd.cpu().double().view(B, cout) * (var64 + 1e-5).sqrt()) - 1).abs().max().item()))
# where is the worst element?
idx = torch.nonzero(eg == eg.max())[0].tolist()
n, c = idx[0], idx[1]
u = (yg[n, c].double() - mean64[n, c]) / (var64[n, c] + 1e-5).sqrt() * g[c].double() + b[c].double()
print("worst at", idx, "u there %.3e, dz %.3f, dy64 %.5f dyg %.5f" % (u[idx[2], idx[3], idx[4]].item(), dz[tuple(idx)].item(), dy64[tuple(idx)].item(), dyg[tuple(idx)].item()))
print("elements with |u| < 1e-6 in that (n,c):", int((u.abs() < 1e-6).sum()))
