#!/usr/bin/env python
"""Diagnostic: who is closer to an fp64 evaluation of the same graph -- the fp32 CPU oracle or the HIP engine?
   python tools/scratch/fp64_check.py [hippo|amos|wgrad]"""
import os, sys, math, random
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle
from oracle import network as onet
from tests.helpers import closed_form_params, seeded_input, seeded_labels
from tests.test_gpu_net import build_net, load_closed_form, HIPPO

torch.set_num_threads(16)


def net_case(tag):
    if tag == "hippo":
        patch, cin, base, k, pools, dens = HIPPO["patch"], 1, 32, 3, HIPPO["pools"], 1.0
        x = seeded_input((1, cin) + patch, seed=81)
    else:
        patch, cin, base, k, pools, dens = (64, 64, 64), 1, 32, 16, [(2, 2, 2)] * 5, 0.5
        x = seeded_input((1, cin) + patch, seed=141)
    net = build_net(patch, cin, base, k, pools)
    shapes, params = load_closed_form(net)
    spec = oracle.make_spec(cin, base, k, pools)
    if dens < 1:
        names = oracle.masked_names(spec)
        random.seed(0)
        masks = oracle.uniform_kernel_masks(shapes, names, dens)
        with torch.no_grad():
            for n in names:
                params[n] = params[n] * masks[n]
                net.get_parameter(n).copy_(params[n])
        net.enable_auto_sparsity(True)
    eng = net.engine(x.cuda())
    outs = [o.cpu() for o in eng.forward(x.cuda(), True)]
    with torch.no_grad():
        ref32 = oracle.forward(spec, params, x)
        p64 = {n: p.double() for n, p in params.items()}
        ref64 = oracle.forward(spec, p64, x.double())
    for i in range(4):
        e_gpu = (outs[i].double() - ref64[i]).abs().max().item()
        e_cpu = (ref32[i].double() - ref64[i]).abs().max().item()
        e_gc = (outs[i] - ref32[i]).abs().max().item()
        print("%s out%d: |gpu-f64| %.2e  |cpu32-f64| %.2e  |gpu-cpu32| %.2e  (max|logit| %.2f)" %
              (tag, i, e_gpu, e_cpu, e_gc, ref64[i].abs().max().item()))


def wgrad_case():
    from tests import test_gpu_ops as T
    from e2enet_medical_amd.engine import ConvOp
    from e2enet_medical_amd._lib import lib
    B, src_desc, cout, dims, stride = 2, [(32, True), (32, False)], 32, (128, 128, 128), (1, 1, 1)
    srcs = [T._make_act((B, c) + dims, normed, 10 + i) for i, (c, normed) in enumerate(src_desc)]
    cin = 64
    w = seeded_input((cout, cin, 1, 3, 3), seed=3) * (1.0 / math.sqrt(cin * 9))
    params = {"blk.conv.weight": w, "blk.conv.bias": torch.zeros(cout), "blk.instnorm.weight": torch.ones(cout),
              "blk.instnorm.bias": torch.zeros(cout)}
    e = T._eng_stub(params); e.batch = B
    op = ConvOp(e, "blk", srcs, cout, stride)
    xs = oracle.depth_shift(torch.cat([T._act_value(a) for a in srcs], 1))         # fp32 values the kernel consumes
    dy = seeded_input((B, cout) + dims, seed=8)
    dw = torch.zeros_like(w, device="cuda")
    lib().conv133_wgrad(op.chans.data_ptr(), dy.cuda().data_ptr(), dw.data_ptr(), e.wgrad_ws.data_ptr(), B, cin, cout, *dims, *stride, None, 0)
    torch.cuda.synchronize()
    print("kernel:", lib().last_kernel().decode())
    dw = dw.cpu()
    wl = w.clone().requires_grad_(True)
    y = F.conv3d(xs, wl, None, stride=stride, padding=(0, 1, 1))
    y.backward(dy)
    cpu = wl.grad
    rng = random.Random(0)
    xp = F.pad(xs, (1, 1, 1, 1)).double()
    dyd = dy.double()
    eg = ec = 0.0
    scale = cpu.abs().max().item()
    for _ in range(24):
        o, c, kh, kw = rng.randrange(cout), rng.randrange(cin), rng.randrange(3), rng.randrange(3)
        ref = (dyd[:, o] * xp[:, c, :, kh:kh + 128, kw:kw + 128]).sum().item()
        eg = max(eg, abs(dw[o, c, 0, kh, kw].item() - ref))
        ec = max(ec, abs(cpu[o, c, 0, kh, kw].item() - ref))
    print("wgrad 64->32 @128^3 B=2: max|gpu-f64| %.3e  max|cpu32-f64| %.3e  max|gpu-cpu32| %.3e  scale %.1f" %
          (eg, ec, (dw - cpu).abs().max().item(), scale))


if __name__ == "__main__":
    for t in (sys.argv[1:] or ["hippo", "amos", "wgrad"]):
        wgrad_case() if t == "wgrad" else net_case(t)
