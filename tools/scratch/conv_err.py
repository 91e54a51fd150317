#!/usr/bin/env python
"""Diagnostic: error of one conv block (pre-norm y and post-norm z) vs fp64, HIP engine vs torch-CPU fp32."""
import os, sys, math
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
import test_gpu_ops as T
from e2enet_medical_amd.engine import ConvOp
from tests.helpers import seeded_input
torch.set_num_threads(16)
CASES = [(1, [(896, True)], 320, (5, 7, 5), (1, 1, 1)), (1, [(320, True), (320, False), (256, False)], 320, (8, 8, 8), (1, 1, 1)),
         (1, [(160, True)], 64, (20, 28, 20), (1, 1, 1)), (1, [(64, True)], 32, (40, 56, 40), (1, 1, 1)), (1, [(64, True)], 32, (64, 64, 64), (1, 1, 1)),
         (1, [(128, True)], 256, (10, 14, 10), (2, 2, 2)), (1, [(256, True)], 256, (5, 7, 5), (1, 1, 1)), (1, [(64, True)], 128, (20, 28, 20), (2, 2, 2)),
         (1, [(256, True)], 320, (5, 7, 5), (1, 1, 1)), (1, [(320, True)], 320, (4, 4, 4), (2, 2, 2)), (1, [(320, True)], 320, (2, 2, 2), (1, 1, 1)),
         (1, [(320, True), (320, False), (256, False)], 320, (4, 4, 4), (1, 1, 1))]
for (B, src_desc, cout, dims, stride) in CASES:
    srcs = [T._make_act((B, c) + dims, normed, 10 + i) for i, (c, normed) in enumerate(src_desc)]
    cin = sum(c for c, _ in src_desc)
    w = seeded_input((cout, cin, 1, 3, 3), seed=3) * (math.sqrt(2.0) / math.sqrt(cin * 9))
    params = {"blk.conv.weight": w, "blk.conv.bias": seeded_input((cout,), seed=4) * 0.1,
              "blk.instnorm.weight": 1 + 0.2 * seeded_input((cout,), seed=6), "blk.instnorm.bias": 0.2 * seeded_input((cout,), seed=7)}
    e = T._eng_stub(params); e.batch = B
    op = ConvOp(e, "blk", srcs, cout, stride)
    op.forward(); torch.cuda.synchronize()
    xs = oracle.depth_shift(torch.cat([T._act_value(a) for a in srcs], 1))
    y32 = F.conv3d(xs, w, params["blk.conv.bias"], padding=(0, 1, 1), stride=stride)
    y64 = F.conv3d(xs.double(), w.double(), params["blk.conv.bias"].double(), padding=(0, 1, 1), stride=stride)
    z32 = F.leaky_relu(F.instance_norm(y32, weight=params["blk.instnorm.weight"], bias=params["blk.instnorm.bias"], eps=1e-5), 0.01)
    z64 = F.leaky_relu(F.instance_norm(y64, weight=params["blk.instnorm.weight"].double(), bias=params["blk.instnorm.bias"].double(), eps=1e-5), 0.01)
    yg = op.out.data.cpu(); zg = T._act_value(op.out)
    # normalisation applied to the *exact* y with the engine's coefficients: isolates the statistics error
    b, c = yg.shape[:2]
    zg_exact_y = F.leaky_relu(y64 * op.out.scale.cpu().double().view(b, c, 1, 1, 1) + op.out.shift.cpu().double().view(b, c, 1, 1, 1), 0.01)
    rms = lambda t: t.double().pow(2).mean().sqrt().item()
    print("cin %4d cout %3d dims %s s%s | y: gpu rms %.2e max %.2e  cpu rms %.2e max %.2e | z: gpu rms %.2e max %.2e  cpu rms %.2e max %.2e | coeff-only max %.2e" % (
        cin, cout, dims, stride[0], rms(yg.double() - y64), (yg.double() - y64).abs().max().item(), rms(y32.double() - y64), (y32.double() - y64).abs().max().item(),
        rms(zg.double() - z64), (zg.double() - z64).abs().max().item(), rms(z32.double() - z64), (z32.double() - z64).abs().max().item(),
        (zg_exact_y - z64).abs().max().item()))
