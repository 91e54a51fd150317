#!/usr/bin/env python
"""Diagnostic: rounding error of the DSFF-masked conv output vs fp64 -- the load-balanced kernel (conv133_sparse.hip, flush every ~72
products) against conv133_kernel (flush every chunk; run with E2E_CONV_SPARSE2=0) and torch-CPU fp32."""
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
CASES = [(2, [(32, True), (32, False)], 32, (16, 64, 64), 0.2), (1, [(64, True), (64, False), (32, False)], 64, (8, 32, 32), 0.2),
         (1, [(128, True), (128, False), (64, False)], 128, (4, 32, 32), 0.2), (2, [(32, True), (32, False)], 32, (16, 64, 64), 0.1),
         (1, [(128, True), (128, False), (64, False)], 128, (4, 32, 32), 0.45)]
for (B, src_desc, cout, dims, dens) in CASES:
    srcs = [T._make_act((B, c) + dims, normed, 10 + i) for i, (c, normed) in enumerate(src_desc)]
    cin = sum(c for c, _ in src_desc)
    km = T._kmask(cout, cin, dens, 5)
    w = seeded_input((cout, cin, 1, 3, 3), seed=3) * (math.sqrt(2.0) / math.sqrt(cin * 9 * dens)) * km.view(cout, cin, 1, 1, 1)
    params = {"blk.conv.weight": w, "blk.conv.bias": seeded_input((cout,), seed=4) * 0.1,
              "blk.instnorm.weight": torch.ones(cout), "blk.instnorm.bias": torch.zeros(cout)}
    e = T._eng_stub(params); e.batch = B
    op = ConvOp(e, "blk", srcs, cout, (1, 1, 1))
    rows = torch.empty(((cout + 3) // 4) * ((cin + 7) // 8), dtype=torch.int32, device=e.device)
    cols = torch.empty(((cin + 3) // 4) * ((cout + 7) // 8), dtype=torch.int32, device=e.device)
    lib().dsff_expand_quads(km.to(e.device).data_ptr(), rows.data_ptr(), cols.data_ptr(), cout, cin, 0)
    op.live, op.live_t, op.density = rows, cols, float(km.float().mean())
    planned = T._plan_and_pack(op, km)
    op.forward(); torch.cuda.synchronize()
    kern = lib().last_kernel()
    xs = oracle.depth_shift(torch.cat([T._act_value(a) for a in srcs], 1))
    y32 = F.conv3d(xs, w, params["blk.conv.bias"], padding=(0, 1, 1))
    y64 = F.conv3d(xs.double(), w.double(), params["blk.conv.bias"].double(), padding=(0, 1, 1))
    yg = op.out.data.cpu()
    rms = lambda t: t.double().pow(2).mean().sqrt().item()
    print("cin %4d cout %3d dims %s d %.2f planned %d flush %s | y rms %.3f | gpu err rms %.3e max %.3e | cpu32 err rms %.3e max %.3e" % (
        cin, cout, dims, dens, planned, op.sp_fwd.flush_every if op.sp_fwd else "-", rms(y64), rms(yg.double() - y64), (yg.double() - y64).abs().max().item(),
        rms(y32.double() - y64), (y32.double() - y64).abs().max().item()))
