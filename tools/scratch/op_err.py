#!/usr/bin/env python
"""Diagnostic (round 5): relative L2 error against fp64 of one conv layer's forward, data gradient and weight gradient -- the fp16
two-piece matrix-pipe kernels against the round-4 kernels (E2E_CONV_MM=0) and torch-CPU fp32, on dy of realistic magnitudes."""
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
CASES = [(1, [(32, True), (32, False)], 32, (16, 64, 64), 0.5), (1, [(64, True), (64, False), (32, False)], 64, (8, 32, 32), 0.5), (1, [(32, True)], 32, (8, 64, 128), 1.0)]
for (B, src_desc, cout, dims, density) in CASES:
    srcs = [T._make_act((B, c) + dims, normed, 10 + i) for i, (c, normed) in enumerate(src_desc)]
    cin = sum(c for c, _ in src_desc)
    w = seeded_input((cout, cin, 1, 3, 3), seed=3) * (math.sqrt(2.0) / math.sqrt(cin * 9))
    km = T._kmask(cout, cin, density, 5)
    if km is not None:
        w = w * km.view(cout, cin, 1, 1, 1)
    params = {"blk.conv.weight": w, "blk.conv.bias": seeded_input((cout,), seed=4) * 0.1,
              "blk.instnorm.weight": 1 + 0.2 * seeded_input((cout,), seed=6), "blk.instnorm.bias": 0.2 * seeded_input((cout,), seed=7)}
    e = T._eng_stub(params); e.batch = B
    op = ConvOp(e, "blk", srcs, cout, (1, 1, 1))
    if max(op.dense_ws_bytes, op.mm_ws_bytes) > 0:
        e.fwd_ws = torch.empty(max(op.dense_ws_bytes, op.mm_ws_bytes) // 4, dtype=torch.float32, device=e.device)
    if km is not None:
        rows = torch.empty(((cout + 3) // 4) * ((cin + 7) // 8), dtype=torch.int32, device=e.device)
        cols = torch.empty(((cin + 3) // 4) * ((cout + 7) // 8), dtype=torch.int32, device=e.device)
        lib().dsff_expand_quads(km.to(e.device).data_ptr(), rows.data_ptr(), cols.data_ptr(), cout, cin, 0)
        op.live, op.live_t = rows, cols
        op.density = float(km.float().mean())
        T._plan_and_pack(op, km)
    op.forward(); torch.cuda.synchronize()
    kf = lib().last_kernel()
    xs = oracle.depth_shift(torch.cat([T._act_value(a) for a in srcs], 1))
    y32 = F.conv3d(xs, w, params["blk.conv.bias"], padding=(0, 1, 1))
    y64 = F.conv3d(xs.double(), w.double(), params["blk.conv.bias"].double(), padding=(0, 1, 1))
    rl2 = lambda a, b: ((a.double() - b).norm() / b.norm()).item()
    print("fwd   %3d->%3d %s d=%.1f : engine relL2 %.2e | torch-cpu fp32 %.2e" % (cin, cout, dims, density, rl2(op.out.data.cpu(), y64), rl2(y32, y64)))
    # backward of the conv alone: dy handed over directly (heavy-tailed, 1e-6), max |dy| recorded as e2e_in_lrelu_bwd would
    dy = T._heavy_tailed(tuple(y64.shape), 8, 1e-6)
    for s in srcs:
        s._grad_written = False
    op.out.alloc_grad(); op.plan_backward(); op.out.grad.copy_(dy)
    op.dy_absmax.copy_(T._absmax_word(op.out.grad))
    L = lib()
    di, hi, wi = dims
    ws = getattr(e, "fwd_ws", None)
    for s in srcs:
        s.grad.fill_(float("nan"))
    if op.use_mm():
        L.conv133_dgrad_mm(op.out.grad.data_ptr(), op.dy_absmax.data_ptr(), e.params["blk.conv.weight"].data_ptr(), op.live_t.data_ptr() if op.live_t is not None else None, op.outs.data_ptr(), B, cin, cout, di, hi, wi, ws.data_ptr(), ws.numel() * 4, 0)
    elif op.use_dense():
        L.conv133_dgrad_dense(op.out.grad.data_ptr(), e.params["blk.conv.weight"].data_ptr(), op.live_t.data_ptr() if op.live_t is not None else None, op.outs.data_ptr(), B, cin, cout, di, hi, wi, ws.data_ptr(), ws.numel() * 4, 0)
    elif op.sp_bwd is not None:
        sp = op.sp_bwd
        L.conv133_dgrad_sparse(op.out.grad.data_ptr(), sp.wpk.data_ptr(), sp.quads.data_ptr(), sp.woff.data_ptr(), sp.kmax, sp.pslot.data_ptr(), op._bwd_table().data_ptr(), None, sp.flush_every, B, cin, cout, di, hi, wi, 0)
    else:
        L.conv133_dgrad(op.out.grad.data_ptr(), e.params["blk.conv.weight"].data_ptr(), op.live_t.data_ptr() if op.live_t is not None else None, op.outs.data_ptr(), B, cin, cout, di, hi, wi, 1, 1, 1, 0)
    kd = L.last_kernel()
    L.conv133_wgrad(op.chans.data_ptr(), op.out.grad.data_ptr(), e.grads["blk.conv.weight"].data_ptr(), e.wgrad_ws.data_ptr(), B, cin, cout, di, hi, wi, 1, 1, 1, op.dy_absmax.data_ptr(), 0)
    torch.cuda.synchronize()
    xsrc = torch.cat([T._act_value(a) for a in srcs], 1)
    xl = xsrc.double().requires_grad_(True); wl = w.double().requires_grad_(True)
    F.conv3d(oracle.depth_shift(xl), wl, None, padding=(0, 1, 1)).backward(dy.double())
    x32 = xsrc.clone().requires_grad_(True); w32 = w.clone().requires_grad_(True)
    F.conv3d(oracle.depth_shift(x32), w32, None, padding=(0, 1, 1)).backward(dy)
    got = torch.cat([s.grad.cpu() for s in srcs], 1)
    dxs, d32 = xl.grad, x32.grad
    print("dgrad %s : engine relL2 %.2e | torch-cpu fp32 %.2e   [%s | %s]" % (" " * 22, rl2(got, dxs), rl2(d32, dxs), kf.decode()[:32], kd.decode()[:32]))
    print("wgrad %s : engine relL2 %.2e | torch-cpu fp32 %.2e" % (" " * 22, rl2(e.grads["blk.conv.weight"].cpu(), wl.grad), rl2(w32.grad, wl.grad)))
