import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import test_gpu_ops as T
import oracle
from e2enet_medical_amd.engine import ConvOp
from e2enet_medical_amd._lib import lib
from tests.helpers import seeded_input
def run(case):
    B, src_desc, cout, dims, stride, density = case
    srcs = [T._make_act((B, c) + dims, normed, 10 + i) for i, (c, normed) in enumerate(src_desc)]
    cin = sum(c for c, _ in src_desc)
    w = seeded_input((cout, cin, 1, 3, 3), seed=3) * (1.0 / math.sqrt(cin * 9))
    params = {"blk.conv.weight": w, "blk.conv.bias": torch.zeros(cout), "blk.instnorm.weight": torch.ones(cout), "blk.instnorm.bias": torch.zeros(cout)}
    e = T._eng_stub(params); e.batch = B
    op = ConvOp(e, "blk", srcs, cout, stride)
    op.forward()
    leaf = [T._act_value(a).requires_grad_(True) for a in srcs]
    wl = w.clone().requires_grad_(True)
    y = F.conv3d(oracle.depth_shift(torch.cat(leaf, 1)), wl, None, stride=stride, padding=(0, 1, 1))
    dy = seeded_input(tuple(y.shape), seed=8)
    y.backward(dy)
    L = lib(); di, hi, wi = dims
    dyd = dy.cuda()
    op.out.alloc_grad(); op.plan_backward()
    dw = torch.zeros_like(w, device="cuda")
    L.conv133_wgrad(op.chans.data_ptr(), dyd.data_ptr(), dw.data_ptr(), e.wgrad_ws.data_ptr(), B, cin, cout, di, hi, wi, *stride, None, 0)
    torch.cuda.synchronize()
    err = (dw.cpu() - wl.grad).abs()
    bad = torch.nonzero(err > 2e-4 * max(1.0, float(wl.grad.abs().max())))
    print(case, "wgrad-only max err %.3e (max|g| %.3e) bad %d" % (float(err.max()), float(wl.grad.abs().max()), bad.shape[0]),
          ("o %d..%d c %d..%d taps %s" % (int(bad[:,0].min()), int(bad[:,0].max()), int(bad[:,1].min()), int(bad[:,1].max()), sorted(set((bad[:,3]*3+bad[:,4]).tolist())))) if bad.shape[0] else "")
for c in [
 (1, [(40, True)], 70, (5, 32, 40), (1, 1, 1), 1.0),
 (2, [(16, True), (33, True)], 64, (6, 40, 68), (2, 2, 2), 0.5),
 (2, [(33, True)], 128, (5, 24, 32), (1, 1, 1), 0.5),
 (2, [(64, False), (32, False), (20, True)], 24, (5, 40, 32), (1, 1, 1), 1.0),
 (2, [(1, True), (20, False), (32, True)], 128, (5, 24, 32), (1, 1, 1), 0.5),
 (2, [(32, False), (64, True), (4, True)], 24, (6, 17, 36), (1, 1, 1), 0.5),
]:
    run(c)
