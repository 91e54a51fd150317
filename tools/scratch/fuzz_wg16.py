"""random shapes for the bf16x3 weight gradient (v5 on 4 x 32 and on 8 x 16 tiles): planes 16..72 wide, ragged rows, channel counts
around the 32-blocks, depth shifts, one or two batch items; prints every case whose error leaves the bar of test_gpu_ops."""
import os, sys, math, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import test_gpu_ops as T
import oracle
from e2enet_medical_amd.engine import ConvOp
from e2enet_medical_amd._lib import lib
from tests.helpers import seeded_input
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
nbad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    B = rng.choice([1, 2])
    nsrc = rng.choice([1, 2, 3])
    src_desc = [(rng.choice([5, 8, 16, 20, 31, 32, 33, 40, 64, 70]), rng.random() < 0.6) for _ in range(nsrc)]
    cout = rng.choice([5, 24, 31, 32, 33, 40, 64, 70])
    dims = (rng.choice([1, 2, 3, 5, 9]), rng.choice([8, 9, 12, 16, 17, 20, 24, 33]), rng.choice([16, 20, 24, 28, 32, 36, 40, 64, 72]))
    case = (B, src_desc, cout, dims, (1, 1, 1), 1.0)
    srcs = [T._make_act((B, c) + dims, normed, 10 + i) for i, (c, normed) in enumerate(src_desc)]
    cin = sum(c for c, _ in src_desc)
    w = seeded_input((cout, cin, 1, 3, 3), seed=3) * (1.0 / math.sqrt(cin * 9))
    params = {"blk.conv.weight": w, "blk.conv.bias": torch.zeros(cout), "blk.instnorm.weight": torch.ones(cout), "blk.instnorm.bias": torch.zeros(cout)}
    e = T._eng_stub(params); e.batch = B
    op = ConvOp(e, "blk", srcs, cout, (1, 1, 1))
    op.forward()
    leaf = [T._act_value(a).requires_grad_(True) for a in srcs]
    wl = w.clone().requires_grad_(True)
    y = F.conv3d(oracle.depth_shift(torch.cat(leaf, 1)), wl, None, stride=1, padding=(0, 1, 1))
    dy = seeded_input(tuple(y.shape), seed=8)
    y.backward(dy)
    L = lib(); di, hi, wi = dims
    dyd = dy.cuda()
    op.out.alloc_grad(); op.plan_backward()
    dw = torch.zeros_like(w, device="cuda")
    L.conv133_wgrad(op.chans.data_ptr(), dyd.data_ptr(), dw.data_ptr(), e.wgrad_ws.data_ptr(), B, cin, cout, di, hi, wi, 1, 1, 1, None, 0)
    torch.cuda.synchronize()
    kern = (L.last_kernel() or b"").decode()
    err = float((dw.cpu() - wl.grad).abs().max()); scale = max(1.0, float(wl.grad.abs().max()))
    ok = err < 2e-4 * scale
    nbad += not ok
    print("%s %-60s %-40s err %.2e scale %.2e" % ("ok " if ok else "BAD", str(case[:4]), kern[:40], err, scale))
print("bad cases:", nbad)
