"""diagnostic: weight gradient of one small layer under E2E_WG_BF3=2 and =4 (subprocesses), error by tap / channel"""
import os, sys, subprocess
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
if len(sys.argv) > 1:
    import kbench
    from e2enet_medical_amd._lib import lib
    torch.manual_seed(0)
    B, srcs, cout, dims = 1, [(33, False)], 34, (3, 40, 64)
    dev = torch.device("cuda")
    from e2enet_medical_amd.engine import Act, ConvOp
    acts = []
    for i, (c, normed) in enumerate(srcs):
        a = Act("s%d" % i, (B, c) + dims, normed, dev); a.data.normal_(); acts.append(a)
    cin = sum(c for c, _ in srcs)
    e = kbench.Stub(); e.device = dev
    w = torch.randn(cout, cin, 1, 3, 3, device=dev)
    e.params = {"b.conv.weight": w, "b.conv.bias": torch.zeros(cout, device=dev), "b.instnorm.weight": torch.ones(cout, device=dev), "b.instnorm.bias": torch.zeros(cout, device=dev)}
    e.grads = {k: torch.zeros_like(v) for k, v in e.params.items()}
    op = ConvOp(e, "b", acts, cout, (1, 1, 1))
    e.wgrad_ws = torch.empty(max(op.wgrad_ws_bytes() // 4, 1), dtype=torch.float32, device=dev)
    op.out.alloc_grad(); op.plan_backward(); op.out.grad.normal_()
    lib().conv133_wgrad(op.chans.data_ptr(), op.out.grad.data_ptr(), e.grads["b.conv.weight"].data_ptr(), e.wgrad_ws.data_ptr(), B, cin, cout, *dims, 1, 1, 1, 0)
    torch.cuda.synchronize()
    torch.save(e.grads["b.conv.weight"].cpu(), sys.argv[1])
else:
    for v in ("2", os.environ.get("WG_NEW", "4")):
        subprocess.check_call([sys.executable, __file__, "/tmp/wg_%s.pt" % v], env=dict(os.environ, E2E_WG_BF3=v))
    a, b = torch.load("/tmp/wg_2.pt"), torch.load("/tmp/wg_%s.pt" % os.environ.get("WG_NEW", "4"))
    d = (a - b).abs()[:, :, 0]
    print("max |v2|", float(a.abs().max()), "max diff", float(d.max()))
    print("by tap (kh, kw):\n", d.amax(dim=(0, 1)))
    print("by out channel:", d.amax(dim=(1, 2, 3)))
    print("by in channel:", d.amax(dim=(0, 2, 3)))
