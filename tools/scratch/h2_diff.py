"""diagnostic (round 5): weight gradient of one layer under the bf16x3 kernel and the fp16x2 kernel (E2E_WG_H2=1, subprocesses),
both against an fp64 evaluation on the CPU: error normalised by sum |dy x| per element, relative L2 of the whole tensor."""
import os, sys, subprocess
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
CASES = {"w64": (1, [(33, True), (20, False)], 34, (3, 40, 64)), "w16": (1, [(40, True)], 33, (4, 24, 16))}
if len(sys.argv) > 2:
    import kbench
    from e2enet_medical_amd._lib import lib
    from e2enet_medical_amd.engine import Act, ConvOp
    torch.manual_seed(0)
    B, srcs, cout, dims = CASES[sys.argv[2]]
    dev = torch.device("cuda")
    acts = []
    for i, (c, normed) in enumerate(srcs):
        a = Act("s%d" % i, (B, c) + dims, normed, dev); a.data.normal_()
        if normed:
            a.scale.uniform_(0.5, 2.0); a.shift.normal_()
        acts.append(a)
    cin = sum(c for c, _ in srcs)
    e = kbench.Stub(); e.device = dev
    w = torch.randn(cout, cin, 1, 3, 3, device=dev)
    e.params = {"b.conv.weight": w, "b.conv.bias": torch.zeros(cout, device=dev), "b.instnorm.weight": torch.ones(cout, device=dev), "b.instnorm.bias": torch.zeros(cout, device=dev)}
    e.grads = {k: torch.zeros_like(v) for k, v in e.params.items()}
    op = ConvOp(e, "b", acts, cout, (1, 1, 1))
    e.wgrad_ws = torch.empty(max(op.wgrad_ws_bytes() // 4, 1), dtype=torch.float32, device=dev)
    op.out.alloc_grad(); op.plan_backward(); op.out.grad.normal_()
    lib().conv133_wgrad(op.chans.data_ptr(), op.out.grad.data_ptr(), e.grads["b.conv.weight"].data_ptr(), e.wgrad_ws.data_ptr(), B, cin, cout, *dims, 1, 1, 1, None, 0)
    torch.cuda.synchronize()
    xs = []
    for a in acts:
        x = a.data.double().cpu()
        if a.normed:
            u = x * a.scale.double().cpu().view(B, -1, 1, 1, 1) + a.shift.double().cpu().view(B, -1, 1, 1, 1)
            x = torch.where(u > 0, u, 0.01 * u)
        xs.append(x)
    torch.save({"g": e.grads["b.conv.weight"].cpu(), "x": torch.cat(xs, 1), "dy": op.out.grad.double().cpu(), "kernel": lib().last_kernel() if hasattr(lib(), "last_kernel") else ""}, sys.argv[1])
else:
    for case in CASES:
        res = {}
        for tag, env in (("bf3", {}), ("h2", {"E2E_WG_H2": "1"})):
            subprocess.check_call([sys.executable, __file__, "/tmp/wg_%s.pt" % tag, case], env=dict(os.environ, **env))
            res[tag] = torch.load("/tmp/wg_%s.pt" % tag)
        x, dy = res["bf3"]["x"], res["bf3"]["dy"]
        xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
        H, W = x.shape[-2:]
        ref = torch.zeros_like(res["bf3"]["g"], dtype=torch.float64)
        mag = torch.zeros_like(ref)
        for kh in range(3):
            for kw in range(3):
                xx = xp[..., kh:kh + H, kw:kw + W]
                ref[:, :, 0, kh, kw] = torch.einsum("nodhw,ncdhw->oc", dy, xx)
                mag[:, :, 0, kh, kw] = torch.einsum("nodhw,ncdhw->oc", dy.abs(), xx.abs())
        for tag in ("bf3", "h2"):
            g = res[tag]["g"].double()
            e = (g - ref) / mag
            print("%s %-4s err/sum|ab|: max %.3e rms %.3e mean %+.3e | rel L2 %.3e" % (case, tag, float(e.abs().max()), float(e.pow(2).mean().sqrt()), float(e.mean()), float((g - ref).norm() / ref.norm())))
        print("%s bf3 vs h2 max diff %.3e of max |g| %.3e" % (case, float((res["bf3"]["g"] - res["h2"]["g"]).abs().max()), float(res["bf3"]["g"].abs().max())))
