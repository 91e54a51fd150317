"""Per-block diagnosis of the large-activation nets (tests/test_gpu_configs.py::test_whole_net_with_large_activations...): the pre-norm
conv output of every block, engine vs an fp64 evaluation and the fp32 CPU oracle vs the same.  usage: python tools/scratch/range_diag.py <what>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
from tests.helpers import seeded_input
from tests.test_gpu_net import build_net, load_closed_form

what = sys.argv[1] if len(sys.argv) > 1 else "up1e3"
patch, cin, base, k = (32, 64, 64), 2, 32, 3
pools = [(2, 2, 2)] * 4 + [(1, 2, 2)]
net = build_net(patch, cin, base, k, pools)
shapes, params = load_closed_form(net)
with torch.no_grad():
    for n in shapes:
        if what in ("gamma50", "up1e3") and n.endswith("instnorm.weight"):
            params[n] = params[n] * 50.0
        if what == "up1e3" and n.startswith("up") and n.endswith(".weight"):
            params[n] = params[n] * 1e3
        net.get_parameter(n).copy_(params[n])
spec = oracle.make_spec(cin, base, k, pools)
x = seeded_input((1, cin) + patch, seed=901) * (1e6 if what == "in1e6" else 1.0)
eng = net.engine(x.cuda())
outs = eng.forward(x.cuda(), True)
taps = {}
for dt in (torch.float64, torch.float32):
    br = oracle.Branches()
    br.taps = {}
    with torch.no_grad():
        pass
    leaves = {n: p.to(dt).clone().requires_grad_(True) for n, p in params.items()}     # (taps call retain_grad)
    ref = oracle.forward(spec, leaves, x.to(dt), branches=br)
    taps[dt] = ({kk: v.detach() for kk, v in br.taps.items()}, [r.detach() for r in ref])
t64, r64 = taps[torch.float64]
t32, r32 = taps[torch.float32]
print("%-34s %10s %12s %12s %s" % ("block", "max|y|", "eng/fp64", "cpu32/fp64", "kernel range word"))
for op in eng.ops:
    if not hasattr(op, "prefix"):
        continue
    y64 = t64[op.prefix]
    e = (op.out.data.cpu().double() - y64).norm() / y64.norm()
    c = (t32[op.prefix].double() - y64).norm() / y64.norm()
    word = float(op.x_absmax.view(torch.float32).item()) if op.range_known else float("nan")
    print("%-34s %10.3e %12.3e %12.3e %.3e mm=%s" % (op.prefix, float(y64.abs().max()), float(e), float(c), word, op.use_mm()))
for i, (o, a, b) in enumerate(zip(outs, r64, r32)):
    print("head %d: max|logit| %.3e  engine-fp64 %.3e  cpu32-fp64 %.3e  engine-cpu32 %.3e" % (
        i, float(a.abs().max()), float((o.cpu().double() - a).abs().max()), float((b.double() - a).abs().max()), float((o.cpu() - b).abs().max())))
