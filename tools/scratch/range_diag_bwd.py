"""Backward companion of range_diag.py: d loss / d (pre-norm conv output) of every block, engine vs the fp64 oracle evaluated with the
engine's own LeakyReLU / pooling decisions (oracle.Branches), in backward order.  usage: python tools/scratch/range_diag_bwd.py <what>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
from tests.helpers import seeded_input, seeded_labels, engine_branches
from tests.test_gpu_net import build_net, load_closed_form

what = sys.argv[1] if len(sys.argv) > 1 else "up30"
patch, cin, base, k = (32, 64, 64), 2, 32, 3
pools = [(2, 2, 2)] * 4 + [(1, 2, 2)]
net = build_net(patch, cin, base, k, pools)
shapes, params = load_closed_form(net)
with torch.no_grad():
    for n in shapes:
        if what in ("gamma50", "up30", "up1e3") and n.endswith("instnorm.weight"):
            params[n] = params[n] * 50.0
        if what in ("up30", "up1e3") and n.startswith("up") and n.endswith(".weight"):
            params[n] = params[n] * (1e3 if what == "up1e3" else 30.0)
        net.get_parameter(n).copy_(params[n])
spec = oracle.make_spec(cin, base, k, pools)
x = seeded_input((1, cin) + patch, seed=901)
eng = net.engine(x.cuda())
outs = eng.forward(x.cuda(), True)
targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), k, seed=910 + i) for i, o in enumerate(outs)]
w = oracle.ds_weights(5)
eng.loss_backward([t.cuda() for t in targets], w, batch_dice=False)
torch.cuda.synchronize()
br = engine_branches(eng)
br.taps = {}
leaves = {n: p.double().clone().requires_grad_(True) for n, p in params.items()}
ref = oracle.forward(spec, leaves, x.double(), branches=br)
oracle.deep_supervision_loss(ref, targets, w, False).backward()
print("%-34s %12s %12s  %s" % ("block (backward order)", "max|dy|", "eng/fp64", "mm"))
from e2enet_medical_amd.engine import ConvOp, UpOp
for i in eng._bwd_order:
    op = eng.ops[i]
    if isinstance(op, ConvOp):
        g64 = br.taps[op.prefix].grad
        e = (op.out.grad.cpu().double() - g64).norm() / g64.norm()
        print("%-34s %12.3e %12.3e  %s" % (op.prefix, float(g64.abs().max()), float(e), op.use_mm()))
    for n in ([op.w_name] if hasattr(op, "w_name") else []):
        r = leaves[n].grad
        print("     grad %-40s rel-L2 %.3e" % (n, float((eng.grads[n].cpu().double() - r).norm() / r.norm())))
