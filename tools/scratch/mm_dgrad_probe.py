"""Round 5 diagnostic: config 5 (AMOS-shaped, K = 16, 64^3) through the engine, then every K1m data gradient re-issued in isolation
on the layer's real dy against the vector-walk kernel on the same inputs (known good: same-branch gradient error 3e-5)."""
import os
import random
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_configs import build_net, load_closed_form          # noqa: E402
from tests.helpers import seeded_input, seeded_labels                   # noqa: E402
import oracle                                                           # noqa: E402
from e2enet_medical_amd._lib import lib                                 # noqa: E402
from e2enet_medical_amd.engine import ConvOp, _ptr, _stream              # noqa: E402
from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay   # noqa: E402

dens = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
net = build_net((64, 64, 64), 1, 32, 16, [(2, 2, 2)] * 5)
shapes, params = load_closed_form(net)
opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)


class A:
    adv = False
    fix = False
    update_frequency = 1200
    final_density = 0.05


random.seed(0)
mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 10), growth_mode='random',
               redistribution_mode='none', args=A())
mask.add_module(net, sparse_init='uniform', density=dens)
x = seeded_input((1, 1, 64, 64, 64), seed=141)
eng = net.engine(x.cuda())
outs = eng.forward(x.cuda(), True)
targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), 16, seed=150 + i) for i, o in enumerate(outs)]
eng.loss_backward([t.cuda() for t in targets], oracle.ds_weights(5), batch_dice=False)
torch.cuda.synchronize()
L = lib()
p = eng.params
for op in eng.ops:
    if not isinstance(op, ConvOp) or not op.use_mm() or not op.do_dgrad:
        continue
    o = op.out
    b = o.shape[0]
    di, hi, wi = op.in_dims
    dy = o.grad
    a = dy.abs()
    nz = a[a > 0]
    word = int(op.dy_absmax.item()) & 0xffffffff
    import struct
    wmax = struct.unpack("f", struct.pack("I", word))[0]
    chmax = a.amax(dim=(0, 2, 3, 4))
    print("%-34s %3d->%3d %s | max|dy| %.3e (word %.3e) median %.3e  min nonzero %.3e | channel max: max/min %.1e, channels all-zero %d"
          % (op.prefix, op.cin, op.cout, tuple(op.in_dims), a.max().item(), wmax, nz.median().item() if nz.numel() else 0.0,
             nz.min().item() if nz.numel() else 0.0, (chmax.max() / chmax[chmax > 0].min()).item(), int((chmax == 0).sum())))
    srcs = [s for s in op.sources if s.grad is not None]
    keep = [s.grad.clone() for s in srcs]
    res = {}
    for mode in ("mm", "walk"):
        for s in srcs:
            s.grad.zero_()
        ws = eng.fwd_ws
        if mode == "mm":
            L.conv133_dgrad_mm(dy.data_ptr(), op.dy_absmax.data_ptr(), p[op.w_name].data_ptr(), _ptr(op.live_t), op.outs.data_ptr(),
                               b, op.cin, op.cout, di, hi, wi, ws.data_ptr(), ws.numel() * 4, _stream())
        else:
            L.conv133_dgrad(dy.data_ptr(), p[op.w_name].data_ptr(), _ptr(op.live_t), op.outs.data_ptr(), b, op.cin, op.cout, di, hi, wi,
                            1, 1, 1, _stream())
        torch.cuda.synchronize()
        res[mode] = [s.grad.clone() for s in srcs]
    for s, k in zip(srcs, keep):
        s.grad.copy_(k)
    for s, gm, gw in zip(srcs, res["mm"], res["walk"]):
        d = (gm.double() - gw.double())
        rel = (d.norm() / gw.double().norm().clamp_min(1e-30)).item()
        # where is the error: per channel and per depth slice
        pc = d.pow(2).sum(dim=(0, 2, 3, 4)).sqrt() / gw.double().pow(2).sum(dim=(0, 2, 3, 4)).sqrt().clamp_min(1e-30)
        pd = d.pow(2).sum(dim=(0, 1, 3, 4)).sqrt() / gw.double().pow(2).sum(dim=(0, 1, 3, 4)).sqrt().clamp_min(1e-30)
        print("    -> %-30s rel L2 mm vs walk %.3e | worst channel %.3e (#%d) | worst depth slice %.3e (#%d) | max |d| %.3e at %s"
              % (s.name, rel, pc.max().item(), int(pc.argmax()), pd.max().item(), int(pd.argmax()), d.abs().max().item(),
                 tuple(int(v) for v in torch.unravel_index(d.abs().argmax(), d.shape))))
