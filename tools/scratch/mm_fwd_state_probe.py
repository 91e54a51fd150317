"""Round 5 diagnostic: the forward state (pre-norm y, scale, shift, mean, rstd) every conv block leaves behind, K1m forward against
the other kernels, on config 5."""
import os
import random
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_gpu_configs as T                                  # noqa: E402
from tests.helpers import seeded_input                                  # noqa: E402
from e2enet_medical_amd import engine as E                               # noqa: E402
from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay   # noqa: E402

dens = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
state = {}
for fwd in (True, False):
    E.MM_FORWARD = fwd
    net = T.build_net((64, 64, 64), 1, 32, 16, [(2, 2, 2)] * 5)
    shapes, params = T.load_closed_form(net)
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
    eng.forward(x.cuda(), True)
    torch.cuda.synchronize()
    st = {}
    for op in eng.ops:
        if isinstance(op, E.ConvOp):
            o = op.out
            st[op.prefix] = (op.use_mm(), o.data.double().cpu(), o.scale.double().cpu(), o.shift.double().cpu(), o.mean.double().cpu(), o.rstd.double().cpu(),
                             op.cin, op.cout)
    state[fwd] = st
    del eng, net
    torch.cuda.empty_cache()
for n, (mm, y1, a1, b1, m1, r1, cin, cout) in state[True].items():
    _, y0, a0, b0, m0, r0, _, _ = state[False][n]

    def rel(u, v):
        return ((u - v).norm() / v.norm().clamp_min(1e-30)).item()
    dy = (y1 - y0).abs()
    c = dy.amax(dim=(0, 2, 3, 4))
    print("%-36s %3d->%3d mm=%d | y rel %.2e max %.2e (worst ch %d: %.2e) | scale rel %.2e shift rel %.2e | mean max|d| %.2e rstd rel %.2e (max rstd %.1f, max rel d %.2e)"
          % (n, cin, cout, mm, rel(y1, y0), dy.max().item(), int(c.argmax()), c.max().item(), rel(a1, a0), rel(b1, b0), (m1 - m0).abs().max().item(), rel(r1, r0),
             r0.max().item(), ((r1 - r0).abs() / r0).max().item()))
