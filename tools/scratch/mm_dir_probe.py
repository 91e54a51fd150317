"""Round 5 diagnostic: which direction of K1m carries the config-5 gradient error?  The test's same-branch check with K1m in the
forward only, in the backward only, both, neither; the ten worst tensors of each."""
import os
import random
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_gpu_configs as T                                  # noqa: E402
from tests.helpers import seeded_input, seeded_labels, engine_branches   # noqa: E402
import oracle                                                           # noqa: E402
from e2enet_medical_amd import engine as E                               # noqa: E402
from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay   # noqa: E402

dens = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
for fwd, bwd in ((True, False),):
    E.MM_FORWARD, E.MM_BACKWARD = fwd, bwd
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
    masked = {n: p.detach().clone() for n, p in net.named_parameters()}
    spec = oracle.make_spec(1, 32, 16)
    x = seeded_input((1, 1, 64, 64, 64), seed=141)
    eng = net.engine(x.cuda())
    outs = eng.forward(x.cuda(), True)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), 16, seed=150 + i) for i, o in enumerate(outs)]
    w = oracle.ds_weights(5)
    eng.loss_backward([t.cuda() for t in targets], w, batch_dice=False)
    br = engine_branches(eng)
    br.taps = {}
    leaves = {n: p.detach().cpu().double().clone().requires_grad_(True) for n, p in masked.items()}
    ref = oracle.forward(spec, leaves, x.double(), branches=br)
    oracle.deep_supervision_loss(ref, targets, w, False).backward()
    errs = []
    num = den = 0.0
    for n in shapes:
        r = leaves[n].grad
        d = eng.grads[n].cpu().double() - r
        num, den = num + d.pow(2).sum().item(), den + r.pow(2).sum().item()
        if r.norm().item() > 1e-6:
            errs.append((d.norm().item() / r.norm().item(), n, r.norm().item()))
    errs.sort(reverse=True)
    print("== K1m forward %s backward %s: global rel-L2 %.3e" % (fwd, bwd, (num / den) ** 0.5))
    for e, n, nr in errs[:10]:
        print("     %.3e  %-50s |g| %.3e" % (e, n, nr))
    # d loss / d (pre-norm conv output) per block, in the engine's backward order: where does the engine leave the fp64 evaluation?
    print("     per block dy, engine vs fp64 (backward order):")
    for op in reversed(eng.ops):
        if isinstance(op, E.ConvOp) and op.prefix in br.taps and op.out.grad is not None:
            r = br.taps[op.prefix].grad
            d = op.out.grad.cpu().double() - r
            print("       %-36s %3d->%3d mm=%d  rel L2 %.3e  max|d| %.3e of max|dy| %.3e" % (op.prefix, op.cin, op.cout, op.use_mm(), (d.norm() / r.norm().clamp_min(1e-30)).item(),
                                                                              d.abs().max().item(), r.abs().max().item()))
            if (d.norm() / r.norm().clamp_min(1e-30)).item() > 5e-4:
                pc = d.pow(2).sum(dim=(0, 2, 3, 4)).sqrt()
                rc = r.pow(2).sum(dim=(0, 2, 3, 4)).sqrt()
                top = torch.argsort(pc, descending=True)[:4].tolist()
                km = mask.masks.get(op.prefix + ".conv.weight")
                for c in top:
                    live_in = int(km[c].flatten(1).amax(1).sum().item()) if km is not None else -1
                    yc = op.out.data[0, c]
                    print("           channel %3d: |d| %.3e |dy| %.3e  rstd %.3f mean %.4f  live incoming kernels %d  y min/max %.6f %.6f  gamma %.4f beta %.4f scale %.5f shift %.5f"
                          % (c, pc[c].item(), rc[c].item(), op.out.rstd[c].item(), op.out.mean[c].item(), live_in, yc.min().item(), yc.max().item(),
                             masked[op.prefix + ".instnorm.weight"][c].item(), masked[op.prefix + ".instnorm.bias"][c].item(), op.out.scale[c].item(), op.out.shift[c].item()))
                    # branch consistency of this channel: the forward's sign (fma(y, scale, shift)) against K7's (fma(gamma, (y - mean) rstd, beta))
                    y32 = op.out.data[0, c]
                    uf = torch.addcmul(op.out.shift[c], y32, op.out.scale[c])
                    ub = (y32 - op.out.mean[c]) * op.out.rstd[c] * masked[op.prefix + ".instnorm.weight"][c] + masked[op.prefix + ".instnorm.bias"][c]
                    print("                    voxels whose sign differs between the two formulas: %d of %d; |u| < 1e-5: %d"
                          % (int(((uf > 0) != (ub > 0)).sum().item()), y32.numel(), int((uf.abs() < 1e-5).sum().item())))
    del eng, net
    torch.cuda.empty_cache()
