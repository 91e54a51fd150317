import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from tests.helpers import seeded_input, seeded_labels
from tests.test_gpu_net import tiny_net, TINY
SHARE = False
def run(use_loss_graph, extra_warm, between):
    net, shapes, _ = tiny_net()
    xa = seeded_input((2, TINY["cin"]) + TINY["patch"], seed=61).cuda()
    eng = net.engine(xa)
    outs = eng.forward(xa, True)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=70 + i).cuda() for i, o in enumerate(outs)]
    gl = [torch.from_numpy(np.random.RandomState(9 + i).standard_normal(tuple(o.shape)).astype(np.float32)).cuda() * 1e-2 for i, o in enumerate(outs)]
    w = oracle.ds_weights(5)
    eng.loss_backward(targets, w, batch_dice=False)
    if extra_warm:
        eng.forward(xa, True); eng.backward(gl)
    eng.forward(xa, True); eng.loss_backward(targets, w, batch_dice=False); eng.backward(gl)
    ref = {n: g.clone() for n, g in eng.grads.items()}
    xin = xa.clone()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.forward(xin, True); eng.loss_backward(targets, w, batch_dice=False); eng.backward(gl)
    torch.cuda.current_stream().wait_stream(side)
    g_loss, g_explicit = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    if use_loss_graph:
        with torch.cuda.graph(g_loss):
            eng.forward(xin, True); eng.loss_backward(targets, w, batch_dice=False)
    with (torch.cuda.graph(g_explicit, pool=g_loss.pool()) if (use_loss_graph and SHARE) else torch.cuda.graph(g_explicit)):
        eng.forward(xin, True); eng.backward(gl)
    if use_loss_graph:
        g_loss.replay(); torch.cuda.synchronize()
    if between == "sync_only":
        x_ = [eng.grads[n].sum().item() for n in list(ref)[:3]]
    if between == "temps":
        for n in ref:
            (eng.grads[n] - ref[n]).abs().max().item()
    g_explicit.replay(); torch.cuda.synchronize()
    nan = sum(1 for n in ref if torch.isnan(eng.grads[n]).any())
    bad = sum(1 for n in ref if not torch.equal(eng.grads[n], ref[n]))
    print("loss_graph=%s extra_warm=%s between=%s -> nan tensors %d, differing %d" % (use_loss_graph, extra_warm, between, nan, bad))
for sh in (False, True):
    SHARE = sh
    print("share pool", sh)
    run(True, False, "temps")
    run(True, False, "sync_only")
