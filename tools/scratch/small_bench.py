"""wall time per forward / per training step at a small patch (host-launch-bound?) (diagnostic)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_net import build_net
from tests.helpers import seeded_input, seeded_labels
import oracle
for name, patch, cin, k, pools, B in (("hippocampus", (40, 56, 40), 1, 3, [(2, 2, 2)] * 3 + [(1, 1, 1)] * 2, 9),
                                      ("hippocampus-b1", (40, 56, 40), 1, 3, [(2, 2, 2)] * 3 + [(1, 1, 1)] * 2, 1),
                                      ("64^3", (64, 64, 64), 4, 4, [(2, 2, 2)] * 5, 2)):
    net = build_net(patch, cin, 32, k, pools)
    x = seeded_input((B, cin) + patch, seed=1).cuda()
    eng = net.engine(x)
    outs = eng.forward(x, True)
    targets = [seeded_labels((B, 1) + tuple(o.shape[2:]), k, seed=2 + i).cuda() for i, o in enumerate(outs)]
    w = oracle.ds_weights(5)
    def fwd(): eng.forward(x, False)
    def step(): eng.forward(x, True); eng.loss_backward(targets, w)
    for fn, tag in ((fwd, "forward"), (step, "fwd+loss+bwd")):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); t_issue = time.perf_counter() - t0
        torch.cuda.synchronize(); t1 = time.perf_counter() - t0
        print("%-16s %-13s wall %.2f ms  host issue %.2f ms  gpu span %.2f ms" % (name, tag, t1 / 20 * 1e3, t_issue / 20 * 1e3, e0.elapsed_time(e1) / 20))
