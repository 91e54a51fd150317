#!/usr/bin/env python
"""Is a training step's backward pass bit-reproducible run to run?  The benchmark's network (128^3, B = 2): the same forward + loss +
backward N times from identical weights; every gradient tensor compared bit for bit with the first run's.
   python tools/scratch/repro_check.py [runs]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench   # noqa: E402
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda")
patch = (128,) * 3
net, opt, mask, fused = bench.build(dev, patch)
x, targets = bench.synthetic_batch(dev, patch, 2, seed=100)
eng = net.engine(x)
ds_w = np.array([8, 4, 2, 1, 0], dtype=np.float64) / 15.0
ref, ref_loss, diff = None, None, {}
for r in range(runs):
    eng.forward(x, True)
    loss = float(eng.loss_backward(targets, ds_w, batch_dice=False).item())
    torch.cuda.synchronize()
    g = {n: t.clone() for n, t in eng.grads.items()}
    if ref is None:
        ref, ref_loss = g, loss
        continue
    if loss != ref_loss:
        diff["<loss>"] = diff.get("<loss>", 0) + 1
    for n in g:
        if not torch.equal(g[n], ref[n]):
            diff[n] = diff.get(n, 0) + 1
print("%d runs against the first: %d tensors ever differed%s" % (runs - 1, len(diff), ": " + ", ".join("%s x%d" % kv for kv in sorted(diff.items())[:12]) if diff else ""))
