"""one small plan (Hippocampus patch, B = 1) for rocprofv3: 30 fwd+loss+bwd steps"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_net import build_net
from tests.helpers import seeded_input, seeded_labels
import oracle
patch, cin, k, pools, B = (40, 56, 40), 1, 3, [(2, 2, 2)] * 3 + [(1, 1, 1)] * 2, 1
net = build_net(patch, cin, 32, k, pools)
x = seeded_input((B, cin) + patch, seed=1).cuda()
eng = net.engine(x)
outs = eng.forward(x, True)
targets = [seeded_labels((B, 1) + tuple(o.shape[2:]), k, seed=2 + i).cuda() for i, o in enumerate(outs)]
w = oracle.ds_weights(5)
for _ in range(30):
    eng.forward(x, True); eng.loss_backward(targets, w)
torch.cuda.synchronize()
