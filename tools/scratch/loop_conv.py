"""run one conv kernel back to back for ~20 s (clock / power observation with rocm-smi from another shell)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.kbench as kb
real_time_ms = kb.time_ms
def long_time(fn, iters=10, warm=3):
    t0 = time.time()
    n = 0
    while time.time() - t0 < 8.0:
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        n += 200
    return real_time_ms(fn, iters, warm)
kb.time_ms = long_time
kb.conv_case(*kb.CASES["L0_64x32"], "L0_64x32")
