#!/usr/bin/env python
"""Randomised parity sweep of the conv / transposed-conv operators against torch on the GPU box (diagnostic).
   Reading the output: a reported "wgrad"/"dgrad" failure whose weight gradient is exact when fed a random dy
   (tools/scratch/fuzz_wg.py) is a LeakyReLU-kink flip -- one element with |u| ~ 1e-8 takes the other branch in fp32,
   which shifts the InstanceNorm-backward sums for that (n, c) -- not a kernel bug; all-shifted-out inputs (Cin small,
   D <= 2) make the InstanceNorm degenerate (constant channel) and are not generated.  Real find so far: planes with
   4 rows and wide columns picked the (4, 8, 8) weight-gradient kernel with a (16, 4, 4) tile plan (fixed).
   Masks that leave an output plane without any live kernel are skipped: the plane is constant, its InstanceNorm has
   rstd = 1/sqrt(eps) = 316 and the (analytically zero) conv-bias gradient is amplified rounding noise on both sides.
   python tools/scratch/fuzz_ops.py [n_cases] [seed]"""
import os, sys, math, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa
import test_gpu_ops as T   # reuse the parametrised test bodies


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for i in range(n):
        B = rng.choice([1, 2])
        nsrc = rng.choice([1, 1, 2, 3])
        srcs = [(rng.choice([1, 3, 4, 7, 16, 20, 32, 33, 40, 64, 70]), rng.random() < 0.6) for _ in range(nsrc)]
        cout = rng.choice([5, 8, 24, 32, 40, 64, 70, 128])
        big = rng.random() < 0.6
        if big:
            dims = (rng.choice([5, 6]), rng.choice([17, 20, 24, 32, 40]), rng.choice([32, 36, 40, 64, 68]))     # D >= 5: no all-zero shifted inputs
        else:
            dims = (rng.choice([5, 6, 7]), rng.choice([4, 6, 8, 9, 12, 16]), rng.choice([4, 8, 10, 12, 16, 20]))
        stride = rng.choice([(1, 1, 1)] * 4 + [(2, 2, 2), (1, 2, 2)])
        density = rng.choice([1.0, 0.2, 0.5])
        case = (B, srcs, cout, dims, stride, density)
        km = T._kmask(cout, sum(c for c, _ in srcs), density, 5)
        if km is not None and bool((km.sum(1) == 0).any()):
            continue          # an output plane without a live kernel is a constant channel: degenerate InstanceNorm (rstd = 316)
        try:
            T.test_conv133_fwd_bwd(case)
        except AssertionError as e:
            bad += 1
            print("FAIL conv", case, str(e)[:200])
        except Exception as e:
            bad += 1
            print("ERROR conv", case, repr(e)[:200])
    for i in range(n // 2):
        B = rng.choice([1, 2])
        cin = rng.choice([8, 20, 33, 48, 64, 72, 130])
        cout = rng.choice([5, 16, 32, 40, 64, 70])
        kernel = rng.choice([(2, 2, 2), (2, 2, 2), (1, 2, 2)])
        if rng.random() < 0.5:
            dims = (rng.choice([8, 16]), rng.choice([32, 64]), rng.choice([32, 34, 64]))
        else:
            dims = (rng.choice([1, 2, 3]), rng.choice([3, 4, 9]), rng.choice([4, 6, 8]))
        density = rng.choice([1.0, 0.2, 0.5])
        normed = rng.random() < 0.7
        args = (B, cin, cout, dims, kernel, density, normed)
        try:
            T.test_convT_fwd_bwd(*args)
        except AssertionError as e:
            bad += 1
            print("FAIL convT", args, str(e)[:200])
        except Exception as e:
            bad += 1
            print("ERROR convT", args, repr(e)[:200])
    print("fuzz done: %d failures" % bad)


if __name__ == "__main__":
    main()
