#!/usr/bin/env python
"""Random shapes through K1m (conv133_mm_kernel, forward + data gradient) and the fp16 two-piece weight gradient against torch
(the operator test body of tests/test_gpu_ops.py): every tile geometry (W = 32 / 64 / 128 full width, 96 general), ragged channel
blocks, 2 ... 20 chunks, several sources with depth shifts, DSFF densities; run it once with the default grid and once per small
E2E_MM_GRID (long item runs per workgroup: every pipeline transition -- the round-5 destination-record bug needed >= 3 items per
workgroup).  Reading failures: as tools/scratch/fuzz_ops.py (a single-element LeakyReLU-kink flip is not a kernel bug).
   python tools/scratch/fuzz_mm.py [n_cases] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ops as T   # noqa: E402
from e2enet_medical_amd._lib import lib   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = ran = 0
    kinds = {}
    while ran < n:
        B = rng.choice([1, 2])
        nsrc = rng.choice([1, 2, 2, 3, 4])
        srcs = [(rng.choice([8, 16, 20, 24, 32, 33, 40, 48, 64, 70, 96]), rng.random() < 0.6) for _ in range(nsrc)]
        cin = sum(c for c, _ in srcs)
        cout = rng.choice([17, 24, 32, 33, 40, 48, 64, 70, 96, 128, 160])
        W = rng.choice([32, 32, 64, 64, 96, 128])
        H = rng.choice([32, 48, 64]) if W < 128 else rng.choice([32, 48])
        D = rng.choice([2, 3, 5, 6, 9]) if W < 128 else rng.choice([2, 3, 5])
        if cin < 17 or cin > (256 if W == 128 else 320) or cout > (256 if W == 128 else 320):
            continue
        if B * D * H * W * (cin + cout) > 60e6:
            continue
        density = rng.choice([1.0, 0.2, 0.5, 0.1])
        km = T._kmask(cout, cin, density, 5)
        if km is not None and (bool((km.sum(1) == 0).any()) or bool((km.sum(0) == 0).any())):
            continue          # a constant plane: degenerate InstanceNorm (fuzz_ops.py)
        case = (B, srcs, cout, (D, H, W), (1, 1, 1), density)
        ran += 1
        try:
            T.test_conv133_fwd_bwd(case)
            k = (lib().last_kernel() or b"").decode()
        except AssertionError as e:
            bad += 1
            print("FAIL", case, str(e)[:300])
        except Exception as e:
            bad += 1
            print("ERROR", case, repr(e)[:300])
    print("fuzz_mm done: %d cases, %d failures (E2E_MM_GRID=%s)" % (ran, bad, os.environ.get("E2E_MM_GRID", "default")))


if __name__ == "__main__":
    main()
