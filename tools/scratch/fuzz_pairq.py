#!/usr/bin/env python
"""K1m launches of two chunks x Q >= 2 out-channel blocks (the shared-staging walk of round 6, E2E_MM_PAIRQ): the data gradient of
layers with 33..200 input and 17..32 output channels (Q = 2..7), the forward of layers with 17..32 input and 33..160 output channels,
every tile geometry, several sources with depth shifts, DSFF densities, accumulate mode;
run with the default grid and with small E2E_MM_GRID (long item runs: every pipeline transition of the skip / request / convert
combinations).   python tools/scratch/fuzz_pairq.py [n_cases] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ops as T   # noqa: E402
from e2enet_medical_amd._lib import lib   # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = ran = 0
while ran < n:
    B = rng.choice([1, 2])
    fwd_kind = rng.random() < 0.3                                   # forward-side sharing: few input, many output channels
    nsrc = rng.choice([1, 1, 2]) if fwd_kind else rng.choice([1, 2, 2, 3, 4, 5, 6])
    srcs = [(rng.choice([8, 11, 16, 17, 20, 24, 32]), rng.random() < 0.6) for _ in range(nsrc)]
    cin = sum(c for c, _ in srcs)
    cout = rng.choice([33, 48, 64, 70, 96, 128, 160]) if fwd_kind else rng.choice([17, 20, 24, 31, 32])
    if (fwd_kind and (cin < 17 or cin > 32)) or (not fwd_kind and (cin < 33 or cin > 200)):
        continue
    W = rng.choice([32, 64, 96, 128])
    H = rng.choice([32, 48, 64]) if W < 128 else rng.choice([32, 48])
    D = rng.choice([1, 2, 3, 5, 6, 9])
    density = rng.choice([1.0, 0.2, 0.5])
    km = T._kmask(cout, cin, density, 5)
    if km is not None and (bool((km.sum(1) == 0).any()) or bool((km.sum(0) == 0).any())):
        continue
    case = (B, srcs, cout, (D, H, W), (1, 1, 1), density)
    ran += 1
    try:
        T.test_conv133_fwd_bwd(case)
    except AssertionError as e:
        bad += 1
        print("FAIL", case, str(e)[:300])
    except Exception as e:
        bad += 1
        print("ERROR", case, repr(e)[:300])
print("fuzz_pairq done: %d cases, %d failures (E2E_MM_GRID=%s, E2E_MM_PAIRQ=%s)" % (ran, bad, os.environ.get("E2E_MM_GRID", "default"),
                                                                                os.environ.get("E2E_MM_PAIRQ", "1")))
