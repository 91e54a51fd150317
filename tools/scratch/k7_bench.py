#!/usr/bin/env python
"""InstanceNorm + LeakyReLU backward (e2e_in_lrelu_bwd: reduce + apply + params) alone, on the benchmark's tensor shapes.
   python tools/scratch/k7_bench.py        (E2E_LIB_PATH selects an A/B build of instnorm.hip)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from e2enet_medical_amd._lib import lib   # noqa: E402
L = lib()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
for name, (B, C, D) in {"L0 2x32x128^3": (2, 32, 128), "L1 2x64x64^3": (2, 64, 64), "L2 2x128x32^3": (2, 128, 32)}.items():
    sp = D ** 3
    g = torch.Generator(device="cuda").manual_seed(1)
    y = torch.randn((B, C, sp), device=dev, generator=g)
    dz0 = torch.randn((B, C, sp), device=dev, generator=g) * 1e-5
    dz = dz0.clone()
    mean, var = y.mean(2), y.var(2, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    gamma = torch.ones(C, device=dev)
    scale, shift = (rstd * 1.0).contiguous(), (-mean * rstd).contiguous()
    dgamma, dbeta, dbias = (torch.zeros(C, device=dev) for _ in range(3))
    sums = torch.zeros(int(L.in_lrelu_bwd_ws_doubles(B, C)), dtype=torch.float64, device=dev)
    word = torch.zeros(1, dtype=torch.int32, device=dev)

    def run():
        L.in_lrelu_bwd(dz.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), gamma.data_ptr(),
                       0.01, dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(), sums.data_ptr(), B, C, sp, None, 0, word.data_ptr(), st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    byt = 5 * B * C * sp * 4          # reduce: dz + y; apply: dz + y read, dy written
    dz.copy_(dz0)
    run()
    chk = float(dz.double().abs().sum())
    print("%-16s reduce + apply + params %.3f ms   %.0f GB/s of the five passes   checksum %.9e" % (name, ms, byt / ms / 1e6, chk))
