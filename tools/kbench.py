#!/usr/bin/env python
"""Kernel micro-benchmarks on the GPU (diagnostic): times single C-ABI entry points on BASELINE layer shapes and
prints achieved HBM GB/s (algorithmic bytes) and TFLOP/s.   python tools/kbench.py [conv|wgrad|convt|all]"""
import math
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from e2enet_medical_amd._lib import lib          # noqa: E402
from e2enet_medical_amd.engine import Act, ConvOp, UpOp   # noqa: E402


class Stub:
    pass


def time_ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def conv_case(B, srcs, cout, dims, stride, density, tag):
    dev = torch.device("cuda")
    if density < 1.0 and os.environ.get("KB_DENSITY"):
        density = float(os.environ["KB_DENSITY"])
    acts = []
    for i, (c, normed) in enumerate(srcs):
        a = Act("s%d" % i, (B, c) + dims, normed, dev)
        a.data.normal_()
        if normed:
            a.scale.fill_(1.0)
            a.shift.fill_(0.0)
        acts.append(a)
    cin = sum(c for c, _ in srcs)
    w = torch.randn(cout, cin, 1, 3, 3, device=dev) / math.sqrt(cin * 9)
    e = Stub()
    e.device = dev
    e.params = {"b.conv.weight": w, "b.conv.bias": torch.zeros(cout, device=dev),
                "b.instnorm.weight": torch.ones(cout, device=dev), "b.instnorm.bias": torch.zeros(cout, device=dev)}
    e.grads = {k: torch.zeros_like(v) for k, v in e.params.items()}
    op = ConvOp(e, "b", acts, cout, stride)
    e.wgrad_ws = torch.empty(max(op.wgrad_ws_bytes() // 4, 1), dtype=torch.float32, device=dev)
    e.in_sums = torch.zeros(int(lib().in_lrelu_bwd_ws_doubles(B, cout)), dtype=torch.float64, device=dev)
    if density < 1.0:
        km = (torch.rand(cout, cin, device=dev) < density).to(torch.uint8)
        if os.environ.get("KB_BALANCED"):      # diagnostic: exactly round(8*density) live kernels per (row, 8-plane chunk)
            nl = max(1, round(8 * density))
            r = torch.rand(cout, cin // 8, 8, device=dev).argsort(dim=-1)
            km = (r < nl).reshape(cout, cin).to(torch.uint8)
        w.mul_(km.view(cout, cin, 1, 1, 1))
        rows = torch.empty(((cout + 3) // 4) * ((cin + 7) // 8), dtype=torch.int32, device=dev)
        cols = torch.empty(((cin + 3) // 4) * ((cout + 7) // 8), dtype=torch.int32, device=dev)
        lib().dsff_expand_quads(km.data_ptr(), rows.data_ptr(), cols.data_ptr(), cout, cin, 0)
        op.live, op.live_t = rows, cols
    op.out.alloc_grad()
    op.plan_backward()
    op.out.grad.normal_()
    vin = B * cin * dims[0] * dims[1] * dims[2]
    vout = op.out.data.numel()
    L = lib()
    di, hi, wi = dims
    p = e.params

    if op.dense_ws_bytes > 0:
        e.fwd_ws = torch.empty(op.dense_ws_bytes // 4, dtype=torch.float32, device=dev)
    if density < 1.0:
        op.density = float(km.float().mean())
    dense_path = op.use_dense() and os.environ.get("KB_NO_DENSE") is None
    mm_path = op.mm_ws_bytes > 0 and os.environ.get("KB_NO_MM") is None
    if mm_path:
        op.pack_mm_standalone("fb")              # once: the engine packs once per optimizer step, not per launch
        op.set_input_range(8.0)                  # (N(0,1)-ish activations: the scale exponent the benchmarked net derives)

    def fwd_mm():
        L.conv133_fwd_mm(op.chans.data_ptr(), cin, op.wpk_fwd.data_ptr(), op.w_absmax.data_ptr(), p["b.conv.bias"].data_ptr(),
                         op.x_absmax_ptr(), op.out.data.data_ptr(), op.part.data_ptr(), B, cout, di, hi, wi, 0)

    def dgrad_mm():
        L.conv133_dgrad_mm(op.out.grad.data_ptr(), amax.data_ptr(), op.wpk_bwd.data_ptr(), op.w_absmax.data_ptr(), op.outs.data_ptr(),
                           B, cin, cout, di, hi, wi, 0)

    def fwd_dense():
        L.conv133_fwd_dense(op.chans.data_ptr(), cin, w.data_ptr(), p["b.conv.bias"].data_ptr(),
                            op.live.data_ptr() if op.live is not None else None, op.out.data.data_ptr(), op.part.data_ptr(),
                            B, cout, di, hi, wi, e.fwd_ws.data_ptr(), e.fwd_ws.numel() * 4, 0)

    def dgrad_dense():
        L.conv133_dgrad_dense(op.out.grad.data_ptr(), w.data_ptr(), op.live_t.data_ptr() if op.live_t is not None else None, op.outs.data_ptr(), B, cin, cout, di, hi, wi, e.fwd_ws.data_ptr(),
                              e.fwd_ws.numel() * 4, 0)

    def fwd():
        L.conv133_fwd(op.chans.data_ptr(), cin, w.data_ptr(), p["b.conv.bias"].data_ptr(), op.live.data_ptr() if op.live is not None else None,
                      op.out.data.data_ptr(), op.part.data_ptr(), B, cout, di, hi, wi, *stride, 0)

    def dgrad():
        L.conv133_dgrad(op.out.grad.data_ptr(), w.data_ptr(), op.live_t.data_ptr() if op.live_t is not None else None,
                        op.outs.data_ptr(), B, cin, cout, di, hi, wi, *stride, 0)

    amax = op.out.grad.abs().max().reshape(1).view(torch.int32)      # what e2e_in_lrelu_bwd records (KB_NO_ABSMAX: bf16x3 operands)

    def wgrad():
        L.conv133_wgrad(op.chans.data_ptr(), op.out.grad.data_ptr(), e.grads["b.conv.weight"].data_ptr(), e.wgrad_ws.data_ptr(),
                        B, cin, cout, di, hi, wi, *stride, None if os.environ.get("KB_NO_ABSMAX") else amax.data_ptr(), op.x_absmax_ptr(), 0)
    def fwd_planned():
        sp = op.sp_fwd
        L.conv133_fwd_sparse(sp.table.data_ptr(), cin, sp.wpk.data_ptr(), p["b.conv.bias"].data_ptr(), sp.quads.data_ptr(),
                             sp.woff.data_ptr(), sp.kmax, sp.qslot.data_ptr(), sp.flush_every, op.out.data.data_ptr(), op.part.data_ptr(), B, cout, di, hi, wi, 0)

    def dgrad_planned():
        sp = op.sp_bwd
        L.conv133_dgrad_sparse(op.out.grad.data_ptr(), sp.wpk.data_ptr(), sp.quads.data_ptr(), sp.woff.data_ptr(), sp.kmax, sp.pslot.data_ptr(),
                               op._bwd_table().data_ptr(), None, sp.flush_every, B, cin, cout, di, hi, wi, 0)
    dense = 2.0 * 9 * cin * cout * (vout / cout)
    res = {}
    if density < 1.0 and not dense_path and not mm_path and os.environ.get("KB_OLD") is None:      # load-balanced kernel (conv133_sparse.hip)
        from e2enet_medical_amd.engine import pack_sparse_weights
        op.build_sparse_plans(km)
        jobs = op.sparse_jobs()
        if jobs:
            table, nj, mx = pack_sparse_weights(jobs, dev)
            L.conv133_sparse_pack(table.data_ptr(), nj, mx, 0)
            print("%-26s pack   %8.3f ms (%d jobs)" % (tag, time_ms(lambda: L.conv133_sparse_pack(table.data_ptr(), nj, mx, 0)), nj))
            fwd, dgrad = fwd_planned, dgrad_planned
            tag = tag + "[plan]"
    if dense_path:
        fwd, dgrad = fwd_dense, dgrad_dense
        tag = tag + "[dense]"
    if mm_path:
        fwd, dgrad = fwd_mm, dgrad_mm
        tag = tag.replace("[dense]", "") + "[mm]"
    import ctypes
    iters = int(os.environ.get("KB_ITERS", "10"))
    for name, fn, flops in (("fwd", fwd, dense * density), ("dgrad", dgrad, dense * density), ("wgrad", wgrad, dense)):
        mhz = ctypes.c_double(0.0)
        for fam in (0, 1):
            L.diag_kernel_clock(fam, ctypes.byref(mhz), None, 1)          # reset
        ms = time_ms(fn, iters=iters)
        gbs = (vin + vout) * 4 / ms / 1e6
        res[name] = ms
        L.diag_kernel_clock(1 if name == "wgrad" else 0, ctypes.byref(mhz), None, 1)
        clk = ("  %4.0f MHz" % mhz.value) if mhz.value else ""       # shader clock of workgroup 0 (matrix-pipe kernels only)
        print("%-26s %-6s %8.3f ms  %7.1f GB/s(alg)  %6.1f TFLOP/s%s" % (tag, name, ms, gbs, flops / ms / 1e9, clk))
    return res


CASES = {
    "L0_64x32_d05": (2, [(32, True), (32, False)], 32, (128, 128, 128), (1, 1, 1), 0.5),
    "L1_64x64d": (2, [(64, True)], 64, (64, 64, 64), (1, 1, 1), 1.0),
    "L0_64x32": (2, [(32, True), (32, False)], 32, (128, 128, 128), (1, 1, 1), 0.2),
    "L0_64x32_d01": (2, [(32, True), (32, False)], 32, (128, 128, 128), (1, 1, 1), 0.1),
    "L0_64x32_d005": (2, [(32, True), (32, False)], 32, (128, 128, 128), (1, 1, 1), 0.05),
    "L1_160x64_d005": (2, [(64, True), (64, False), (32, False)], 64, (64, 64, 64), (1, 1, 1), 0.05),
    "L0_32x32d": (2, [(32, True)], 32, (128, 128, 128), (1, 1, 1), 1.0),
    "L0_96x32": (2, [(32, True), (32, True), (32, False)], 32, (128, 128, 128), (1, 1, 1), 0.2),
    "L0_128x32": (2, [(32, True), (32, True), (32, True), (32, False)], 32, (128, 128, 128), (1, 1, 1), 0.2),
    "L0_160x32": (2, [(32, True), (32, True), (32, True), (32, True), (32, False)], 32, (128, 128, 128), (1, 1, 1), 0.2),
    "L0_4x32d": (2, [(4, False)], 32, (128, 128, 128), (1, 1, 1), 1.0),
    "L1_160x64": (2, [(64, True), (64, False), (32, False)], 64, (64, 64, 64), (1, 1, 1), 0.2),
    "L1_s2_32x64d": (2, [(32, True)], 64, (128, 128, 128), (2, 2, 2), 1.0),
    "L2_320x128": (2, [(128, True), (128, False), (64, False)], 128, (32, 32, 32), (1, 1, 1), 0.2),
    "L3_640x256": (2, [(256, True), (256, False), (128, False)], 256, (16, 16, 16), (1, 1, 1), 0.2),
    "L4_896x320": (2, [(320, True), (320, False), (256, False)], 320, (8, 8, 8), (1, 1, 1), 0.2),
}

if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a in CASES] or ([] if sys.argv[1:] else list(CASES))
    for k in which:
        conv_case(*CASES[k], k)


def convt_case(B, cin, cout, dims, kernel, density, tag):
    dev = torch.device("cuda")
    src = Act("s", (B, cin) + dims, True, dev)
    src.data.normal_(); src.scale.fill_(1.0); src.shift.fill_(0.0)
    w = torch.randn((cin, cout) + kernel, device=dev) / math.sqrt(cin)
    e = Stub(); e.device = dev; e.params = {"up.weight": w}; e.grads = {"up.weight": torch.zeros_like(w)}
    op = UpOp(e, "up.weight", src, cout, kernel)
    e.wgrad_ws = torch.empty(max(op.wgrad_ws_bytes() // 4, 1), dtype=torch.float32, device=dev)
    if density < 1.0:
        km = (torch.rand(cin, cout, device=dev) < density).to(torch.uint8)
        w.mul_(km.view(cin, cout, 1, 1, 1))
        rows = torch.empty(cin * ((cout + 31) // 32), dtype=torch.int32, device=dev)
        cols = torch.empty(cout * ((cin + 31) // 32), dtype=torch.int32, device=dev)
        lib().dsff_expand(km.data_ptr(), None, rows.data_ptr(), cols.data_ptr(), cin, cout, 1, 0)
        op.live, op.live_t = cols, rows
    op.out.alloc_grad(); op.plan_backward(); op.out.grad.normal_()
    vin, vout = src.data.numel(), op.out.data.numel()
    kt = kernel[0] * kernel[1] * kernel[2]
    dense = 2.0 * cin * cout * kt * (vin / cin)
    L = lib()
    b_, _, d_, h_, w_ = src.shape

    # operand ranges: measured maxima (the engine derives bounds from the parameters); KB_NO_RANGES: bf16 three-piece operands
    dyw = op.out.grad.abs().max().reshape(1).view(torch.int32)
    if not os.environ.get("KB_NO_RANGES"):
        xv = torch.nn.functional.leaky_relu(src.data * src.scale.view(b_, cin, 1, 1, 1) + src.shift.view(b_, cin, 1, 1, 1), 0.01)
        op.set_ranges(float(xv.abs().max()), float(w.abs().max()), 1.0)
        op.dy_word = dyw.data_ptr()

    def wgrad():
        L.convT_wgrad(src.data.data_ptr(), src.scale.data_ptr(), src.shift.data_ptr(), 0.01, op.out.grad.data_ptr(),
                      e.grads["up.weight"].data_ptr(), e.wgrad_ws.data_ptr(), b_, cin, cout, d_, h_, w_, *kernel,
                      op._w(0), getattr(op, "dy_word", None), None, 0)

    def dgrad():
        L.convT_dgrad(op.out.grad.data_ptr(), e.params["up.weight"].data_ptr(), op.live_t.data_ptr() if op.live_t is not None else None,
                      src.grad.data_ptr(), int(os.environ.get("KB_ACC", "0")), b_, cin, cout, d_, h_, w_, *kernel,
                      op._w(1), getattr(op, "dy_word", None), None, 0)
    for name, fn, fl in (("fwd", op.forward, dense * density), ("wgrad", wgrad, dense), ("dgrad", dgrad, dense * density)):
        ms = time_ms(fn)
        print("%-26s %-8s %8.3f ms  %7.1f GB/s(alg)  %6.1f TFLOP/s" % (tag, name, ms, (vin + vout) * 4 / ms / 1e6, fl / ms / 1e9))


CT_CASES = {
    "up_L0_64x32": (2, 64, 32, (64, 64, 64), (2, 2, 2), 0.2),
    "up_L1_128x64": (2, 128, 64, (32, 32, 32), (2, 2, 2), 0.2),
    "up_L2_256x128": (2, 256, 128, (16, 16, 16), (2, 2, 2), 0.2),
    "up_L3_320x256": (2, 320, 256, (8, 8, 8), (2, 2, 2), 0.2),
}
if __name__ == "__main__" and any(a.startswith("up") or a == "convt" for a in sys.argv[1:]):
    for k, v in CT_CASES.items():
        if "convt" in sys.argv[1:] or k in sys.argv[1:]:
            convt_case(*v, k)
