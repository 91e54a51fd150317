"""diagnostic: where does the fp16x2 weight gradient produce NaN (case one_tile_runs of test_conv133_wgrad_h2_and_bf3_vs_fp64)?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_gpu_ops as T
from e2enet_medical_amd._lib import lib
from e2enet_medical_amd.engine import ConvOp
for (B, src_desc, cout, dims) in [(1, [(16, False)], 16, (1, 20, 32)), (1, [(16, False)], 16, (2, 20, 32)), (1, [(32, False)], 16, (1, 20, 32)), (1, [(16, False)], 32, (1, 20, 32)), (1, [(16, False)], 16, (1, 32, 32)), (1, [(16, False)], 16, (1, 20, 64))]:
    srcs = [T._make_act((B, c) + dims, normed, 70 + i) for i, (c, normed) in enumerate(src_desc)]
    cin = sum(c for c, _ in src_desc)
    params = {"blk.conv.weight": torch.zeros(cout, cin, 1, 3, 3), "blk.conv.bias": torch.zeros(cout), "blk.instnorm.weight": torch.ones(cout), "blk.instnorm.bias": torch.zeros(cout)}
    e = T._eng_stub(params)
    op = ConvOp(e, "blk", srcs, cout, (1, 1, 1))
    for scale in (1e-7, 1.0):
        dy = T._heavy_tailed((B, cout) + dims, 7, scale).cuda()
        word = T._absmax_word(dy)
        dw = torch.full((cout, cin, 1, 3, 3), float("nan"), device="cuda")
        lib().conv133_wgrad(op.chans.data_ptr(), dy.data_ptr(), dw.data_ptr(), e.wgrad_ws.data_ptr(), B, cin, cout, *dims, 1, 1, 1, word.data_ptr(), 0)
        torch.cuda.synchronize()
        bad = ~torch.isfinite(dw.cpu())
        print(src_desc, cout, dims, "scale", scale, lib().last_kernel(), "non-finite:", int(bad.sum()), "of", bad.numel(),
              "| by out ch:", bad.flatten(1).any(1).nonzero().flatten().tolist()[:40], "| by in ch:", bad.permute(1, 0, 2, 3, 4).flatten(1).any(1).nonzero().flatten().tolist()[:40])
