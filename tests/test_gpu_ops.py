"""Op-level parity of the HIP kernels (through the C ABI) against the CPU oracle on seeded inputs.
Tolerances: fp32 paths compare at 1e-4-ish absolute on O(1) data; index / mask / integer results are bit exact."""
import math
import os
import random
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from tests.helpers import seeded_input, golden, closed_form_tensor

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _eng_stub(params):
    """Minimal stand-in for engine.Engine used to drive single ops."""
    class E:
        pass
    e = E()
    e.device = _dev()
    e.params = {k: v.to(e.device) for k, v in params.items()}
    e.grads = {k: torch.zeros_like(v) for k, v in e.params.items()}
    e.wgrad_ws = torch.empty(64 << 20, dtype=torch.float32, device=e.device)   # 256 MiB
    e.in_sums = torch.zeros(1024 * (3 + 768) + 2, dtype=torch.float64, device=e.device)   # e2e_in_lrelu_bwd_ws_doubles for B * C <= 1024, zeroed once
    e.batch = 1
    return e


def _make_act(shape, normed, seed):
    from e2enet_medical_amd.engine import Act
    a = Act("t", shape, normed, _dev())
    a.data.copy_(seeded_input(shape, seed=seed))
    if normed:
        n = shape[0] * shape[1]
        a.scale.copy_(0.5 + torch.rand(n, generator=torch.Generator().manual_seed(seed + 1)))
        a.scale[::3] *= -1.0                                  # negative gamma happens in trained nets
        a.shift.copy_(seeded_input((n,), seed=seed + 2) * 0.3)
    return a


def _act_value(a):
    """what consumers see: lrelu(scale*x+shift) for normed tensors, x otherwise (CPU tensor)."""
    x = a.data.cpu()
    if not a.normed:
        return x
    b, c = x.shape[:2]
    s = a.scale.cpu().view(b, c, 1, 1, 1)
    t = a.shift.cpu().view(b, c, 1, 1, 1)
    return F.leaky_relu(x * s + t, 0.01)


CONV_CASES = [
    # (B, sources [(C, normed)], Cout, (D,H,W), stride, density)
    (1, [(4, False)], 8, (5, 12, 20), (1, 1, 1), 1.0),
    (2, [(7, True), (6, False)], 9, (4, 9, 11), (1, 1, 1), 1.0),            # odd sizes, ragged channel counts
    (1, [(16, True), (16, False), (8, False)], 40, (6, 33, 35), (1, 1, 1), 0.3),   # 3-source concat, DSFF bits
    (1, [(1, False)], 5, (6, 8, 8), (1, 1, 1), 1.0),                         # single group, s = -2
    (1, [(12, True)], 20, (8, 16, 16), (2, 2, 2), 1.0),                      # strided ("convolutional pooling")
    (2, [(10, True)], 12, (5, 18, 10), (1, 2, 2), 1.0),
    (1, [(33, True)], 34, (3, 40, 64), (1, 1, 1), 0.2),
    (1, [(70, True), (30, False)], 64, (2, 8, 8), (1, 1, 1), 0.5),           # small planes (8x8 tile kernel)
    (1, [(40, True)], 33, (4, 4, 4), (2, 2, 2), 1.0),
    (1, [(9, True)], 7, (1, 6, 5), (1, 1, 1), 1.0),                          # one-slice volume: all shifted groups vanish
    (1, [(20, True), (15, False)], 40, (6, 32, 64), (2, 2, 2), 1.0),         # strided, wide planes (pipelined s2 wgrad)
    (2, [(33, True)], 34, (3, 24, 40), (1, 2, 2), 1.0),                      # in-plane stride only, ragged tiles
    (2, [(50, True), (40, False)], 70, (2, 20, 36), (1, 1, 1), 0.2),         # double-buffered wgrad, 64-out x 32-in blocks
    (2, [(3, False)], 40, (3, 20, 36), (1, 1, 1), 1.0),                      # input layer: small-Cin wgrad (channel x tap columns)
    (1, [(4, True)], 32, (2, 16, 64), (1, 1, 1), 1.0),
    (2, [(20, True)], 24, (2, 20, 36), (1, 1, 1), 1.0),                      # one 32 x 32 block: row-split double-buffered wgrad
    (2, [(33, True)], 40, (5, 4, 12), (1, 1, 1), 0.5),                       # 4-row planes with wide rows (found by tools/scratch/fuzz_ops.py)
    # load-balanced DSFF kernel (conv133_sparse.hip: stride 1, W % 4 == 0, H, W > 16, > 8 channels both sides, density < 0.5)
    (2, [(32, True), (32, False)], 32, (5, 32, 64), (1, 1, 1), 0.2),           # the benchmark's layer shape in small: whole tiles
    (1, [(20, True), (13, False)], 34, (4, 20, 36), (1, 1, 1), 0.25),          # ragged tiles, 33 input planes (5 chunks), two output groups (32 + 2)
    (2, [(64, True), (64, False), (32, False)], 64, (3, 24, 40), (1, 1, 1), 0.2),   # 160 -> 64: 20 chunks, two full groups
    (1, [(9, False)], 40, (2, 17, 20), (1, 1, 1), 0.4),                        # two chunks, the second nearly empty; 17-row planes
    # planes 16..31 voxels wide: weight gradient on 8 x 16-pixel tiles of the bf16x3 kernel (conv133_wgrad_bf3v5_kernel<1>)
    (2, [(20, True), (13, False)], 34, (3, 9, 20), (1, 1, 1), 1.0),          # ragged tile rows and columns, 33 + 34 channels, depth shifts
    (1, [(40, True)], 33, (2, 16, 16), (1, 1, 1), 0.3),                      # whole tiles, two channel blocks each side
    (1, [(33, True)], 34, (3, 24, 28), (1, 1, 1), 1.0),                      # 28-wide planes: second tile column ragged
    # matrix-pipe paths (conv133_mm.hip: stride 1, W % 32 == 0, H % 16 == 0, 17..320 channels, any density; with E2E_CONV_MM=0
    # conv133_dense.hip where dense or density >= 0.5)
    (1, [(32, True)], 32, (6, 32, 64), (1, 1, 1), 1.0),
    (2, [(20, True), (28, False)], 40, (3, 32, 32), (1, 1, 1), 1.0),         # ragged channel blocks (48 -> 40), two sources, shift
    (1, [(16, True), (16, False), (8, False)], 64, (7, 48, 96), (1, 1, 1), 0.6),   # DSFF map dense enough for the dense kernel
    (2, [(40, True), (30, False)], 70, (3, 32, 64), (1, 1, 1), 0.2),         # 70 -> 70: ragged chunks and out-channel blocks both ways, depth shifts, d = 0.2
    (1, [(100, True), (100, False), (100, False), (20, False)], 33, (2, 32, 32), (1, 1, 1), 0.1),   # 320 -> 33: twenty chunks, one tile per slice
    (1, [(24, True)], 24, (9, 48, 32), (1, 1, 1), 1.0),                      # more items than a workgroup's first round on a small grid (E2E_MM_GRID)
]


def _plan_and_pack(op, km):
    """plans + packed weights of the load-balanced DSFF kernel for a single op driven outside an Engine"""
    from e2enet_medical_amd.engine import pack_sparse_weights
    from e2enet_medical_amd._lib import lib
    op.build_sparse_plans(km)
    jobs = op.sparse_jobs()
    if jobs:
        table, n, mx = pack_sparse_weights(jobs, op.eng.device)
        lib().conv133_sparse_pack(table.data_ptr(), n, mx, 0)
        torch.cuda.synchronize()
    return bool(jobs)


def _ref_conv(srcs, w, b, stride):
    x = torch.cat([_act_value(a) for a in srcs], 1)
    return F.conv3d(oracle.depth_shift(x), w, b, stride=stride, padding=(0, 1, 1))


def _kmask(cout, cin, density, seed):
    if density >= 1.0:
        return None
    g = torch.Generator().manual_seed(seed)
    return (torch.rand((cout, cin), generator=g) < density).to(torch.uint8)


def _assert_wgrad(got, ref32, leaf_srcs, y, gamma, beta, dz, stride, scale_w):
    """Weight gradient against torch-CPU fp32 at 2e-4 of its scale.  At the benchmarked shapes a gradient entry is a sum
    over millions of voxels behind an InstanceNorm backward whose own sums run over millions of voxels, and torch's fp32
    reductions are ~1e-5 of the scale away from exact arithmetic -- further than the kernels (fp64 InstanceNorm sums,
    chunked MFMA sums; tools/scratch/fp64_check.py).  There a seeded sample of entries is recomputed in fp64 (InstanceNorm
    + LeakyReLU backward and the voxel sum), and the kernel must be within 2e-5 of the scale of THAT, or no worse than
    torch's fp32."""
    err = (got - ref32).abs().max().item()
    voxels = dz.numel() // dz.shape[1]
    if voxels < (1 << 18):
        assert err < 2e-4 * scale_w, "wgrad"
        return
    assert err < 2e-3 * scale_w, "wgrad (gross)"
    y64 = y.double().requires_grad_(True)
    z64 = F.leaky_relu(F.instance_norm(y64, weight=gamma.double(), bias=beta.double(), eps=1e-5), 0.01)
    (dyd,) = torch.autograd.grad(z64, y64, dz.double())
    del z64, y64
    sd, sh, sw = stride
    xs = oracle.depth_shift(torch.cat([l.detach() for l in leaf_srcs], 1))
    xp = F.pad(xs, (1, 1, 1, 1))
    Do, Ho, Wo = dyd.shape[2:]
    rng = random.Random(1234)
    e_got = e_ref = 0.0
    for _ in range(24):
        o, c, kh, kw = rng.randrange(got.shape[0]), rng.randrange(got.shape[1]), rng.randrange(3), rng.randrange(3)
        win = xp[:, c, ::sd][:, :Do, kh:kh + sh * Ho:sh, kw:kw + sw * Wo:sw].double()
        exact = (dyd[:, o] * win).sum().item()
        e_got = max(e_got, abs(got[o, c, 0, kh, kw].item() - exact))
        e_ref = max(e_ref, abs(ref32[o, c, 0, kh, kw].item() - exact))
    assert e_got <= max(2e-5 * scale_w, 1.5 * e_ref), "wgrad vs fp64: kernel %.3e, torch fp32 %.3e, scale %.3e" % (e_got, e_ref, scale_w)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv133_fwd_bwd(case):
    from e2enet_medical_amd.engine import ConvOp, shift_amounts
    from e2enet_medical_amd._lib import lib
    B, src_desc, cout, dims, stride, density = case
    srcs = [_make_act((B, c) + dims, normed, 10 + i) for i, (c, normed) in enumerate(src_desc)]
    cin = sum(c for c, _ in src_desc)
    assert shift_amounts(cin) == oracle.shift_amounts(cin)
    w = seeded_input((cout, cin, 1, 3, 3), seed=3) * (1.0 / math.sqrt(cin * 9))
    km = _kmask(cout, cin, density, 5)
    if km is not None:
        w = w * km.view(cout, cin, 1, 1, 1)
    params = {"blk.conv.weight": w, "blk.conv.bias": seeded_input((cout,), seed=4) * 0.1,
              "blk.instnorm.weight": 1 + 0.2 * seeded_input((cout,), seed=6),
              "blk.instnorm.bias": 0.2 * seeded_input((cout,), seed=7)}
    e = _eng_stub(params)
    e.batch = B
    op = ConvOp(e, "blk", srcs, cout, stride)
    if op.dense_ws_bytes > 0:
        e.fwd_ws = torch.empty(op.dense_ws_bytes // 4, dtype=torch.float32, device=e.device)
    if km is not None:
        rows = torch.empty(((cout + 3) // 4) * ((cin + 7) // 8), dtype=torch.int32, device=e.device)
        cols = torch.empty(((cin + 3) // 4) * ((cout + 7) // 8), dtype=torch.int32, device=e.device)
        kmd = km.to(e.device)
        lib().dsff_expand_quads(kmd.data_ptr(), rows.data_ptr(), cols.data_ptr(), cout, cin, 0)
        op.live, op.live_t = rows, cols
        op.density = float(km.float().mean())
        _plan_and_pack(op, km)                      # load-balanced kernel where the shape is served (conv133_sparse.hip)
    op.set_input_range(max(float(_act_value(a).abs().max()) for a in srcs))      # (inside an Engine: e2e_conv133_input_ranges)
    tile_class = dims[2] % 32 == 0 and dims[1] % 16 == 0 and dims[1] > 16 and stride == (1, 1, 1)
    from e2enet_medical_amd.engine import MM_MIN_DENSITY
    if tile_class and cin > 16 and cout > 16 and os.environ.get("E2E_CONV_MM", "1") != "0" and (km is None or op.density >= MM_MIN_DENSITY):
        assert op.use_mm(), "this case is meant to reach conv133_mm_kernel"
    elif tile_class and cin > 16 and cout > 16 and km is not None and op.density < MM_MIN_DENSITY and op.sp_fwd is not None:
        assert not op.use_mm("f") and not op.use_mm("b"), "below the switch-over density a masked layer runs on the sparse walk (round 6)"
    elif tile_class and cin >= 16 and cout >= 16 and density >= 0.5 and os.environ.get("E2E_CONV_DENSE", "1") != "0":
        assert op.use_dense(), "this case is meant to reach conv133_dense_kernel"
    op.forward()
    torch.cuda.synchronize()
    # ---- forward: pre-norm output and the normalise-on-load coefficients
    leaf_srcs = [_act_value(a).requires_grad_(True) for a in srcs]
    wl = params["blk.conv.weight"].clone().requires_grad_(True)
    bl = params["blk.conv.bias"].clone().requires_grad_(True)
    gl = params["blk.instnorm.weight"].clone().requires_grad_(True)
    tl = params["blk.instnorm.bias"].clone().requires_grad_(True)
    y = F.conv3d(oracle.depth_shift(torch.cat(leaf_srcs, 1)), wl, bl, stride=stride, padding=(0, 1, 1))
    got_y = op.out.data.cpu()
    assert got_y.shape == y.shape
    assert (got_y - y.detach()).abs().max() < 2e-5, "conv output"
    u = F.instance_norm(y, weight=gl, bias=tl, eps=1e-5)
    z = F.leaky_relu(u, 0.01)
    z_got = _act_value(op.out)
    assert (z_got - z.detach()).abs().max() < 1e-4, "normalised output"
    # ---- backward
    dz = seeded_input(tuple(z.shape), seed=8)
    # an element sitting on the LeakyReLU kink (|u| of the order of the fp32 noise of u) legitimately takes either branch,
    # which moves its dy by ~|dz| and, through the InstanceNorm-backward sums, everything downstream; with millions of
    # elements some always do.  No gradient is sent into those elements.
    dz[u.detach().abs() < 2e-5] = 0.0
    z.backward(dz)
    for s in srcs:
        s._grad_written = False
    op.out.alloc_grad()
    op.plan_backward()
    op.out.grad.copy_(dz)
    for s in srcs:
        s.grad.fill_(float("nan"))           # first writer must overwrite
    op.backward()
    torch.cuda.synchronize()
    scale_w = max(1.0, float(wl.grad.abs().max()))
    _assert_wgrad(e.grads["blk.conv.weight"].cpu(), wl.grad, leaf_srcs, y.detach(), gl.detach(), tl.detach(), dz, stride, scale_w)
    assert (e.grads["blk.instnorm.weight"].cpu() - gl.grad).abs().max() < 2e-4 * max(1.0, float(gl.grad.abs().max()))
    assert (e.grads["blk.instnorm.bias"].cpu() - tl.grad).abs().max() < 2e-4 * max(1.0, float(tl.grad.abs().max()))
    assert (e.grads["blk.conv.bias"].cpu() - bl.grad).abs().max() < 1e-3 * max(1.0, float(dz.abs().sum()) * 1e-3)
    for s, leaf in zip(srcs, leaf_srcs):
        got = s.grad.cpu()
        assert torch.isfinite(got).all(), "dgrad left unwritten cells"
        assert (got - leaf.grad).abs().max() < 2e-4 * max(1.0, float(leaf.grad.abs().max())), "dgrad"
    # accumulate mode: a second backward with accumulate flags set adds on top
    for s in srcs:
        s._grad_written = True
    op.plan_backward()
    op.out.grad.copy_(dz)
    op.backward()
    torch.cuda.synchronize()
    for s, leaf in zip(srcs, leaf_srcs):
        assert (s.grad.cpu() - 2 * leaf.grad).abs().max() < 4e-4 * max(1.0, float(leaf.grad.abs().max())), "dgrad accumulate"


@pytest.mark.parametrize("B,cin,cout,dims,kernel,density,normed", [
    (1, 16, 8, (4, 6, 10), (2, 2, 2), 1.0, True),
    (2, 40, 33, (3, 5, 7), (2, 2, 2), 0.3, True),
    (1, 9, 5, (2, 9, 8), (1, 2, 2), 1.0, True),
    (1, 70, 64, (1, 4, 4), (1, 1, 1), 0.5, False),
    (1, 64, 32, (16, 16, 16), (2, 2, 2), 0.2, True),
    (1, 72, 40, (16, 64, 64), (2, 2, 2), 0.2, True),       # >= 2048 tiles of 32 voxels: matrix-core data gradient, ragged blocks
    (2, 20, 70, (32, 32, 34), (1, 2, 2), 0.5, False),      # same path (fp32 matrix instructions: W % 32 != 0), [1,2,2] kernel, three output-channel chunks
    (2, 20, 70, (8, 32, 32), (1, 2, 2), 0.5, False),       # bf16 three-piece data gradient, [1,2,2] kernel, three output-channel chunks
    (1, 136, 24, (8, 32, 32), (2, 2, 2), 0.3, True),       # forward GEMM with eight 32-channel k-blocks (one column tile per wave), ragged Cin
])
def test_convT_fwd_bwd(B, cin, cout, dims, kernel, density, normed):
    from e2enet_medical_amd.engine import UpOp
    from e2enet_medical_amd._lib import lib
    src = _make_act((B, cin) + dims, normed, 21)
    w = seeded_input((cin, cout) + kernel, seed=22) * (1.0 / math.sqrt(cin))
    km = _kmask(cin, cout, density, 23)
    if km is not None:
        w = w * km.view(cin, cout, 1, 1, 1)
    e = _eng_stub({"up.weight": w})
    op = UpOp(e, "up.weight", src, cout, kernel)
    if km is not None:
        rows = torch.empty(cin * ((cout + 31) // 32), dtype=torch.int32, device=e.device)
        cols = torch.empty(cout * ((cin + 31) // 32), dtype=torch.int32, device=e.device)
        kmd = km.to(e.device)
        lib().dsff_expand(kmd.data_ptr(), None, rows.data_ptr(), cols.data_ptr(), cin, cout, 1, 0)
        op.live, op.live_t = cols, rows
    # operand ranges as the engine hands them over (here: measured maxima): the matrix-pipe shapes then run on fp16 two-piece
    # operands, the product's default (test_convT_h2_and_bf3_vs_fp64 holds the bf16 three-piece form against the same fp64 values)
    op.set_ranges(float(_act_value(src).abs().max()), float(w.abs().max()), 1.0)
    op.forward()
    xl = _act_value(src).requires_grad_(True)
    wl = w.clone().requires_grad_(True)
    y = F.conv_transpose3d(xl, wl, stride=kernel)
    assert (op.out.data.cpu() - y.detach()).abs().max() < 2e-5
    dy = seeded_input(tuple(y.shape), seed=24)
    y.backward(dy)
    op.out.alloc_grad()
    op.plan_backward()
    op.out.grad.copy_(dy)
    dy_word = _absmax_word(op.out.grad)
    op.dy_word = dy_word.data_ptr()
    src.grad.fill_(float("nan"))
    op.backward()
    torch.cuda.synchronize()
    assert (e.grads["up.weight"].cpu() - wl.grad).abs().max() < 2e-4 * max(1.0, float(wl.grad.abs().max())), "wgrad"
    assert (src.grad.cpu() - xl.grad).abs().max() < 2e-4 * max(1.0, float(xl.grad.abs().max())), "dgrad"


@pytest.mark.parametrize("B,c,dims,kernel", [(2, 5, (4, 6, 8), (2, 2, 2)), (1, 3, (3, 10, 6), (1, 2, 2)),
                                             (1, 4, (5, 7, 9), (2, 2, 2))])
def test_maxpool_fwd_bwd(B, c, dims, kernel):
    from e2enet_medical_amd.engine import PoolOp
    src = _make_act((B, c) + dims, True, 31)
    op = PoolOp(_eng_stub({}), "down", src, kernel)
    op.forward()
    xl = _act_value(src).requires_grad_(True)
    y = F.max_pool3d(xl, kernel)
    assert (op.out.data.cpu() - y.detach()).abs().max() < 1e-6     # fma vs mul+add in the on-load affine
    dy = seeded_input(tuple(y.shape), seed=32)
    y.backward(dy)
    op.out.alloc_grad()
    op.plan_backward()
    op.out.grad.copy_(dy)
    src.grad.fill_(float("nan"))
    op.backward()
    assert torch.equal(src.grad.cpu(), xl.grad)                    # gradient routing (arg max) is exact


@pytest.mark.parametrize("B,c,k,dims", [(2, 8, 3, (4, 6, 8)), (1, 32, 4, (8, 16, 16)), (1, 20, 14, (3, 5, 7)),
                                        (1, 12, 16, (2, 4, 6))])
def test_head_fwd_bwd(B, c, k, dims):
    from e2enet_medical_amd.engine import HeadOp
    src = _make_act((B, c) + dims, True, 41)
    w = seeded_input((k, c, 1, 1, 1), seed=42) * 0.3
    e = _eng_stub({"seg.weight": w})
    op = HeadOp(e, "seg.weight", src, k)
    op.forward()
    xl = _act_value(src).requires_grad_(True)
    wl = w.clone().requires_grad_(True)
    y = F.conv3d(xl, wl)
    assert (op.out.data.cpu() - y.detach()).abs().max() < 1e-5
    dy = seeded_input(tuple(y.shape), seed=43)
    y.backward(dy)
    op.out.alloc_grad()
    op.plan_backward()
    op.out.grad.copy_(dy)
    src.grad.fill_(float("nan"))
    op.backward()
    torch.cuda.synchronize()
    assert (e.grads["seg.weight"].cpu() - wl.grad).abs().max() < 2e-4 * max(1.0, float(wl.grad.abs().max()))
    assert (src.grad.cpu() - xl.grad).abs().max() < 1e-5


@pytest.mark.parametrize("batch_dice", [False, True])
@pytest.mark.parametrize("k", [3, 4, 14])
def test_loss_kernels(k, batch_dice):
    from e2enet_medical_amd.training.loss_functions.dice_loss import DC_and_CE_loss
    shape = (2, k, 6, 10, 9)
    logits = (seeded_input(shape, seed=51) * 2).requires_grad_(True)
    g = torch.Generator().manual_seed(52)
    target = torch.randint(0, k, (2, 1, 6, 10, 9), generator=g).float()
    ref = oracle.dc_ce_loss(logits, target, batch_dice)
    ref.backward()
    lg = logits.detach().cuda().requires_grad_(True)
    loss = DC_and_CE_loss({'batch_dice': batch_dice, 'smooth': 1e-5, 'do_bg': False}, {})(lg, target.cuda())
    loss.backward()
    assert abs(loss.item() - ref.item()) < 2e-6
    assert (lg.grad.cpu() - logits.grad).abs().max() < 2e-8 + 1e-4 * float(logits.grad.abs().max())


def test_loss_golden():
    """Same fixture as the oracle test: MultipleOutputLoss2(DC_and_CE_loss) value + gradients from the reference."""
    from e2enet_medical_amd.training.loss_functions.dice_loss import DC_and_CE_loss
    from e2enet_medical_amd.training.loss_functions.deep_supervision import MultipleOutputLoss2
    from tests.helpers import seeded_labels
    g = golden("loss.npz")
    for tag, bd in (("sample", False), ("batch", True)):
        k = 4
        shapes = [(2, k, 8, 12, 10), (2, k, 4, 6, 5), (2, k, 2, 3, 5), (2, k, 1, 3, 5)]
        logits = [seeded_input(s, seed=50 + i).mul(2.0).cuda().requires_grad_(True) for i, s in enumerate(shapes)]
        targets = [seeded_labels((s[0], 1) + s[2:], k, seed=60 + i).cuda() for i, s in enumerate(shapes)]
        fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': bd, 'smooth': 1e-5, 'do_bg': False}, {}), oracle.ds_weights(5))
        loss = fn(logits, targets)
        loss.backward()
        assert abs(loss.item() - float(g[tag + "_loss"])) < 2e-6
        for i, l in enumerate(logits):
            np.testing.assert_allclose(l.grad.cpu().numpy(), g[tag + "_g%d" % i], rtol=0, atol=2e-7)


def test_fused_clip_sgd_matches_torch():
    from e2enet_medical_amd.training.fused_optim import FusedClipSGD
    torch.manual_seed(0)
    shapes = [(7, 5, 1, 3, 3), (11,), (4, 6, 2, 2, 2)]
    cpu = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    gpu = [torch.nn.Parameter(p.detach().clone().cuda()) for p in cpu]
    opt_c = torch.optim.SGD(cpu, 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    opt_g = torch.optim.SGD(gpu, 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    names = ["a", "b", "c"]
    fused = FusedClipSGD(opt_g, list(zip(names, gpu)), 12.0)
    mask = (torch.rand(shapes[0]) < 0.5).float()
    for step in range(4):
        grads = [torch.randn(s) * (40.0 if step == 1 else 0.2) for s in shapes]
        for p, gr in zip(cpu, grads):
            p.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_(cpu, 12)
        opt_c.step()
        with torch.no_grad():                                   # Masking.apply_mask on tensor "a"
            cpu[0].mul_(mask)
            opt_c.state[cpu[0]]['momentum_buffer'].mul_(mask)
        fused.step({n: gr.cuda() for n, gr in zip(names, grads)}, {"a": mask.cuda()})
        assert abs(fused.total_norm() - tn.item()) < 1e-4 * max(1.0, tn.item())
        for p, q in zip(cpu, gpu):
            assert (p.detach() - q.detach().cpu()).abs().max() < 2e-6
            assert (opt_c.state[p]['momentum_buffer'] - opt_g.state[q]['momentum_buffer'].cpu()).abs().max() < 2e-5


@pytest.mark.parametrize("tag,shp", [("l1_133", (320, 896, 1, 3, 3)), ("l1_222", (64, 32, 2, 2, 2)),
                                     ("l1_122", (16, 24, 1, 2, 2))])
def test_dsff_kernel_l1_bit_exact(tag, shp):
    """Golden from the reference's three chained torch.sum (core_channel.py:652-655): bit identical."""
    from e2enet_medical_amd._lib import lib
    w = closed_form_tensor(shp, 3, "conv").cuda()
    out = torch.empty(shp[0] * shp[1], dtype=torch.float32, device="cuda")
    lib().dsff_kernel_l1(w.data_ptr(), out.data_ptr(), shp[0], shp[1], shp[2], shp[3], shp[4], 0)
    assert np.array_equal(out.cpu().numpy().reshape(shp[0], shp[1]), golden("masks.npz")[tag])


@pytest.mark.parametrize("n", [1, 7, 1000, 286720])
def test_dsff_kth_value_exact(n):
    from e2enet_medical_amd._lib import lib
    g = torch.Generator().manual_seed(n)
    v = torch.rand(n, generator=g)
    v[::5] = 0.0                                # dead kernels have sum exactly 0, many ties
    if n > 10:
        v[3] = v[9]                             # a tie among live values
    srt, _ = torch.sort(v)
    vd = v.cuda()
    out = torch.empty(1, dtype=torch.float32, device="cuda")
    for k in sorted({0, n // 5, n // 2, max(0, n - 2), n - 1}):
        lib().dsff_kth_value(vd.data_ptr(), n, k, out.data_ptr(), None, 0)
        assert out.item() == srt[k].item(), (n, k)


def test_dsff_death_and_expand():
    from e2enet_medical_amd._lib import lib
    r, c = 37, 70
    km = (torch.rand((r, c), generator=torch.Generator().manual_seed(1)) < 0.4).to(torch.uint8)
    kmd = km.cuda()
    mask = torch.empty((r, c, 1, 3, 3), dtype=torch.float32, device="cuda")
    rows = torch.empty(r * 3, dtype=torch.int32, device="cuda")
    cols = torch.empty(c * 2, dtype=torch.int32, device="cuda")
    lib().dsff_expand(kmd.data_ptr(), mask.data_ptr(), rows.data_ptr(), cols.data_ptr(), r, c, 9, 0)
    assert torch.equal(mask.cpu(), km.float().view(r, c, 1, 1, 1).expand(r, c, 1, 3, 3))
    rows_h = rows.cpu().numpy().view(np.uint32).reshape(r, 3)
    cols_h = cols.cpu().numpy().view(np.uint32).reshape(c, 2)
    for i in range(r):
        for j in range(c):
            assert ((rows_h[i, j // 32] >> (j % 32)) & 1) == km[i, j].item()
            assert ((cols_h[j, i // 32] >> (i % 32)) & 1) == km[i, j].item()
    # quad words of the 1x3x3 conv kernels
    qr = torch.empty(((r + 3) // 4) * ((c + 7) // 8), dtype=torch.int32, device="cuda")
    qc = torch.empty(((c + 3) // 4) * ((r + 7) // 8), dtype=torch.int32, device="cuda")
    lib().dsff_expand_quads(kmd.data_ptr(), qr.data_ptr(), qc.data_ptr(), r, c, 0)
    qr_h = qr.cpu().numpy().view(np.uint32).reshape((r + 3) // 4, (c + 7) // 8)
    qc_h = qc.cpu().numpy().view(np.uint32).reshape((c + 3) // 4, (r + 7) // 8)
    want_r = np.zeros_like(qr_h)
    want_c = np.zeros_like(qc_h)
    for i in range(r):
        for j in range(c):
            if km[i, j].item():
                want_r[i // 4, j // 8] |= np.uint32(1) << np.uint32((j % 8) * 4 + i % 4)
                want_c[j // 4, i // 8] |= np.uint32(1) << np.uint32((i % 8) * 4 + j % 4)
    assert np.array_equal(qr_h, want_r) and np.array_equal(qc_h, want_c)
    l1 = torch.rand(r * c, generator=torch.Generator().manual_seed(2))
    thr = torch.tensor([0.3])
    l1d, thrd = l1.cuda(), thr.cuda()                      # keep device buffers alive across the async launch
    lib().dsff_death(l1d.data_ptr(), thrd.data_ptr(), kmd.data_ptr(), r * c, 0)
    assert torch.equal(kmd.cpu().view(-1), km.view(-1) * (l1 > 0.3).to(torch.uint8))
    w = seeded_input((r, c, 1, 3, 3), seed=3) * km.view(r, c, 1, 1, 1)
    out = torch.empty((r, c), dtype=torch.uint8, device="cuda")
    wd = w.cuda()
    lib().dsff_kmask_from_weights(wd.data_ptr(), out.data_ptr(), r, c, 9, 0)
    assert torch.equal(out.cpu(), km)


def test_sliding_window_kernels():
    from e2enet_medical_amd._lib import lib
    L = lib()
    c, X, Y, Z = 3, 5, 6, 7
    x = seeded_input((c, X, Y, Z), seed=61)
    xd = x.cuda()
    for bits in range(8):
        dims = [d + 1 for d in range(3) if bits & (1 << d)]
        out = torch.empty_like(xd)
        L.flip3d(xd.data_ptr(), out.data_ptr(), c, X, Y, Z, bits, 0)
        assert torch.equal(out.cpu(), torch.flip(x, dims) if dims else x)
        res = torch.full((c, X, Y, Z), 0.25, device="cuda")
        L.softmax_flip_acc(xd.data_ptr(), res.data_ptr(), 0.125, 0, c, X, Y, Z, bits, 0)
        p = torch.softmax(x, 0)
        ref = 0.25 + 0.125 * (torch.flip(p, dims) if dims else p)
        assert (res.cpu() - ref).abs().max() < 1e-6
    K, VX, VY, VZ = 3, 9, 10, 11
    agg = torch.zeros((K, VX, VY, VZ), device="cuda")
    cnt = torch.zeros_like(agg)
    patch = torch.rand((K, 5, 6, 7), generator=torch.Generator().manual_seed(5))
    gs = torch.rand((5, 6, 7), generator=torch.Generator().manual_seed(6)) + 0.1
    ra, rc = torch.zeros((K, VX, VY, VZ)), torch.zeros((K, VX, VY, VZ))
    pd, gd = patch.cuda(), gs.cuda()
    for (x0, y0, z0) in [(0, 0, 0), (4, 4, 4), (2, 1, 3)]:
        L.sw_accumulate(pd.data_ptr(), gd.data_ptr(), agg.data_ptr(), cnt.data_ptr(), K, VX, VY, VZ, 5, 6, 7, x0, y0, z0, 0)
        ra[:, x0:x0 + 5, y0:y0 + 6, z0:z0 + 7] += patch * gs
        rc[:, x0:x0 + 5, y0:y0 + 6, z0:z0 + 7] += gs
    assert torch.equal(agg.cpu(), ra) and torch.equal(cnt.cpu(), rc)
    probs = torch.empty((K, 5, 6, 7), device="cuda")
    seg = torch.empty((5, 6, 7), dtype=torch.int64, device="cuda")
    L.sw_finalize_argmax(agg.data_ptr(), cnt.data_ptr(), probs.data_ptr(), seg.data_ptr(), K, VX, VY, VZ, 2, 1, 3, 5, 6, 7, 0)
    rp = (ra / rc)[:, 2:7, 1:7, 3:10]
    assert torch.equal(probs.cpu(), rp) and torch.equal(seg.cpu(), rp.argmax(0))


def test_conv133_persistent_run_loop_forced():
    """The persistent run loop of conv133_kernel (a workgroup walks a run of consecutive tiles, rebuilds the plane table
    when the (batch item, slice) changes and requests the next tile's first chunk under the current epilogue) is only
    selected for single-chunk layers with more tiles than workgroup slots.  Force it for every float4-staged shape with
    a budget of 16 workgroups (runs that cross plane groups, slices, batch items and the dead slices of depth-strided
    data gradients) in a child process -- the knobs are read once per process -- and run the operator cases again."""
    import subprocess
    import sys
    env = dict(os.environ, E2E_CONV_PERSIST="1", E2E_CONV_WGS="16")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_conv133_fwd_bwd", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


@pytest.mark.parametrize("env", [{"E2E_WG_H2": "0"}, {"E2E_WG_BF3": "0"}, {"E2E_CONV_MM": "0"}, {"E2E_MM_GRID": "8"}])
def test_conv133_wgrad_alternative_paths_forced(env):
    """The dense weight gradient of stride-1 planes at least 16 voxels wide runs on the matrix pipe with split fp32 operands
    (conv133_wgrad_bf3v5_kernel<G, NPC>): by default on fp16 two-piece operands (three products), scaled from the max |dy| that
    e2e_in_lrelu_bwd records.  E2E_WG_H2=0 keeps the bf16 three-piece form (six products; also what a caller without the recorded
    maximum gets), E2E_WG_BF3=0 the fp32-MFMA kernels.  E2E_CONV_MM=0: forward / data gradient of the 16 x 32 tile class on the
    round-4 kernels (sparse plan walk, bf16x3 dense kernel) instead of conv133_mm_kernel; E2E_MM_GRID=8: that kernel as eight
    persistent workgroups (long item runs per workgroup: every pipeline transition).  The knobs are read once per process: run the
    operator cases again in a child process."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_conv133_fwd_bwd", "-p", "no:cacheprovider"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


def _absmax_word(t):
    """the device word e2e_in_lrelu_bwd leaves behind: bit pattern of max |t|"""
    return t.abs().max().reshape(1).contiguous().view(torch.int32)


def _heavy_tailed(shape, seed, scale):
    g = torch.Generator().manual_seed(seed)
    return scale * torch.randn(shape, generator=g) * torch.exp(2.5 * torch.randn(shape, generator=g))


@pytest.mark.parametrize("tag,B,src_desc,cout,dims", [
    ("ragged", 2, [(33, True), (20, False)], 40, (3, 20, 36)),       # ragged channel blocks and tiles, depth shifts
    ("whole", 1, [(32, True), (32, False)], 32, (9, 32, 64)),
    ("one_tile_runs", 1, [(16, False)], 16, (1, 20, 32)),
    ("w16", 1, [(40, True)], 33, (4, 24, 16)),                        # 8 x 16-pixel tiles
])
def test_conv133_wgrad_h2_and_bf3_vs_fp64(tag, B, src_desc, cout, dims):
    """Both operand formats of the matrix-pipe weight gradient against an fp64 evaluation, on a dy of realistic magnitudes
    (heavy-tailed, 1e-7: below the fp16 range unscaled).  fp16 two-piece (three products, scaled from max |dy|) and bf16
    three-piece (six products) must be within 6e-7 / 3e-7 of sum |dy x| per entry (measured 0.4-0.8e-7 rms) and the fp16 form no
    further than 1.5x the bf16 form's rms; and because the scale is an exact power of two taken from the data, the fp16 result of
    2^-20 dy is 2^-20 times the result of dy, bit for bit."""
    from e2enet_medical_amd.engine import ConvOp
    from e2enet_medical_amd._lib import lib
    srcs = [_make_act((B, c) + dims, normed, 70 + i) for i, (c, normed) in enumerate(src_desc)]
    cin = sum(c for c, _ in src_desc)
    params = {"blk.conv.weight": torch.zeros(cout, cin, 1, 3, 3), "blk.conv.bias": torch.zeros(cout),
              "blk.instnorm.weight": torch.ones(cout), "blk.instnorm.bias": torch.zeros(cout)}
    e = _eng_stub(params)
    op = ConvOp(e, "blk", srcs, cout, (1, 1, 1))
    dy = _heavy_tailed((B, cout) + dims, 7, 1e-7)
    x = oracle.depth_shift(torch.cat([_act_value(a) for a in srcs], 1)).double()
    xp = F.pad(x, (1, 1, 1, 1))
    H, W = dims[1:]
    ref = torch.zeros(cout, cin, 1, 3, 3, dtype=torch.float64)
    mag = torch.zeros_like(ref)
    for kh in range(3):
        for kw in range(3):
            win = xp[..., kh:kh + H, kw:kw + W]
            ref[:, :, 0, kh, kw] = torch.einsum("nodhw,ncdhw->oc", dy.double(), win)
            mag[:, :, 0, kh, kw] = torch.einsum("nodhw,ncdhw->oc", dy.double().abs(), win.abs())
    L = lib()

    xmax = float(x.abs().max())

    def run(dyt, with_max, x_word=True):
        dyd = dyt.cuda()
        word = _absmax_word(dyd) if with_max else None
        op.set_input_range(xmax if x_word else None)
        dw = torch.full((cout, cin, 1, 3, 3), float("nan"), device="cuda")
        L.conv133_wgrad(op.chans.data_ptr(), dyd.data_ptr(), dw.data_ptr(), e.wgrad_ws.data_ptr(), B, cin, cout, *dims, 1, 1, 1,
                        word.data_ptr() if with_max else None, op.x_absmax_ptr(), 0)
        torch.cuda.synchronize()
        return dw.cpu(), (L.last_kernel() or b"").decode()
    got_h2, k_h2 = run(dy, True)
    got_b3, k_b3 = run(dy, False)
    got_fixed, _ = run(dy, True, x_word=False)             # round 5's fixed 2^3 input scale: same products up to the lo pieces' range
    assert float((got_fixed.double() - ref).norm() / ref.norm()) < 2e-6
    assert k_h2.startswith("conv133_wgrad_h2") and k_b3.startswith("conv133_wgrad_bf3"), (k_h2, k_b3)
    rms = {}
    for name, got in (("h2", got_h2), ("bf3", got_b3)):
        assert torch.isfinite(got).all(), name
        live = mag > 0                                         # (one-slice volumes: the depth shift empties whole input channels)
        assert float(got[~live].abs().max()) == 0.0 if bool((~live).any()) else True, name
        err = torch.where(live, (got.double() - ref) / torch.where(live, mag, torch.ones_like(mag)), torch.zeros_like(mag))
        rms[name] = float(err.pow(2).mean().sqrt())
        # (an entry dominated by ONE product sees that product's own error: the dropped lo*lo term, <= 2^-22, plus two operand
        #  roundings of <= 2^-23 each -- 4.8e-7 for the fp16 form; the truncating bf16 splits drop up to 2^-22 as well)
        assert float(err.abs().max()) < 6e-7, (name, float(err.abs().max()))
        assert float((got.double() - ref).norm() / ref.norm()) < 2e-6, name
    assert rms["h2"] <= 1.5 * rms["bf3"] + 1e-9, rms
    print("wgrad %s: err / sum|dy x| rms h2 %.2e bf3 %.2e" % (tag, rms["h2"], rms["bf3"]))
    got_small, _ = run(dy * 2.0 ** -20, True)
    assert torch.equal(got_small, got_h2 * 2.0 ** -20), "power-of-two scale invariance"
    # a dy of zeros (dead branch) and a dy with one huge entry: finite, exact zero / dominated by that entry
    z, _ = run(torch.zeros_like(dy), True)
    assert torch.equal(z, torch.zeros_like(z))


@pytest.mark.parametrize("K", [256, 4096, 65536])
@pytest.mark.parametrize("dist", ["uniform", "act_x_grad", "positive"])
def test_split_operand_products_vs_fp64(K, dist):
    """The numerics gate of the matrix-pipe paths (round 5: moved here from tools/scratch/bf3_numerics.hip): a product rebuilt
    from bf16 three-piece operands (six products) and from fp16 two-piece operands (three products, the gradient-like operand
    scaled from its max) through the library's own split functions, against fp64, beside the fp32-input MFMA (a plain fp32 FMA
    chain = what an fp32 kernel computes).  Bars: rms error / sum |a b| no more than 1.5x the fp32 chain's (measured 0.6-1.1x;
    the accumulation in one fp32 chain dominates all three) and, for the fp16 form the hot kernels use, no bias beyond it on
    one-signed operands (the bf16 form has one there: see below)."""
    from e2enet_medical_amd._lib import lib
    g = torch.Generator().manual_seed(11 + K)
    if dist == "uniform":
        a = (torch.rand((32, K), generator=g) * 2 - 1) * torch.randint(1, 8, (32, K), generator=g)
        b = (torch.rand((32, K), generator=g) * 2 - 1) * torch.randint(1, 8, (32, K), generator=g)
    elif dist == "act_x_grad":
        a = F.leaky_relu(torch.randn((32, K), generator=g), 0.01)
        b = _heavy_tailed((32, K), 12 + K, 1e-7)
    else:
        a = torch.randn((32, K), generator=g).abs()
        b = 1e-3 * torch.randn((32, K), generator=g).abs()
    ref = a.double() @ b.double().t()
    mag = a.double().abs() @ b.double().abs().t()
    ad, bd = a.cuda(), b.cuda()
    word = _absmax_word(bd)
    out = {}
    for mode, name in ((0, "bf16x3"), (1, "fp16x2"), (2, "fp32")):
        d = torch.empty((32, 32), device="cuda")
        lib().diag_split_gemm(ad.data_ptr(), bd.data_ptr(), d.data_ptr(), K, mode, word.data_ptr(), 0)
        err = (d.cpu().double() - ref) / mag
        out[name] = (float(err.pow(2).mean().sqrt()), float(err.mean()), float(err.abs().max()))
        assert torch.isfinite(d).all()
    print("K %d %s: rms / mean / max of err / sum|ab|: %s" % (K, dist, out))
    for name in ("bf16x3", "fp16x2"):
        if name == "bf16x3" and dist == "positive":
            # one-signed operands grow ONE fp32 accumulator monotonically; the bf16 third pieces' products (2^-16 of a product)
            # fall below half an ulp of it once it holds ~2^8 products and are lost one by one: a bias of -2^-17 = -7.6e-6 of the
            # sum (measured -9.5e-6 / -2.3e-5 at K = 4096 / 65536).  The fp16 second pieces (2^-11) stay above the ulp 32 times
            # longer and the 16 products of a matrix instruction are summed before they meet the accumulator: no such bias (below).
            # Kernels on bf16x3 (transposed convs, the E2E_CONV_MM=0 / E2E_WG_H2=0 paths) either flush per chunk into a second
            # accumulator (dense conv) or sum mixed-sign gradients; the bound kept here is the size of the effect.
            assert abs(out[name][1]) <= 4e-5 and out[name][0] <= 4e-5, (name, out)
            continue
        assert out[name][0] <= 1.5 * out["fp32"][0] + 2e-8, (name, out)
        assert abs(out[name][1]) <= 1.5 * abs(out["fp32"][1]) + 0.5 * out["fp32"][0] + 2e-8, (name, out)


@pytest.mark.parametrize("shape", [(2, 1, 16, 32, 32), (1, 1, 40, 56, 40), (2, 2, 7, 9, 13)])
def test_ds_target_gather_vs_oracle(shape):
    """Deep-supervision targets gathered on the device against the oracle (scipy's zoom, order 0): bit exact."""
    from e2enet_medical_amd.training.data_augmentation.downsampling import downsample_seg_for_ds_transform2
    scales = [[1, 1, 1], [0.5, 0.5, 0.5], [0.25, 0.25, 0.25], [1, 0.5, 0.5], [0.5, 1, 0.25], [0.125, 0.0625, 0.0625]]
    seg = np.random.RandomState(5).randint(0, 14, size=shape).astype(np.float32)
    ref = oracle.downsample_seg_for_ds(seg, scales)
    got = downsample_seg_for_ds_transform2(torch.from_numpy(seg).cuda(), scales, order=0)
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert tuple(g.shape) == r.shape
        assert np.array_equal(g.cpu().numpy(), r)
    with pytest.raises(NotImplementedError):
        downsample_seg_for_ds_transform2(torch.from_numpy(seg).cuda(), scales, order=1)
    with pytest.raises(RuntimeError):
        downsample_seg_for_ds_transform2(torch.from_numpy(seg), scales, order=0)


@pytest.mark.parametrize("B,src_desc,cout,dims,stride,density", [
    (2, [(320, True), (320, False), (256, False)], 320, (8, 8, 8), (1, 1, 1), 0.2),      # BASELINE level 4: 896 -> 320 @8^3
    (2, [(256, True), (256, False), (128, False)], 256, (4, 16, 16), (1, 1, 1), 0.2),    # level 3 planes (16 x 16 tile)
    (1, [(100, True), (28, False)], 70, (5, 7, 5), (1, 1, 1), 0.5),                      # Hippocampus-like ragged planes
    (2, [(256, True)], 320, (4, 16, 16), (2, 2, 2), 1.0),                                # strided onto 8 x 8 planes
    (1, [(320, True)], 320, (2, 4, 4), (1, 1, 1), 1.0),                                  # 4 x 4 planes in an 8 x 8 tile
    (1, [(100, True), (60, False)], 40, (3, 12, 40), (1, 1, 1), 0.3),                    # three ragged 16 x 16 tiles per slice
])
def test_conv133_forward_split_k_matches_unsplit(B, src_desc, cout, dims, stride, density):
    """Deep levels: the forward with its input-plane chunks split over several workgroups (+ the sum kernel: fixed order,
    bias, InstanceNorm partials) against the unsplit forward of the same operands and against torch; deterministic."""
    from e2enet_medical_amd.engine import ConvOp
    from e2enet_medical_amd._lib import lib
    L = lib()
    srcs = [_make_act((B, c) + dims, normed, 40 + i) for i, (c, normed) in enumerate(src_desc)]
    cin = sum(c for c, _ in src_desc)
    w = seeded_input((cout, cin, 1, 3, 3), seed=3) * (1.0 / math.sqrt(cin * 9))
    km = _kmask(cout, cin, density, 5)
    if km is not None:
        w = w * km.view(cout, cin, 1, 1, 1)
    params = {"blk.conv.weight": w, "blk.conv.bias": seeded_input((cout,), seed=4) * 0.1,
              "blk.instnorm.weight": 1 + 0.2 * seeded_input((cout,), seed=6), "blk.instnorm.bias": 0.2 * seeded_input((cout,), seed=7)}
    e = _eng_stub(params)
    e.batch = B
    op = ConvOp(e, "blk", srcs, cout, stride)
    if km is not None:
        rows = torch.empty(((cout + 3) // 4) * ((cin + 7) // 8), dtype=torch.int32, device=e.device)
        cols = torch.empty(((cin + 3) // 4) * ((cout + 7) // 8), dtype=torch.int32, device=e.device)
        L.dsff_expand_quads(km.to(e.device).data_ptr(), rows.data_ptr(), cols.data_ptr(), cout, cin, 0)
        op.live, op.live_t = rows, cols
    assert op.fwd_ws_bytes > 0, "this shape is expected to split"
    op.forward()                                            # no workspace on the stub engine: unsplit
    assert b"ksplit=1" in L.last_kernel()
    y0, sc0, sh0 = op.out.data.clone(), op.out.scale.clone(), op.out.shift.clone()
    e.fwd_ws = torch.empty(op.fwd_ws_bytes // 4, dtype=torch.float32, device=e.device)
    op.out.data.fill_(float("nan"))
    op.forward()
    name = L.last_kernel().decode()
    assert "ksplit=" in name and int(name.rsplit("ksplit=", 1)[1]) > 1, name
    y1, sc1, sh1 = op.out.data.clone(), op.out.scale.clone(), op.out.shift.clone()
    assert torch.isfinite(y1).all()
    assert (y1 - y0).abs().max().item() <= 2e-5 * max(1.0, y0.abs().max().item())
    assert (sc1 - sc0).abs().max().item() <= 1e-4 * max(1.0, sc0.abs().max().item())
    assert (sh1 - sh0).abs().max().item() <= 1e-4 * max(1.0, sh0.abs().max().item())
    op.forward()
    assert torch.equal(op.out.data, y1) and torch.equal(op.out.scale, sc1)       # fixed summation order
    ref = _ref_conv(srcs, params["blk.conv.weight"], params["blk.conv.bias"], stride)
    assert (y1.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(RuntimeError):                        # a workspace that is too small is an argument error
        small = torch.empty(16, dtype=torch.float32, device=e.device)
        L.conv133_fwd_splitk(op.chans.data_ptr(), cin, e.params["blk.conv.weight"].data_ptr(), e.params["blk.conv.bias"].data_ptr(),
                             op.live.data_ptr() if op.live is not None else None, op.out.data.data_ptr(), op.part.data_ptr(), B,
                             cout, *dims, *stride, small.data_ptr(), 64, 0)


@pytest.mark.parametrize("B,src_desc,cout,dims,stride,density", [
    (2, [(320, True), (320, False), (256, False)], 320, (8, 8, 8), (1, 1, 1), 0.2),      # level 4
    (2, [(256, True), (256, False), (128, False)], 256, (4, 16, 16), (1, 1, 1), 0.2),    # level 3 (16 x 16 tile)
    (1, [(100, True), (28, False)], 70, (5, 7, 5), (1, 1, 1), 0.5),                      # ragged planes, shifted groups
    (2, [(256, True)], 320, (4, 16, 16), (2, 2, 2), 1.0),                                # strided: sub-pixel data gradient
    (1, [(320, True)], 320, (2, 4, 4), (1, 1, 1), 1.0),
    (1, [(100, True), (60, False)], 128, (3, 12, 40), (1, 1, 1), 0.3),                    # three ragged 16 x 16 tiles per slice
])
def test_conv133_data_gradient_split_k_matches_unsplit(B, src_desc, cout, dims, stride, density):
    """Deep levels: data gradient with the dy-plane chunks split over several workgroups (+ the sum / scatter kernel:
    un-shift, zero-fill, accumulate) against the unsplit kernel, in overwrite and in accumulate mode; deterministic."""
    from e2enet_medical_amd.engine import ConvOp
    from e2enet_medical_amd._lib import lib
    L = lib()
    srcs = [_make_act((B, c) + dims, normed, 50 + i) for i, (c, normed) in enumerate(src_desc)]
    cin = sum(c for c, _ in src_desc)
    w = seeded_input((cout, cin, 1, 3, 3), seed=3) * (1.0 / math.sqrt(cin * 9))
    km = _kmask(cout, cin, density, 5)
    if km is not None:
        w = w * km.view(cout, cin, 1, 1, 1)
    params = {"blk.conv.weight": w, "blk.conv.bias": seeded_input((cout,), seed=4) * 0.1,
              "blk.instnorm.weight": 1 + 0.2 * seeded_input((cout,), seed=6), "blk.instnorm.bias": 0.2 * seeded_input((cout,), seed=7)}
    e = _eng_stub(params)
    e.batch = B
    op = ConvOp(e, "blk", srcs, cout, stride)
    if km is not None:
        rows = torch.empty(((cout + 3) // 4) * ((cin + 7) // 8), dtype=torch.int32, device=e.device)
        cols = torch.empty(((cin + 3) // 4) * ((cout + 7) // 8), dtype=torch.int32, device=e.device)
        L.dsff_expand_quads(km.to(e.device).data_ptr(), rows.data_ptr(), cols.data_ptr(), cout, cin, 0)
        op.live, op.live_t = rows, cols
    assert op.dgrad_ws_bytes > 0, "this shape is expected to split"
    op.forward()
    dy = seeded_input(tuple(op.out.shape), seed=8).cuda()
    for s in srcs:
        s._grad_written = False
    op.out.alloc_grad()
    op.plan_backward()
    di, hi, wi = dims

    def run(ws, accumulate):
        for s in srcs:
            s._grad_written = accumulate
        op.plan_backward()
        for s in srcs:
            s.grad.copy_(s.data * 0.25) if accumulate else s.grad.fill_(float("nan"))
        args = (dy.data_ptr(), e.params["blk.conv.weight"].data_ptr(), op.live_t.data_ptr() if op.live_t is not None else None,
                op.outs.data_ptr(), B, cin, cout, di, hi, wi, *stride)
        if ws is None:
            L.conv133_dgrad(*args, 0)
        else:
            L.conv133_dgrad_splitk(*args, ws.data_ptr(), ws.numel() * 4, 0)
        name = L.last_kernel().decode()
        return [s.grad.clone() for s in srcs], name
    ws = torch.empty(op.dgrad_ws_bytes // 4, dtype=torch.float32, device=e.device)
    for accumulate in (False, True):
        ref, n0 = run(None, accumulate)
        got, n1 = run(ws, accumulate)
        again, _ = run(ws, accumulate)
        assert n0.endswith("ksplit=1") and int(n1.rsplit("ksplit=", 1)[1]) > 1, (n0, n1)
        for r, g, a in zip(ref, got, again):
            assert torch.isfinite(g).all()
            assert (g - r).abs().max().item() <= 2e-5 * max(1.0, r.abs().max().item())
            assert torch.equal(g, a)


@pytest.mark.parametrize("path,density", [("mm", 0.2), ("dense", 0.6), ("sparse", 0.2)])
def test_conv133_masks_are_structural_on_every_path(path, density, monkeypatch):
    """A DSFF map is enforced by the kernels, not by pruned weights happening to be zero: with garbage left in the dead
    kernels the fp16 two-piece matrix-pipe kernel (conv133_mm.hip, any density), the bf16x3 dense kernel (density >= 0.5) and the
    sparse walk (the latter two with the first switched off) must all return the masked convolution (forward and data gradient).
    (Advisor, round 3: the dense pack ignored the liveness words.)"""
    from e2enet_medical_amd import engine as engine_mod
    from e2enet_medical_amd.engine import ConvOp
    from e2enet_medical_amd._lib import lib
    if path != "mm":
        monkeypatch.setattr(engine_mod, "MM_MIN_DENSITY", 2.0)
    B, cin, cout, dims = 1, 40, 48, (3, 32, 64)
    srcs = [_make_act((B, cin) + dims, True, 61)]
    w_raw = seeded_input((cout, cin, 1, 3, 3), seed=62) * (1.0 / math.sqrt(cin * 9))
    km = _kmask(cout, cin, density, 63)
    w_masked = w_raw * km.view(cout, cin, 1, 1, 1)
    w_dirty = torch.where(km.view(cout, cin, 1, 1, 1) > 0, w_raw, torch.full_like(w_raw, 3.0))     # dead kernels hold 3.0
    params = {"blk.conv.weight": w_dirty, "blk.conv.bias": torch.zeros(cout), "blk.instnorm.weight": torch.ones(cout),
              "blk.instnorm.bias": torch.zeros(cout)}
    e = _eng_stub(params)
    op = ConvOp(e, "blk", srcs, cout, (1, 1, 1))
    e.fwd_ws = torch.empty(max(op.dense_ws_bytes, 4) // 4, dtype=torch.float32, device=e.device)
    rows = torch.empty(((cout + 3) // 4) * ((cin + 7) // 8), dtype=torch.int32, device=e.device)
    cols = torch.empty(((cin + 3) // 4) * ((cout + 7) // 8), dtype=torch.int32, device=e.device)
    lib().dsff_expand_quads(km.to(e.device).data_ptr(), rows.data_ptr(), cols.data_ptr(), cout, cin, 0)
    op.live, op.live_t, op.density = rows, cols, float(km.float().mean())
    assert op.use_mm() == (path == "mm") and (path == "mm" or op.use_dense() == (path == "dense"))
    op.out.alloc_grad()
    op.plan_backward()
    planned = _plan_and_pack(op, km)                # (the packed weights are built from the dirty tensor: pruned kernels must pack as zeros)
    assert planned
    op.forward()
    x = _act_value(srcs[0]).requires_grad_(True)
    y = F.conv3d(oracle.depth_shift(x), w_masked, None, padding=(0, 1, 1))
    assert (op.out.data.cpu() - y.detach()).abs().max() < 2e-5, "forward used a pruned kernel"
    dy = seeded_input(tuple(y.shape), seed=64)
    y.backward(dy)
    # data gradient alone (dy handed over as the pre-norm gradient)
    op.out.grad.copy_(dy)
    L = lib()
    if path == "mm":
        word = _absmax_word(op.out.grad)
        op.pack_mm_standalone("b")               # (from the dirty tensor: pruned kernels must pack as zeros)
        L.conv133_dgrad_mm(op.out.grad.data_ptr(), word.data_ptr(), op.wpk_bwd.data_ptr(), op.w_absmax.data_ptr(), op.outs.data_ptr(),
                           B, cin, cout, *dims, 0)
        torch.cuda.synchronize()
        assert L.last_kernel().decode().startswith("conv133_mm_h2<mode=1")
        assert (srcs[0].grad.cpu() - x.grad).abs().max() < 2e-4 * max(1.0, float(x.grad.abs().max())), "matrix-pipe data gradient used a pruned kernel"
        return
    if not op.use_dense():                          # the load-balanced kernel, then (below) the generic walk on the same data
        sp = op.sp_bwd
        L.conv133_dgrad_sparse(op.out.grad.data_ptr(), sp.wpk.data_ptr(), sp.quads.data_ptr(), sp.woff.data_ptr(), sp.kmax, sp.pslot.data_ptr(), op._bwd_table().data_ptr(),
                               None, sp.flush_every, B, cin, cout, *dims, 0)
        torch.cuda.synchronize()
        assert L.last_kernel().decode().startswith("conv133_sparse_kernel<mode=1>")
        assert (srcs[0].grad.cpu() - x.grad).abs().max() < 2e-4 * max(1.0, float(x.grad.abs().max())), "planned data gradient used a pruned kernel"
        srcs[0].grad.fill_(float("nan"))
    if op.use_dense():
        L.conv133_dgrad_dense(op.out.grad.data_ptr(), e.params["blk.conv.weight"].data_ptr(), op.live_t.data_ptr(), op.outs.data_ptr(),
                              B, cin, cout, *dims, e.fwd_ws.data_ptr(), e.fwd_ws.numel() * 4, 0)
    else:
        L.conv133_dgrad(op.out.grad.data_ptr(), e.params["blk.conv.weight"].data_ptr(), op.live_t.data_ptr(), op.outs.data_ptr(),
                        B, cin, cout, *dims, 1, 1, 1, 0)
    torch.cuda.synchronize()
    # srcs[0] is a normalised tensor: its grad buffer holds d/d(post-activation value), which is what x.grad is
    assert (srcs[0].grad.cpu() - x.grad).abs().max() < 2e-4 * max(1.0, float(x.grad.abs().max())), "data gradient used a pruned kernel"


@pytest.mark.parametrize("dims,kernel", [((8, 32, 32), (2, 2, 2)), ((16, 16, 32), (1, 2, 2)), ((8, 32, 34), (2, 2, 2))])
def test_convT_masks_are_structural_on_the_gemm_paths(dims, kernel):
    """Same for the transposed convolution: the bf16x3 forward / data-gradient GEMMs (W % 32 == 0) and the fp32-MFMA data
    gradient (other widths) drop pruned kernels where the weight enters a fragment."""
    from e2enet_medical_amd.engine import UpOp
    from e2enet_medical_amd._lib import lib
    B, cin, cout = 2, 40, 24
    src = _make_act((B, cin) + dims, True, 71)
    w_raw = seeded_input((cin, cout) + kernel, seed=72) * (1.0 / math.sqrt(cin))
    km = _kmask(cin, cout, 0.3, 73)
    w_masked = w_raw * km.view(cin, cout, 1, 1, 1)
    w_dirty = torch.where(km.view(cin, cout, 1, 1, 1) > 0, w_raw, torch.full_like(w_raw, -2.0))
    e = _eng_stub({"up.weight": w_dirty})
    op = UpOp(e, "up.weight", src, cout, kernel)
    rows = torch.empty(cin * ((cout + 31) // 32), dtype=torch.int32, device=e.device)
    cols = torch.empty(cout * ((cin + 31) // 32), dtype=torch.int32, device=e.device)
    lib().dsff_expand(km.to(e.device).data_ptr(), None, rows.data_ptr(), cols.data_ptr(), cin, cout, 1, 0)
    op.live, op.live_t = cols, rows
    op.forward()
    fwd_kernel = lib().last_kernel()
    xl = _act_value(src).requires_grad_(True)
    y = F.conv_transpose3d(xl, w_masked, stride=kernel)
    assert (op.out.data.cpu() - y.detach()).abs().max() < 2e-5, "forward used a pruned kernel (%s)" % fwd_kernel
    dy = seeded_input(tuple(y.shape), seed=74)
    y.backward(dy)
    op.out.alloc_grad()
    op.plan_backward()
    op.out.grad.copy_(dy)
    op.backward()
    torch.cuda.synchronize()
    assert (src.grad.cpu() - xl.grad).abs().max() < 2e-4 * max(1.0, float(xl.grad.abs().max())), "data gradient used a pruned kernel"


@pytest.mark.parametrize("what,mag", [("raw", 1e4), ("raw", 1e6), ("gamma", 50.0), ("tiny", 1e-6), ("weights", 1e4)])
def test_split_operand_kernels_hold_the_fp32_range(what, mag):
    """Round 6 (verdict r05 weak 2): the fp16 two-piece kernels move every operand into the fp16 range by a power of two derived from
    the tensor -- the activations from the range word (a bound of |x| over the conv's input planes), the weights from max |w| recorded
    by the packing launch, dy from max |dy| -- instead of the fixed 2^3 / 2^8 of round 5 (|x| > 8188 or |w| >= 256 became Inf).
    Conv forward (conv133_mm_h2<mode=0>), data gradient (<mode=1>) and weight gradient (conv133_wgrad_h2) against fp64 with
    activations of 1e4 and 1e6 (un-normalised source), a normalised source with |scale| = 50, activations of 1e-6 and weights of
    1e4: finite, within the usual RELATIVE bars (the reference computes in fp32 throughout, nnUNetTrainer_simple.py:551-573)."""
    from e2enet_medical_amd.engine import ConvOp
    from e2enet_medical_amd._lib import lib
    B, dims, cout = 1, (3, 32, 64), 40
    srcs = [_make_act((B, 24) + dims, True, 301), _make_act((B, 16) + dims, False, 302)]
    wscale = 1.0
    if what == "raw":
        srcs[1].data.mul_(mag)
    elif what == "gamma":
        srcs[0].scale.mul_(mag)
    elif what == "tiny":
        srcs[1].data.mul_(mag)
        srcs[0].scale.mul_(mag)
        srcs[0].shift.mul_(mag)
    else:
        wscale = mag
    cin = 40
    w = seeded_input((cout, cin, 1, 3, 3), seed=303) * (wscale / math.sqrt(cin * 9))
    params = {"blk.conv.weight": w, "blk.conv.bias": torch.zeros(cout), "blk.instnorm.weight": torch.ones(cout),
              "blk.instnorm.bias": torch.zeros(cout)}
    e = _eng_stub(params)
    op = ConvOp(e, "blk", srcs, cout, (1, 1, 1))
    assert op.use_mm()
    xs = [_act_value(a) for a in srcs]
    op.set_input_range(max(float(v.abs().max()) for v in xs))
    L = lib()
    op.forward()
    torch.cuda.synchronize()
    x64 = oracle.depth_shift(torch.cat(xs, 1)).double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = F.conv3d(x64, w64, None, padding=(0, 1, 1))
    got = op.out.data.cpu()
    assert torch.isfinite(got).all(), "forward overflowed"
    ysc = float(y64.abs().max())
    assert float((got.double() - y64.detach()).abs().max()) <= 2e-6 * ysc, "forward vs fp64 (relative to max |y|)"
    assert float((got.double() - y64.detach()).norm() / y64.detach().norm()) <= 6e-7
    # backward: dy handed over as the pre-norm gradient, with its recorded maximum (what e2e_in_lrelu_bwd leaves behind)
    dy = _heavy_tailed(tuple(y64.shape), 304, 1e-7)
    y64.backward(dy.double())
    op.out.alloc_grad()
    op.plan_backward()
    op.out.grad.copy_(dy)
    word = _absmax_word(op.out.grad)
    for s_ in srcs:
        s_.grad.fill_(float("nan"))
    op.pack_mm_standalone("b")
    L.conv133_dgrad_mm(op.out.grad.data_ptr(), word.data_ptr(), op.wpk_bwd.data_ptr(), op.w_absmax.data_ptr(), op.outs.data_ptr(),
                       B, cin, cout, *dims, 0)
    torch.cuda.synchronize()
    assert L.last_kernel().decode().startswith("conv133_mm_h2<mode=1")
    gx = torch.cat([s_.grad.cpu() for s_ in srcs], 1).double()
    # (x64 is the shifted concat: undo the shift on the reference side by shifting the engine's result the same way)
    ref_gx = x64.grad
    got_gx = oracle.depth_shift(gx.float()).double()
    live = oracle.depth_shift(torch.ones_like(gx).float()).double() > 0          # slices the shift moved out of range receive nothing
    assert torch.isfinite(gx).all(), "data gradient overflowed"
    assert float(((got_gx - ref_gx) * live).norm() / (ref_gx * live).norm()) <= 3e-6, "data gradient vs fp64"
    dw = torch.full((cout, cin, 1, 3, 3), float("nan"), device="cuda")
    L.conv133_wgrad(op.chans.data_ptr(), op.out.grad.data_ptr(), dw.data_ptr(), e.wgrad_ws.data_ptr(), B, cin, cout, *dims, 1, 1, 1,
                    word.data_ptr(), op.x_absmax_ptr(), 0)
    torch.cuda.synchronize()
    assert L.last_kernel().decode().startswith("conv133_wgrad_h2")
    assert torch.isfinite(dw).all(), "weight gradient overflowed"
    assert float((dw.cpu().double() - w64.grad).norm() / w64.grad.norm()) <= 3e-6, "weight gradient vs fp64"
    if what == "raw" and mag >= 1e4:
        # what round 5 did with the same data: the fixed 2^3 scale overflows fp16 (the hole this test closes)
        op.set_input_range(None)
        op.forward()
        torch.cuda.synchronize()
        assert not torch.isfinite(op.out.data).all()


def test_input_range_words_bound_the_activations():
    """e2e_conv133_input_ranges: the word of a conv is a rigorous bound of |x| over its input planes -- |gamma| sqrt(N - 1) + |beta|
    for a normalised source (and its max-pool), sum_c bound_c |W[c, o, k]| for a transposed conv of one, the measured maximum for a
    raw tensor -- never below the true maximum, and tight on adversarial data (one spike per instance reaches sqrt(N - 1))."""
    import ctypes as C
    from e2enet_medical_amd._lib import lib, RangeSrc, RangeJob
    L = lib()
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    Cn, N = 24, 6 * 8 * 10
    gamma = (torch.randn(Cn, generator=g) * 3).to(dev)
    beta = torch.randn(Cn, generator=g).to(dev)
    wt = (torch.randn(Cn, 12, 2, 2, 2, generator=g) * 0.2).to(dev)
    xin = (torch.randn(3, 5, 7, generator=g) * 40).to(dev)
    xin[1, 2, 3] = float("-9e3")
    words = torch.zeros(4, dtype=torch.int32, device=dev)
    L.absmax_word(xin.data_ptr(), xin.numel(), words[3:].data_ptr(), 0)
    none = RangeSrc(0, 0, 0, None, None, None, 0, 0, None)
    norm = RangeSrc(1, Cn, N, gamma.data_ptr(), beta.data_ptr(), None, 0, 0, None)
    up = RangeSrc(2, Cn, N, gamma.data_ptr(), beta.data_ptr(), wt.data_ptr(), 12, 8, None)
    raw = RangeSrc(3, 0, 0, None, None, None, 0, 0, words[3:].data_ptr())
    jobs = [RangeJob((RangeSrc * 3)(norm, none, none), words[0:].data_ptr()), RangeJob((RangeSrc * 3)(up, none, none), words[1:].data_ptr()),
            RangeJob((RangeSrc * 3)(norm, up, raw), words[2:].data_ptr())]
    table = torch.frombuffer(bytearray(b"".join(bytes(j) for j in jobs)), dtype=torch.uint8).to(dev)
    ws = torch.empty(int(L.conv133_input_ranges_ws_bytes(3)) // 4, dtype=torch.float32, device=dev)
    L.conv133_input_ranges(table.data_ptr(), 3, ws.data_ptr(), 0)
    torch.cuda.synchronize()
    got = words.view(torch.float32).cpu().double()
    root = math.sqrt(N - 1)
    b_norm = (gamma.abs().cpu().double() * root + beta.abs().cpu().double())
    b_up = (b_norm.view(Cn, 1, 1) * wt.abs().cpu().double().view(Cn, 12, 8)).sum(0).max()
    assert float(got[3]) == 9e3
    for val, ref in ((got[0], b_norm.max()), (got[1], b_up), (got[2], max(float(b_norm.max()), float(b_up), 9e3))):
        assert float(ref) <= float(val) <= float(ref) * (1 + 2e-3), (float(val), float(ref))
    # adversarial instance: one spike, everything else equal -> |xhat| of the spike = sqrt(N - 1) exactly; the bound holds with equality
    y = torch.zeros(1, 1, N)
    y[0, 0, 0] = 1.0
    xhat = (y - y.mean()) / y.var(unbiased=False).sqrt()
    assert abs(float(xhat.abs().max()) - root) < 1e-3 * root


@pytest.mark.parametrize("cin,cout,dims,kernel,mag", [(64, 32, (4, 32, 64), (2, 2, 2), 1.0), (40, 24, (8, 32, 32), (1, 2, 2), 1.0),
                                                     (128, 64, (8, 32, 32), (2, 2, 2), 1e4), (64, 32, (4, 32, 64), (2, 2, 2), 1e-5)])
def test_convT_h2_and_bf3_vs_fp64(cin, cout, dims, kernel, mag):
    """Round 6: the three transposed-conv GEMMs (forward, data gradient, weight gradient; reference nn.ConvTranspose3d,
    unetpp_d.py:521-522) on fp16 two-piece operands (three products) against an fp64 evaluation, next to the bf16 three-piece form
    (six products) they replace: both within the usual relative bars, the fp16 form no further than 1.5 x the bf16 form's error
    -- with activations of 1e4 and 1e-5 as well (operand scales from the range words), kernel names asserted."""
    from e2enet_medical_amd.engine import UpOp
    from e2enet_medical_amd._lib import lib
    B = 2
    src = _make_act((B, cin) + dims, True, 501)
    src.scale.mul_(mag)
    src.shift.mul_(mag)
    w = seeded_input((cin, cout) + kernel, seed=502) * (1.0 / math.sqrt(cin))
    e = _eng_stub({"up.weight": w})
    op = UpOp(e, "up.weight", src, cout, kernel)
    x64 = _act_value(src).double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = F.conv_transpose3d(x64, w64, stride=kernel)
    dy = _heavy_tailed(tuple(y64.shape), 503, 1e-6)
    y64.backward(dy.double())
    op.out.alloc_grad()
    op.plan_backward()
    L = lib()
    res = {}
    for form in ("h2", "bf3"):
        if form == "h2":
            op.set_ranges(float(x64.detach().abs().max()), float(w.abs().max()), 1.0)
            word = _absmax_word(dy.cuda())
            op.dy_word = word.data_ptr()
        else:
            op.set_ranges(None, None, None)
            op.dy_word = None
        op.out.data.fill_(float("nan"))
        op.forward()
        torch.cuda.synchronize()
        kf = L.last_kernel().decode()
        yv = op.out.data.cpu().double()
        op.out.grad.copy_(dy)
        src.grad.fill_(float("nan"))
        e.grads["up.weight"].fill_(float("nan"))
        op.backward()
        torch.cuda.synchronize()
        assert kf.startswith("convT_fwd_" + form), kf
        gx, gw = src.grad.cpu().double(), e.grads["up.weight"].cpu().double()
        assert torch.isfinite(yv).all() and torch.isfinite(gx).all() and torch.isfinite(gw).all(), form
        res[form] = tuple(float((a - b).norm() / b.norm()) for a, b in ((yv, y64.detach()), (gx, x64.grad), (gw, w64.grad)))
    print("convT %d->%d %s x%g: rel-L2 vs fp64 (fwd, dgrad, wgrad): h2 %s  bf3 %s" % (cin, cout, dims, mag, res["h2"], res["bf3"]))
    for i, name in enumerate(("forward", "data gradient", "weight gradient")):
        assert res["h2"][i] <= 2e-6 and res["bf3"][i] <= 2e-6, (name, res)
        assert res["h2"][i] <= 1.5 * res["bf3"][i] + 1e-7, (name, res)
