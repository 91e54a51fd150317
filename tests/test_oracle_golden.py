"""The oracle (CPU restatement) against golden vectors produced by the reference itself
(tools/make_golden.py) and the reference's own known-answer vectors."""
import random
import numpy as np
import pytest
import torch

import oracle
from oracle import network as onet
from tests.helpers import (golden, closed_form_params, closed_form_tensor, seeded_input, seeded_labels,
                           pack_kernel_mask, sha_of)

TINY = dict(patch=(16, 32, 32), cin=2, base=8, k=3, pools=[(2, 2, 2)] * 3 + [(1, 2, 2)] * 2, max_feat=32)
SPARSE_PATCH = (16, 64, 64)       # sparse training fixture: 2x2x2 = 8-voxel bottleneck (tools/make_golden.py)
HIPPO = dict(patch=(40, 56, 40), cin=1, k=3, pools=[(2, 2, 2)] * 3 + [(1, 1, 1)] * 2)


def tiny_spec():
    return oracle.make_spec(TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], 2, TINY["max_feat"])


# ------------------------------------------------------------------ a1 depth shift (bit exact)
@pytest.mark.parametrize("c", [1, 2, 3, 4, 5, 7, 12, 32, 64, 160, 896])
def test_shift_matches_reference(c):
    g = golden("shift.npz")
    x = seeded_input((2, c, 7, 2, 3), seed=100 + c)
    assert np.array_equal(oracle.depth_shift(x).numpy(), g["c%d" % c])


def test_shift_shallow_volumes():
    g = golden("shift.npz")
    assert np.array_equal(oracle.depth_shift(seeded_input((1, 10, 1, 2, 2), seed=7)).numpy(), g["d1_c10"])
    assert np.array_equal(oracle.depth_shift(seeded_input((1, 10, 2, 2, 2), seed=8)).numpy(), g["d2_c10"])


def test_shift_amount_groups():
    assert oracle.shift_amounts(64) == [c // 13 - 2 for c in range(64)]
    assert oracle.shift_amounts(1) == [-2]
    assert oracle.shift_amounts(12)[-1] == 1            # only four groups
    assert oracle.shift_amounts(896)[-1] == 2 and oracle.shift_amounts(896)[179] == -2


# ------------------------------------------------------------------ a2 conv block fwd + bwd
@pytest.mark.parametrize("tag,stride,cin,cout,shape", [("s1", (1, 1, 1), 8, 6, (6, 8, 8)),
                                                       ("s2", (2, 2, 2), 8, 12, (6, 8, 8)),
                                                       ("s122", (1, 2, 2), 5, 7, (5, 10, 6)),
                                                       ("odd", (1, 1, 1), 13, 9, (4, 7, 9))])
def test_conv_block(tag, stride, cin, cout, shape):
    g = golden("block.npz")
    shapes = {"conv.weight": (cout, cin, 1, 3, 3), "conv.bias": (cout,),
              "instnorm.weight": (cout,), "instnorm.bias": (cout,)}
    p = closed_form_params(shapes)
    for v in p.values():
        v.requires_grad_(True)
    x = seeded_input((2, cin) + shape, seed=11).requires_grad_(True)
    y = oracle.conv_block(x, p["conv.weight"], p["conv.bias"], p["instnorm.weight"], p["instnorm.bias"], stride)
    y.backward(seeded_input(tuple(y.shape), seed=12))
    np.testing.assert_allclose(y.detach().numpy(), g[tag + "_y"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g[tag + "_dx"], rtol=0, atol=2e-6)
    for n in shapes:
        np.testing.assert_allclose(p[n].grad.numpy(), g[tag + "_d_" + n], rtol=1e-5, atol=2e-5)


# ------------------------------------------------------------------ a4/a5 whole net
def test_param_names_and_order_match_reference():
    g = golden("init.npz")
    spec = tiny_spec()
    shapes = onet.param_shapes(spec)
    assert list(shapes.keys()) == [str(s) for s in g["tiny_param_order"]]
    ref_shapes = dict(zip([str(s) for s in g["tiny_names"]], [str(s) for s in g["tiny_shapes"]]))
    for n, s in shapes.items():
        assert str(tuple(s)) == ref_shapes[n]
    spec32 = oracle.make_spec(4, 32, 4)
    assert list(onet.param_shapes(spec32).keys()) == [str(s) for s in g["b32_param_order"]]
    assert len(onet.param_shapes(spec32)) == 147
    assert len(oracle.masked_names(spec32)) == 35


def test_net_tiny_forward_backward():
    g = golden("net_tiny.npz")
    spec = tiny_spec()
    shapes = onet.param_shapes(spec)
    assert [str(s) for s in g["names"]] == list(shapes.keys())
    params = closed_form_params(shapes)
    for v in params.values():
        v.requires_grad_(True)
    x = seeded_input((2, TINY["cin"]) + TINY["patch"], seed=21)
    outs = oracle.forward(spec, params, x, do_ds=True)
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), g["logits%d" % i], rtol=0, atol=2e-5)
    w = oracle.ds_weights(5)
    np.testing.assert_allclose(w, g["ds_weights"])
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=30 + i) for i, o in enumerate(outs)]
    loss = oracle.deep_supervision_loss(outs, targets, w)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    l2 = np.array([params[n].grad.double().norm().item() for n in shapes])
    np.testing.assert_allclose(l2, g["grad_l2"], rtol=2e-3, atol=1e-6)
    for key in g.files:
        if key.startswith("grad::"):
            ref = g[key]
            got = params[key[6:]].grad.numpy()
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    with torch.no_grad():
        full = oracle.forward(spec, params, x, do_ds=False)
    assert abs(full.double().sum().item() - float(g["logits_nods_sum"])) < 1e-2


def test_forced_branch_evaluation_removes_the_kink_noise():
    """oracle.Branches (test infrastructure of the GPU gradient checks): (1) an evaluation that replays its OWN recorded LeakyReLU
    masks and pooling arg-maxes is the plain evaluation, bit for bit, values and gradients; (2) an fp32 evaluation that replays the
    fp64 evaluation's decisions has gradients within 1e-4 (relative L2 per tensor; measured 2e-5) of the fp64 ones -- rounding only.
    (On this tiny net the plain fp32 evaluation happens to take the same decisions; on the 64^3 nets of tests/test_gpu_configs.py it
    does not, and sits per cents away.)"""
    spec = tiny_spec()
    shapes = onet.param_shapes(spec)
    base = closed_form_params(shapes)
    x = seeded_input((2, TINY["cin"]) + TINY["patch"], seed=21)
    w = oracle.ds_weights(5)

    def run(dtype, branches):
        params = {n: v.detach().to(dtype).clone().requires_grad_(True) for n, v in base.items()}
        outs = oracle.forward(spec, params, x.to(dtype), branches=branches)
        targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=30 + i) for i, o in enumerate(outs)]
        oracle.deep_supervision_loss(outs, targets, w).backward()
        return outs, params

    rec = oracle.Branches(recording=True)
    outs_a, par_a = run(torch.float32, rec)
    assert len(rec.lrelu) == sum(1 for n in shapes if n.endswith(".conv.weight")) and len(rec.pool) > 0
    rec.recording = False
    outs_b, par_b = run(torch.float32, rec)
    outs_c, par_c = run(torch.float32, None)
    for a_, b_, c_ in zip(outs_a, outs_b, outs_c):
        assert torch.equal(a_, b_) and torch.equal(a_, c_)
    for n in shapes:
        assert torch.equal(par_a[n].grad, par_c[n].grad)
        assert torch.allclose(par_a[n].grad, par_b[n].grad, rtol=0, atol=1e-6 * max(1.0, par_a[n].grad.abs().max().item()))
    rec64 = oracle.Branches(recording=True)
    _, par64 = run(torch.float64, rec64)
    rec64.recording = False
    _, par32f = run(torch.float32, rec64)

    def rel(p):
        return {n: ((p[n].grad.double() - par64[n].grad).norm() / par64[n].grad.norm()).item() for n in shapes if par64[n].grad.norm() > 1e-9}
    forced, plain = rel(par32f), rel(par_c)
    print("[branches] worst relative L2 from fp64: forced %.2e plain fp32 %.2e" % (max(forced.values()), max(plain.values())))
    assert max(forced.values()) <= 1e-4


def test_net64_sparse_forward():
    """64^3, base 32, Cin 4, K 4, density 0.2 masks from random.seed(0) (SURVEY golden #4)."""
    g = golden("net64.npz")
    spec = oracle.make_spec(4, 32, 4)
    shapes = onet.param_shapes(spec)
    params = closed_form_params(shapes)
    names = oracle.masked_names(spec)
    assert names == [str(s) for s in g["mask_names"]]
    random.seed(0)
    masks = oracle.uniform_kernel_masks(shapes, names, 0.2)
    assert [sha_of(pack_kernel_mask(masks[n])) for n in names] == [str(s) for s in g["mask_sha"]]
    assert [int(masks[n].sum().item()) for n in names] == list(g["mask_nnz"])
    for n in names:
        params[n] = params[n] * masks[n]
    x = seeded_input((1, 4, 64, 64, 64), seed=41)
    with torch.no_grad():
        outs = oracle.forward(spec, params, x)
    np.testing.assert_allclose(outs[0][0, :, 32].numpy(), g["slice_d32"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(outs[0][0, :, :, 5].numpy(), g["slice_h5"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(outs[1].numpy()[0, :, ::4], g["logits1"], rtol=0, atol=5e-5)
    for i, o in enumerate(outs):
        assert abs(o.double().abs().sum().item() - float(g["abs%d" % i])) <= 1e-5 * float(g["abs%d" % i])


def test_hippocampus_config1_vs_reference():
    """BASELINE config 1: 40x56x40, Cin 1, K 3, pools [[2,2,2]]*3 + [[1,1,1]]*2, density 1.0.
    The [1,1,1] stages make 'strided' convs, transposed convs and poolings with unit kernels and 5x7 planes."""
    g = golden("net_hippo.npz")
    spec = oracle.make_spec(HIPPO["cin"], 32, HIPPO["k"], HIPPO["pools"])
    shapes = onet.param_shapes(spec)
    assert list(shapes.keys()) == [str(s) for s in g["names"]]
    params = closed_form_params(shapes)
    for v in params.values():
        v.requires_grad_(True)
    x = seeded_input((1, HIPPO["cin"]) + HIPPO["patch"], seed=81)
    outs = oracle.forward(spec, params, x)
    assert [list(o.shape) for o in outs] == g["out_shapes"].tolist()
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), HIPPO["k"], seed=90 + i) for i, o in enumerate(outs)]
    loss = oracle.deep_supervision_loss(outs, targets, oracle.ds_weights(5))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    for i, o in enumerate(outs):
        od = o.detach().numpy()
        ref = g["b32_logits%d" % i]
        np.testing.assert_allclose(od[:, :, ::2, ::2, ::2] if i == 0 else od, ref, rtol=0, atol=5e-5)
        assert abs(o.detach().double().abs().sum().item() - float(g["b32_abs%d" % i])) <= 1e-5 * float(g["b32_abs%d" % i])
    l2 = np.array([params[n].grad.double().norm().item() for n in shapes])
    np.testing.assert_allclose(l2, g["grad_l2"], rtol=2e-3, atol=1e-6)
    for key in g.files:
        if key.startswith("grad::"):
            ref = g[key]
            got = params[key[6:]].grad.numpy()[:8]
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), key
    # the reference trainer's hard-coded width (nnUNetTrainer_simple.py:296), forward only
    spec48 = oracle.make_spec(HIPPO["cin"], 48, HIPPO["k"], HIPPO["pools"])
    p48 = closed_form_params(onet.param_shapes(spec48))
    with torch.no_grad():
        o = oracle.forward(spec48, p48, x, do_ds=False)
    np.testing.assert_allclose(o.numpy()[:, :, ::2, ::2, ::2], g["b48_logits"], rtol=0, atol=5e-5)
    assert abs(o.double().abs().sum().item() - float(g["b48_abs"])) <= 1e-5 * float(g["b48_abs"])


@pytest.mark.parametrize("dens", [0.1, 0.5])
def test_amos_config5_densities_vs_reference(dens):
    """BASELINE config 5: AMOS-shaped net (Cin 1, K 16, base 32, 64^3) at DSFF density 0.1 / 0.5: fwd + loss + bwd."""
    g = golden("net_amos.npz")
    tag = "d%s" % dens
    spec = oracle.make_spec(1, 32, 16)
    shapes = onet.param_shapes(spec)
    assert list(shapes.keys()) == [str(s) for s in g["names"]]
    params = closed_form_params(shapes)
    names = oracle.masked_names(spec)
    random.seed(0)
    masks = oracle.uniform_kernel_masks(shapes, names, dens)
    assert [sha_of(pack_kernel_mask(masks[n])) for n in names] == [str(s) for s in g[tag + "_mask_sha"]]
    for n in names:
        params[n] = params[n] * masks[n]
    for n in params:
        params[n] = params[n].detach().requires_grad_(True)
    x = seeded_input((1, 1, 64, 64, 64), seed=141)
    outs = oracle.forward(spec, params, x)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), 16, seed=150 + i) for i, o in enumerate(outs)]
    loss = oracle.deep_supervision_loss(outs, targets, oracle.ds_weights(5))
    loss.backward()
    assert abs(loss.item() - float(g[tag + "_loss"])) < 2e-5
    np.testing.assert_allclose(outs[0].detach().numpy()[0, :, 31, ::2, ::2], g[tag + "_slice_d31"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(outs[3].detach().numpy(), g[tag + "_logits3"], rtol=0, atol=5e-5)
    for i, o in enumerate(outs):
        assert abs(o.detach().double().abs().sum().item() - float(g[tag + "_abs%d" % i])) <= 1e-5 * float(g[tag + "_abs%d" % i])
    l2 = np.array([params[n].grad.double().norm().item() for n in shapes])
    np.testing.assert_allclose(l2, g[tag + "_grad_l2"], rtol=2e-3, atol=1e-6)
    for key in g.files:
        if key.startswith(tag + "_grad::"):
            ref = g[key]
            got = params[key[len(tag) + 7:]].grad.numpy()
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), key


def test_width48_net_vs_reference():
    """The width the reference trainer hard-codes (nnUNetTrainer_simple.py:296), end to end: 4 x 64^3, K 4, DSFF density 0.2
    (with the `shape[0] == 48 => 0.2` quirk of Masking.init, core_channel.py:147-151): masks, logits, loss and gradients of the
    oracle against the reference's (tests/golden/net_w48.npz, tools/make_golden.py gen_net_w48)."""
    g = golden("net_w48.npz")
    spec = oracle.make_spec(4, 48, 4)
    shapes = onet.param_shapes(spec)
    assert list(shapes.keys()) == [str(s) for s in g["names"]]
    params = closed_form_params(shapes)
    names = oracle.masked_names(spec)
    assert names == [str(s) for s in g["mask_names"]]
    random.seed(0)
    masks = oracle.uniform_kernel_masks(shapes, names, 0.2)
    assert [sha_of(pack_kernel_mask(masks[n])) for n in names] == [str(s) for s in g["mask_sha"]]
    assert [int(masks[n].sum().item()) for n in names] == g["mask_nnz"].tolist()
    for n in names:
        params[n] = params[n] * masks[n]
    for n in params:
        params[n] = params[n].detach().requires_grad_(True)
    x = seeded_input((1, 4, 64, 64, 64), seed=241)
    outs = oracle.forward(spec, params, x)
    assert [list(o.shape) for o in outs] == g["out_shapes"].tolist()
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), 4, seed=250 + i) for i, o in enumerate(outs)]
    loss = oracle.deep_supervision_loss(outs, targets, oracle.ds_weights(5))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    np.testing.assert_allclose(outs[0].detach().numpy()[0, :, 31, ::2, ::2], g["slice_d31"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(outs[0].detach().numpy()[0, :, ::2, 7, ::2], g["slice_h7"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(outs[2].detach().numpy(), g["logits2"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(outs[3].detach().numpy(), g["logits3"], rtol=0, atol=5e-5)
    for i, o in enumerate(outs):
        assert abs(o.detach().double().abs().sum().item() - float(g["abs%d" % i])) <= 1e-5 * float(g["abs%d" % i])
    l2 = np.array([params[n].grad.double().norm().item() for n in shapes])
    np.testing.assert_allclose(l2, g["grad_l2"], rtol=2e-3, atol=1e-6)
    for key in g.files:
        if key.startswith("grad::"):
            ref = g[key]
            got = params[key[6:]].grad.numpy()
            got = got[:8] if got.ndim > 1 else got
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), key


# ------------------------------------------------------------------ a9-a13 DSFF
@pytest.mark.parametrize("base", [32, 48])
@pytest.mark.parametrize("dens", [0.1, 0.2, 0.5])
def test_uniform_masks_bit_exact(base, dens):
    g = golden("masks.npz")
    tag = "b%d_d%s" % (base, dens)
    spec = oracle.make_spec(4, base, 4)
    shapes = onet.param_shapes(spec)
    names = oracle.masked_names(spec)
    assert names == [str(s) for s in g[tag + "_names"]]
    random.seed(0)
    masks = oracle.uniform_kernel_masks(shapes, names, dens)
    assert [sha_of(pack_kernel_mask(masks[n])) for n in names] == [str(s) for s in g[tag + "_sha"]]
    assert [int(masks[n].sum().item()) for n in names] == list(g[tag + "_nnz"])
    if tag + "_loc4.0" in g.files:
        assert np.array_equal(pack_kernel_mask(masks["loc4.0.0.blocks.0.conv.weight"]), g[tag + "_loc4.0"])
        assert np.array_equal(pack_kernel_mask(masks["up0.0.weight"]), g[tag + "_up0.0"])


def test_death_rate_schedule_bit_exact():
    g = golden("masks.npz")
    for key, tmax, n in (("death_rate_T10", 10, 12), ("death_rate_T250k", 250000, 5)):
        d = oracle.CosineDeathRate(0.5, tmax)
        seq = []
        for _ in range(n):
            d.step()
            seq.append(d.get_dr())
        assert np.array_equal(np.array(seq, dtype=np.float64), g[key])


@pytest.mark.parametrize("tag,shp", [("l1_133", (320, 896, 1, 3, 3)), ("l1_222", (64, 32, 2, 2, 2)),
                                     ("l1_122", (16, 24, 1, 2, 2))])
def test_kernel_l1_association_order(tag, shp):
    g = golden("masks.npz")
    assert np.array_equal(oracle.kernel_l1(closed_form_tensor(shp, 3, "conv")).numpy(), g[tag])


def _train_two_steps_tiny():
    """nnUNetTrainer_simple.run_iteration (non-AMP) x2 on the tiny net with DSFF, all oracle code."""
    spec = tiny_spec()
    shapes = onet.param_shapes(spec)
    params = closed_form_params(shapes)
    names = oracle.masked_names(spec)
    random.seed(5)
    mom = {}
    st = oracle.DsffState(params, names, density=0.3, death_rate=0.5, t_max=10, update_frequency=2,
                          momentum_buffers=mom)
    masks0 = {n: st.masks[n].clone() for n in names}
    x = seeded_input((2, TINY["cin"]) + SPARSE_PATCH, seed=21)
    w = oracle.ds_weights(5)
    rec = dict(losses=[], total_norm=[], death_rate=[])
    for it in range(2):
        leaves = {n: p.detach().clone().requires_grad_(True) for n, p in params.items()}
        outs = oracle.forward(spec, leaves, x)
        targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=30 + i)
                   for i, o in enumerate(outs)]
        loss = oracle.deep_supervision_loss(outs, targets, w)
        loss.backward()
        grads = {n: leaves[n].grad for n in leaves}
        if it == 0:
            rec["logits0_it0"] = outs[0].detach().numpy()[:, :, :, ::2, ::2]
            rec["logits0_it0_sum"] = outs[0].detach().double().sum().item()
            rec["grad_l2_it0"] = np.array([grads[n].double().norm().item() for n in shapes])
        tn = oracle.clip_and_sgd_step(params, grads, mom, lr=1e-2)
        st.step()
        rec["losses"].append(loss.item())
        rec["total_norm"].append(tn.item())
        rec["death_rate"].append(st.death_rate)
    return spec, shapes, names, params, masks0, st, rec


def test_sparse_train_steps_and_prune_grow_bit_exact_masks():
    g = golden("net_sparse_tiny.npz")
    spec, shapes, names, params, masks0, st, rec = _train_two_steps_tiny()
    assert tuple(g["patch"]) == SPARSE_PATCH
    assert names == [str(s) for s in g["names"]]
    assert abs(rec["logits0_it0_sum"] - float(g["logits0_it0_sum"])) < 0.05
    for n in names:
        assert np.array_equal(pack_kernel_mask(masks0[n]), g["mask0::" + n]), n
    np.testing.assert_allclose(rec["logits0_it0"], g["logits0_it0"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(rec["grad_l2_it0"], g["grad_l2_it0"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(rec["losses"], g["losses"], rtol=0, atol=2e-5)
    for it in range(2):
        assert abs(rec["total_norm"][it] - float(g["total_norm_it%d" % it])) <= 1e-3 * float(g["total_norm_it%d" % it])
        assert rec["death_rate"][it] == float(g["death_rate_it%d" % it])
    # masks after the magnitude-death / random-growth update: bit exact
    for n in names:
        assert np.array_equal(pack_kernel_mask(st.masks[n]), g["mask2::" + n]), n
    for key in g.files:
        if key.startswith("param_after::"):
            np.testing.assert_allclose(params[key[13:]].numpy(), g[key], rtol=0, atol=2e-5)
    got_abs = np.array([params[n].double().abs().sum().item() for n in shapes])
    np.testing.assert_allclose(got_abs, g["param_abs_after"], rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ a18 loss
@pytest.mark.parametrize("tag,batch_dice", [("sample", False), ("batch", True)])
def test_loss_value_and_grad(tag, batch_dice):
    g = golden("loss.npz")
    k = 4
    shapes = [(2, k, 8, 12, 10), (2, k, 4, 6, 5), (2, k, 2, 3, 5), (2, k, 1, 3, 5)]
    logits = [seeded_input(s, seed=50 + i).mul(2.0).requires_grad_(True) for i, s in enumerate(shapes)]
    targets = [seeded_labels((s[0], 1) + s[2:], k, seed=60 + i) for i, s in enumerate(shapes)]
    loss = oracle.deep_supervision_loss(logits, targets, oracle.ds_weights(5), batch_dice)
    assert abs(loss.item() - float(g[tag + "_loss"])) < 1e-6
    loss.backward()
    for i, l in enumerate(logits):
        np.testing.assert_allclose(l.grad.numpy(), g[tag + "_g%d" % i], rtol=0, atol=1e-7)


def test_optimizer_step_matches_torch_sgd():
    """The third-party arithmetic (torch clip_grad_norm_ + SGD nesterov) vs the oracle's spelled-out rule."""
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(11))]
    opt = torch.optim.SGD(ps, 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    mine = {str(i): p.detach().clone() for i, p in enumerate(ps)}
    mom = {}
    for step in range(3):
        grads = [torch.randn_like(p) * (30.0 if step == 1 else 0.1) for p in ps]
        for p, gr in zip(ps, grads):
            p.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_(ps, 12)
        opt.step()
        tn2 = oracle.clip_and_sgd_step(mine, {str(i): gr.clone() for i, gr in enumerate(grads)}, mom, 1e-2)
        assert abs(tn.item() - tn2.item()) < 1e-5
        for i, p in enumerate(ps):
            np.testing.assert_allclose(mine[str(i)].numpy(), p.detach().numpy(), rtol=0, atol=1e-6)
    assert abs(oracle.poly_lr(10, 1000, 1e-2) - 1e-2 * (1 - 10 / 1000) ** 0.9) < 1e-18


# ------------------------------------------------------------------ a14-a17 sliding window
def test_steps_reference_known_answers():
    """Verbatim vectors of reference tests/test_steps_for_sliding_window_prediction.py:60-163."""
    cs = oracle.compute_steps
    for step in (1, 0.125, 0.5):
        assert cs((24, 845, 321), (24, 845, 321), step) == [[0], [0], [0]]
        assert cs((123, 143), (123, 143), step) == [[0], [0]]
    assert cs((64, 130), (128, 260), 0.5) == [[0, 32, 64], [0, 65, 130]]
    assert cs((64, 130), (128, 260), 0.85) == [[0, 32, 64], [0, 65, 130]]
    assert cs((64, 130), (128, 260), 1) == [[0, 64], [0, 130]]
    assert cs((128, 128, 128), (146, 176, 148), 0.5) == [[0, 18], [0, 48], [0, 20]]
    assert cs((80, 192, 160), (130, 320, 244), 0.5) == [[0, 25, 50], [0, 64, 128], [0, 42, 84]]
    assert cs((80, 192, 160), (130, 320, 244), 0.75) == [[0, 50], [0, 128], [0, 84]]
    assert cs((128, 128, 128), (424, 456, 456), 0.5) == [[0, 59, 118, 178, 237, 296],
                                                         [0, 55, 109, 164, 219, 273, 328],
                                                         [0, 55, 109, 164, 219, 273, 328]]
    assert cs((40, 56, 40), (40, 56, 40), 0.5) == [[0], [0], [0]]
    assert cs((64, 192, 192), (94, 308, 308), 0.5) == [[0, 30], [0, 58, 116], [0, 58, 116]]


def test_steps_properties_random():
    """Property checks of reference tests/...:25-58,165-181 (2000 draws)."""
    rng = np.random.RandomState(0)
    for _ in range(2000):
        dim = rng.choice((2, 3))
        patch = tuple(int(v) for v in rng.randint(16, 1024, dim))
        image = tuple(max(int(rng.randint(p // 2, p * 10)), p) for p in patch)
        step = float(rng.uniform(0.01, 1))
        steps = oracle.compute_steps(patch, image, step)
        for d in range(dim):
            s = steps[d]
            assert s[0] == 0 and s[-1] + patch[d] == image[d]
            assert all(s[i + 1] <= s[i] + patch[d] for i in range(len(s) - 1))
            assert all(s[i] + np.ceil(patch[d] * step) >= s[i + 1] for i in range(len(s) - 1))


@pytest.mark.parametrize("ps", [(64, 64, 64), (128, 128, 128), (16, 32, 32), (40, 56, 40)])
def test_gaussian_map(ps):
    g = golden("sliding.npz")
    m = oracle.gaussian_map(ps)
    tag = "g%dx%dx%d" % ps
    c = [i // 2 for i in ps]
    assert np.array_equal(m[:, c[1], c[2]], g[tag + "_line0"])
    assert np.array_equal(m[c[0], c[1], :], g[tag + "_line2"])
    st = g[tag + "_stats"]
    assert m.min() == st[0] and m.max() == st[1] and abs(m.astype(np.float64).sum() - st[2]) < 1e-6 * st[2]
    if tag + "_full" in g.files:
        assert np.array_equal(m, g[tag + "_full"])


@pytest.mark.parametrize("tag,kw", [("tta", dict(do_mirroring=True, mirror_axes=(0, 1, 2))),
                                    ("notta", dict(do_mirroring=False, mirror_axes=(0, 1, 2))),
                                    ("tta01", dict(do_mirroring=True, mirror_axes=(0, 1)))])
def test_predict_tiled_matches_reference(tag, kw):
    g = golden("sliding.npz")
    spec = tiny_spec()
    params = closed_form_params(onet.param_shapes(spec))
    vol = seeded_input((TINY["cin"], 13, 50, 70), seed=71).numpy()

    def net_fn(t):
        with torch.no_grad():
            return torch.softmax(oracle.forward(spec, params, t, do_ds=False), 1)

    seg, probs = oracle.predict_tiled(net_fn, vol, TINY["k"], TINY["patch"], 0.5, use_gaussian=True, **kw)
    assert seg.shape == (13, 50, 70) and seg.dtype == np.int64
    ref_seg = g["pred_%s_seg" % tag].astype(np.int64)
    assert oracle.hard_dice(seg, ref_seg, 1) >= 1 - 1e-3 and (seg != ref_seg).mean() < 1e-3
    np.testing.assert_allclose(probs[:, 6, ::2, ::2], g["pred_%s_probs_slice" % tag], rtol=0, atol=1e-5)
    np.testing.assert_allclose(probs.astype(np.float64).sum(axis=(1, 2, 3)), g["pred_%s_probs_sum" % tag], rtol=1e-5)


def test_hard_dice():
    g = golden("dice.npz")
    a, b = g["a"].astype(np.int64), g["b"].astype(np.int64)
    got = [oracle.hard_dice(a, b, l) for l in range(1, 4)]
    np.testing.assert_allclose(got, g["dice"], rtol=0, atol=1e-12)
    assert abs(oracle.hard_dice(np.array([0, 1, 1, 0]), np.array([0, 1, 0, 0])) - float(g["dice_small"])) < 1e-12


# ------------------------------------------------------------------ N1 fold ensemble + export
@pytest.mark.parametrize("tag", ["plain", "transposed", "regions"])
def test_export_matches_reference(tag):
    """oracle.export_segmentation against the uint8 volume the reference hands to its NIfTI writer
    (segmentation_export.py:118-148), ensemble average as predict.py:282-296."""
    g = golden("export.npz")
    folds = [g["fold%d" % i] for i in range(3)]
    total = oracle.ensemble_softmax(folds)
    tb = [int(v) for v in g[tag + "_tb"]]
    size = [total.shape[1 + i] for i in tb]
    props = {'size_after_cropping': np.array(size), 'original_size_of_raw_data': np.array([size[0] + 3, size[1] + 1, size[2] + 4]),
             'crop_bbox': [[2, 2 + size[0]], [0, size[1]], [3, 3 + size[2]]]}
    regions = tuple(int(v) for v in g[tag + "_regions"]) if tag + "_regions" in g.files else None
    seg = oracle.export_segmentation(total, props, tb, regions)
    assert seg.dtype == np.uint8 and np.array_equal(seg, g[tag + "_seg"])


# ------------------------------------------------------------------------------------------------------------------
# ablation networks unetpp_d_313 / unetpp_d_331 (conv kernel (3,1,3) / (3,3,1), no shift)
VARIANT = dict(patch=(16, 16, 64), cin=2, base=8, k=3, pools=[(2, 2, 2), (2, 2, 2), (1, 2, 2), (2, 1, 2), (1, 1, 2)], max_feat=32)


@pytest.mark.parametrize("var", ["313", "331"])
def test_conv_variants_vs_reference(var):
    g = golden("net_variants.npz")
    V = VARIANT
    spec = oracle.make_spec(V["cin"], V["base"], V["k"], V["pools"], 2, V["max_feat"], conv_variant=var)
    shapes = oracle.param_shapes(spec)
    assert list(shapes.keys()) == [str(n) for n in g[var + "_names"]]
    assert [str(shapes[n]) for n in shapes] == [str(s) for s in g[var + "_shapes"]]
    params = {n: p.requires_grad_(True) for n, p in closed_form_params(shapes).items()}
    x = seeded_input((2, V["cin"]) + V["patch"], seed=121)
    outs = oracle.forward(spec, params, x)
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g[var + "_logits%d" % i])
        got = o.detach()[..., ::2, ::2] if i == 0 else o.detach()
        assert (got - ref).abs().max().item() < 1e-5, "logits %d" % i
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), V["k"], seed=130 + i) for i, o in enumerate(outs)]
    loss = oracle.deep_supervision_loss(outs, targets, oracle.ds_weights(5))
    assert abs(loss.item() - float(g[var + "_loss"])) < 1e-5
    loss.backward()
    for n, ref in zip(shapes, g[var + "_grad_l2"]):
        got = params[n].grad.double().norm().item()
        assert abs(got - ref) <= 2e-4 * max(ref, 1e-3), n
    for key in g.files:
        if key.startswith(var + "_grad::"):
            n = key.split("::", 1)[1]
            ref = torch.from_numpy(g[key])
            assert (params[n].grad - ref).abs().max().item() <= 2e-4 * max(ref.abs().max().item(), 1e-3), n


def test_nodff_unet_graph_vs_reference_golden():
    """the 'shiftConvPP_nodff' ablation (reference unetpp_d_nodff.py:171-378): state-dict names and
    shapes, the five logits, the deep-supervision loss, every gradient norm and the Masking selection of the oracle's 'unet'
    graph against the golden produced by the reference module (tools/make_golden.py nodff)."""
    import oracle
    from tests.helpers import golden, closed_form_params, seeded_input, seeded_labels
    g = golden("net_nodff.npz")
    pools = [(2, 2, 2)] * 3 + [(1, 2, 2)] * 2
    spec = oracle.make_spec(2, 8, 3, pools, 2, 32, shift_size=3, graph="unet")
    shapes = oracle.param_shapes(spec)
    assert list(shapes.keys()) == [str(s) for s in g["names"]]
    assert [str(tuple(v)) for v in shapes.values()] == [str(s) for s in g["shapes"]]
    assert oracle.masked_names(spec) == [str(s) for s in g["masked_names"]]
    leaves = {n: p.clone().requires_grad_(True) for n, p in closed_form_params(shapes).items()}
    x = seeded_input((2, 2, 16, 32, 32), seed=221)
    outs = oracle.forward(spec, leaves, x)
    assert [list(o.shape) for o in outs] == g["out_shapes"].tolist()
    for i, o in enumerate(outs):
        got = o.detach().numpy()
        assert np.abs((got[..., ::2, ::2] if i == 0 else got) - g["logits%d" % i]).max() <= 1e-6
        assert abs(o.detach().double().sum().item() - float(g["sum%d" % i])) <= 1e-4
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), 3, seed=230 + i) for i, o in enumerate(outs)]
    loss = oracle.deep_supervision_loss(outs, targets, g["ds_weights"], False)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    loss.backward()
    l2 = np.array([0.0 if leaves[n].grad is None else leaves[n].grad.double().norm().item() for n in shapes])
    np.testing.assert_allclose(l2, g["grad_l2"], rtol=1e-4, atol=1e-7)
    for key in g.files:
        if key.startswith("grad::"):
            assert np.abs(leaves[key[6:]].grad.numpy() - g[key]).max() <= 1e-6 * max(1.0, np.abs(g[key]).max())


def _grad_growth_fixture(tag):
    """the recipe of tools/make_golden.py:gen_grad_growth: tiny net, closed-form weights, closed-form heavy-tailed gradients"""
    from tests.helpers import closed_form_tensor
    g = golden("grad_growth.npz")
    pools = [(2, 2, 2)] * 3 + [(1, 2, 2)] * 2
    spec = oracle.make_spec(2, 8, 3, pools, 2, 32)
    shapes = oracle.param_shapes(spec)
    params = closed_form_params(shapes)
    grads = {}
    for i, (n, shp) in enumerate(shapes.items()):
        t = closed_form_tensor(tuple(shp), 200 + i, "conv" if len(shp) > 1 else "bias")
        grads[n] = (t * (1.0 + 3.0 * t.abs())).clone()
    names = [str(s_) for s_ in g[tag + "_names"]]

    def unpack(key, n):
        shp = shapes[n]
        return np.unpackbits(g[key + "::" + n])[:shp[0] * shp[1]].reshape(shp[0], shp[1]).astype(np.uint8)
    before = {n: unpack(tag + "_before", n) for n in names}
    after = {n: unpack(tag + "_after", n) for n in names}
    num_death = {n: int(g[tag + "_num_death::" + n][0]) for n in names}
    return g, shapes, params, grads, names, before, after, num_death


@pytest.mark.parametrize("tag", ["raw", "clip"])
def test_gradient_growth_matches_reference_truncate_weights(tag):
    """oracle.kernel_death + oracle.kernel_grad_growth reproduce the masks of the reference's truncate_weights with
    growth_mode='gradient' (core_channel.py:556-611, :771-790) bit for bit, also behind clip_grad_norm_."""
    g, shapes, params, grads, names, before, after, num_death = _grad_growth_fixture(tag)
    if tag == "clip":
        tot = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(t) for t in grads.values()]))
        np.testing.assert_allclose(float(tot), g["clip_total_norm"][0], rtol=1e-6)
        coef = torch.clamp(float(g["clip_max_norm"][0]) / (tot + 1e-6), max=1.0)
        grads = {n: t * coef for n, t in grads.items()}
    for n in names:
        shp = shapes[n]
        m = torch.from_numpy(before[n]).float().reshape(shp[0], shp[1], 1, 1, 1).expand(shp).clone()
        w = params[n] * m
        m, prune_num = oracle.kernel_death(m, w, 0.3)
        assert prune_num == num_death[n]
        m = oracle.kernel_grad_growth(m, grads[n], prune_num)
        km = (m.reshape(shp[0], shp[1], -1).sum(-1) > 0).numpy().astype(np.uint8)
        assert np.array_equal(km, after[n]), n
