"""Whole-network parity of the MI355X engine: against the reference's golden vectors (tests/golden, produced by
tools/make_golden.py from the reference itself) and against the CPU oracle on the same seeded inputs.
Bars (BASELINE.json north_star): |dlogit| <= 1e-4, Dice >= 1 - 1e-3, DSFF mask indices bit exact."""
import os
import random
import numpy as np
import pytest
import torch
from torch import nn

import oracle
from oracle import network as onet
from tests.helpers import (golden, closed_form_params, seeded_input, seeded_labels, pack_kernel_mask, sha_of)

pytestmark = pytest.mark.gpu

TINY = dict(patch=(16, 32, 32), cin=2, base=8, k=3, pools=[(2, 2, 2)] * 3 + [(1, 2, 2)] * 2, max_feat=32)
SPARSE_PATCH = (16, 64, 64)       # sparse training fixture: 2x2x2 = 8-voxel bottleneck (tools/make_golden.py)
HIPPO = dict(patch=(40, 56, 40), cin=1, k=3, pools=[(2, 2, 2)] * 3 + [(1, 1, 1)] * 2)


def build_net(patch, cin, base, k, pools, max_feat=None, **extra):
    from e2enet_medical_amd.network_architecture.unetpp_d import Generic_UNetPlusPlus
    from e2enet_medical_amd.network_architecture.initialization import InitWeights_He
    net = Generic_UNetPlusPlus(patch, cin, base, k, 5, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True},
                               nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU,
                               {'negative_slope': 1e-2, 'inplace': True}, True, False, lambda x: x, InitWeights_He(1e-2),
                               pools, None, False, True, True, max_num_features=max_feat, **extra)
    return net.cuda()


def load_closed_form(net):
    shapes = {n: tuple(p.shape) for n, p in net.named_parameters()}
    params = closed_form_params(shapes)
    with torch.no_grad():
        for n, p in net.named_parameters():
            p.copy_(params[n])
    return shapes, params


def tiny_net(patch=None):
    net = build_net(patch or TINY["patch"], TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], TINY["max_feat"])
    shapes, params = load_closed_form(net)
    return net, shapes, params


def test_native_library_is_loaded():
    from e2enet_medical_amd._lib import lib, LIB_PATH
    from e2enet_medical_amd._lib import ABI_VERSION
    assert lib().abi_version() == ABI_VERSION
    with open("/proc/self/maps") as f:
        assert any(LIB_PATH in line for line in f), "libe2e_hip.so is not mapped into this process"


def test_cpu_input_fails_loudly():
    net, _, _ = tiny_net()
    with pytest.raises(RuntimeError):
        net(torch.zeros((1, TINY["cin"]) + TINY["patch"]))


def test_tiny_forward_backward_vs_reference_golden():
    g = golden("net_tiny.npz")
    net, shapes, _ = tiny_net()
    x = seeded_input((2, TINY["cin"]) + TINY["patch"], seed=21).cuda()
    outs = net(x)                                          # autograd path (one node for the whole net)
    assert len(outs) == 4
    for i, o in enumerate(outs):
        err = np.abs(o.detach().cpu().numpy() - g["logits%d" % i]).max()
        assert err <= 1e-4, "logits%d: %g" % (i, err)
    from e2enet_medical_amd.training.loss_functions.dice_loss import DC_and_CE_loss
    from e2enet_medical_amd.training.loss_functions.deep_supervision import MultipleOutputLoss2
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=30 + i).cuda() for i, o in enumerate(outs)]
    loss_fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), g["ds_weights"])
    loss = loss_fn(outs, targets)
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    loss.backward()
    names = [str(s) for s in g["names"]]
    got_l2 = np.array([net.get_parameter(n).grad.double().norm().item() for n in names])
    np.testing.assert_allclose(got_l2, g["grad_l2"], rtol=5e-3, atol=2e-6)
    for key in g.files:
        if key.startswith("grad::"):
            ref = g[key]
            got = net.get_parameter(key[6:]).grad.cpu().numpy()
            assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), key
    net.do_ds = False
    with torch.no_grad():
        full = net(x)
    assert full.shape == outs[0].shape
    assert abs(full.double().sum().item() - float(g["logits_nods_sum"])) < 5e-2


def test_tiny_engine_fastpath_matches_oracle_all_grads():
    """Engine fast path (fused loss kernels + backward) against the oracle's autograd for every parameter."""
    net, shapes, params = tiny_net()
    spec = oracle.make_spec(TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], 2, TINY["max_feat"])
    x = seeded_input((2, TINY["cin"]) + TINY["patch"], seed=77)
    eng = net.engine(x.cuda())
    outs = eng.forward(x.cuda(), True)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=80 + i) for i, o in enumerate(outs)]
    w = oracle.ds_weights(5)
    for bd in (False, True):
        loss = eng.loss_backward([t.cuda() for t in targets], w, batch_dice=bd)
        leaves = {n: p.clone().requires_grad_(True) for n, p in params.items()}
        ref = oracle.forward(spec, leaves, x)
        ref_loss = oracle.deep_supervision_loss(ref, targets, w, bd)
        ref_loss.backward()
        assert abs(loss.item() - ref_loss.item()) < 2e-5
        for o, r in zip(outs, ref):
            assert (o.cpu() - r.detach()).abs().max() <= 1e-4
        for n in shapes:
            rg = leaves[n].grad
            err = (eng.grads[n].cpu() - rg).abs().max().item()
            assert err <= 2e-4 * max(1.0, rg.abs().max().item()) + 1e-6, (n, err)


@pytest.mark.parametrize("shift_size", [1, 3, 7])
def test_shift_size_variants_match_oracle(shift_size):
    """the other restricted-shift sizes (the reference's comment at unetpp_d.py:89 lists 3/7/11; 1 is the
    'noshift' ablation) only change the per-channel depth offset table: logits and all gradients vs the oracle."""
    net = build_net(TINY["patch"], TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], TINY["max_feat"], shift_size=shift_size)
    shapes, params = load_closed_form(net)
    spec = oracle.make_spec(TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], 2, TINY["max_feat"], shift_size=shift_size)
    x = seeded_input((2, TINY["cin"]) + TINY["patch"], seed=78)
    eng = net.engine(x.cuda())
    outs = eng.forward(x.cuda(), True)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=90 + i) for i, o in enumerate(outs)]
    w = oracle.ds_weights(5)
    loss = eng.loss_backward([t.cuda() for t in targets], w, batch_dice=False)
    leaves = {n: p.clone().requires_grad_(True) for n, p in params.items()}
    ref = oracle.forward(spec, leaves, x)
    ref_loss = oracle.deep_supervision_loss(ref, targets, w, False)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-5
    for o, r in zip(outs, ref):
        assert (o.cpu() - r.detach()).abs().max() <= 1e-4
    for n in shapes:
        rg = leaves[n].grad
        err = (eng.grads[n].cpu() - rg).abs().max().item()
        assert err <= 5e-4 * max(1.0, rg.abs().max().item()) + 1e-6, (n, err)   # fp32 summation order over 16k voxels


def test_net64_sparse_forward_vs_reference_golden():
    """64^3, base 32, Cin 4, K 4, DSFF density 0.2 (SURVEY golden #4): masks bit exact, logits within 1e-4."""
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    g = golden("net64.npz")
    net = build_net((64, 64, 64), 4, 32, 4, [(2, 2, 2)] * 5)
    load_closed_form(net)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        update_frequency = 1
        final_density = 0.05
    random.seed(0)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 10), growth_mode='random',
                   redistribution_mode='none', args=A())
    mask.add_module(net, sparse_init='uniform', density=0.2)
    assert list(mask.masks.keys()) == [str(s) for s in g["mask_names"]]
    assert [sha_of(pack_kernel_mask(m.cpu())) for m in mask.masks.values()] == [str(s) for s in g["mask_sha"]]
    assert [int(m.sum().item()) for m in mask.masks.values()] == list(g["mask_nnz"])
    net.eval()
    x = seeded_input((1, 4, 64, 64, 64), seed=41).cuda()
    with torch.no_grad():
        outs = net(x)
    assert np.abs(outs[0][0, :, 32].cpu().numpy() - g["slice_d32"]).max() <= 1e-4
    assert np.abs(outs[0][0, :, :, 5].cpu().numpy() - g["slice_h5"]).max() <= 1e-4
    assert np.abs(outs[1].cpu().numpy()[0, :, ::4] - g["logits1"]).max() <= 1e-4
    for i, o in enumerate(outs):
        assert abs(o.double().abs().sum().item() - float(g["abs%d" % i])) <= 2e-5 * float(g["abs%d" % i])
    # dense execution of the same masked weights gives the same logits (liveness bits only skip exact zeros)
    net.set_kernel_masks(None)
    with torch.no_grad():
        dense = net(x)
    assert (dense[0] - outs[0]).abs().max().item() <= 2e-5
    # inference-style liveness derived from the zero kernels of the weights
    net.enable_auto_sparsity(True)
    with torch.no_grad():
        auto = net(x)
    assert torch.equal(auto[0], outs[0])


def test_sparse_training_two_steps_vs_reference_golden():
    """Two reference training iterations (run_iteration, non-AMP) on the tiny net with DSFF, prune/grow at step 2:
    losses, clip norm, death rate, updated weights within tolerance; mask indices bit exact."""
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    from e2enet_medical_amd.training.fused_optim import FusedClipSGD
    g = golden("net_sparse_tiny.npz")
    net, shapes, _ = tiny_net(SPARSE_PATCH)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        update_frequency = 2
        final_density = 0.05
    random.seed(5)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 10), growth_mode='random',
                   redistribution_mode='none', args=A())
    mask.add_module(net, sparse_init='uniform', density=0.3)
    names = [str(s) for s in g["names"]]
    assert list(mask.masks.keys()) == names
    for n in names:
        assert np.array_equal(pack_kernel_mask(mask.masks[n].cpu()), g["mask0::" + n]), n
    assert tuple(g["patch"]) == SPARSE_PATCH
    x = seeded_input((2, TINY["cin"]) + SPARSE_PATCH, seed=21).cuda()
    w = oracle.ds_weights(5)
    fused = FusedClipSGD(opt, list(net.named_parameters()), 12.0)
    eng = net.engine(x)
    losses = []
    for it in range(2):
        outs = eng.forward(x, True)
        targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=30 + i).cuda() for i, o in enumerate(outs)]
        if it == 0:
            assert np.abs(outs[0].cpu().numpy()[:, :, :, ::2, ::2] - g["logits0_it0"]).max() <= 1e-4
        loss = eng.loss_backward(targets, w, batch_dice=False)
        if it == 0:
            l2 = np.array([eng.grads[n].double().norm().item() for n in shapes])
            np.testing.assert_allclose(l2, g["grad_l2_it0"], rtol=5e-3, atol=2e-6)
        fused.step(eng.grads, mask.masks)
        tn = fused.total_norm()
        assert abs(tn - float(g["total_norm_it%d" % it])) <= 1e-3 * float(g["total_norm_it%d" % it])
        mask.step(masks_already_applied=True)
        assert mask.death_rate == float(g["death_rate_it%d" % it])
        losses.append(loss.item())
    assert abs(losses[0] - g["losses"][0]) <= 5e-5 and abs(losses[1] - g["losses"][1]) <= 5e-5
    nnz = {n: int(mask.masks[n].sum().item()) for n in names}
    assert nnz == {n: int(np.unpackbits(g["mask2::" + n]).sum()) * int(np.prod(mask.masks[n].shape[-3:])) for n in names}
    sd = net.state_dict()
    got_abs = np.array([sd[n].double().abs().sum().item() for n in shapes])
    np.testing.assert_allclose(got_abs, g["param_abs_after"], rtol=5e-3, atol=1e-3)


def test_prune_grow_replay_bit_exact_masks():
    """The reference's prune/grow decision replayed on the GPU from the reference's own pre-update weights
    (golden pre_prune::*): kernel-L1 -> k-th threshold -> death -> random growth gives bit-identical mask indices."""
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    g = golden("net_sparse_tiny.npz")
    net, shapes, _ = tiny_net(SPARSE_PATCH)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        update_frequency = 2
        final_density = 0.05
    random.seed(5)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 10), growth_mode='random',
                   redistribution_mode='none', args=A())
    mask.add_module(net, sparse_init='uniform', density=0.3)
    names = [str(s) for s in g["names"]]
    for n in names:
        assert np.array_equal(pack_kernel_mask(mask.masks[n].cpu()), g["mask0::" + n]), n
    mask.step()                                                  # iteration 0: no update (update_frequency = 2)
    assert mask.death_rate == float(g["death_rate_it0"])
    with torch.no_grad():
        for n in names:
            net.get_parameter(n).copy_(torch.from_numpy(g["pre_prune::" + n]))
    assert mask.step() is True                                   # iteration 1: apply_mask, decay, truncate_weights
    assert mask.death_rate == float(g["death_rate_it1"])
    for n in names:
        assert np.array_equal(pack_kernel_mask(mask.masks[n].cpu()), g["mask2::" + n]), n
        m = mask.masks[n]
        assert float((net.get_parameter(n).detach() * (1 - m)).abs().max()) == 0.0
    for key in g.files:
        if key.startswith("param_after::") and key[13:] in names:
            assert np.array_equal(net.get_parameter(key[13:]).detach().cpu().numpy(), g[key]), key


@pytest.mark.parametrize("tag,kw", [("tta", dict(do_mirroring=True, mirror_axes=(0, 1, 2))),
                                    ("notta", dict(do_mirroring=False, mirror_axes=(0, 1, 2))),
                                    ("tta01", dict(do_mirroring=True, mirror_axes=(0, 1)))])
def test_predict_3d_vs_reference_golden(tag, kw):
    from e2enet_medical_amd.utilities.nd_softmax import softmax_helper
    g = golden("sliding.npz")
    net, _, _ = tiny_net()
    net.inference_apply_nonlin = softmax_helper
    net.eval()
    net.do_ds = False
    vol = seeded_input((TINY["cin"], 13, 50, 70), seed=71).numpy()
    seg, probs = net.predict_3D(vol, use_sliding_window=True, step_size=0.5, patch_size=TINY["patch"], use_gaussian=True,
                                all_in_gpu=False, verbose=False, mixed_precision=False, **kw)
    assert seg.shape == (13, 50, 70) and seg.dtype == np.int64 and probs.dtype == np.float32
    ref_seg = g["pred_%s_seg" % tag].astype(np.int64)
    for label in range(1, TINY["k"]):
        assert oracle.hard_dice(seg, ref_seg, label) >= 1 - 1e-3
    assert (seg != ref_seg).mean() < 1e-3
    assert np.abs(probs[:, 6, ::2, ::2] - g["pred_%s_probs_slice" % tag]).max() <= 2e-5
    np.testing.assert_allclose(probs.astype(np.float64).sum(axis=(1, 2, 3)), g["pred_%s_probs_sum" % tag], rtol=1e-5)
    # the mirrors run as one batched forward by default: bit-identical to one forward per mirror
    net.tta_batched = False
    seg1, probs1 = net.predict_3D(vol, use_sliding_window=True, step_size=0.5, patch_size=TINY["patch"], use_gaussian=True,
                                  all_in_gpu=False, verbose=False, mixed_precision=False, **kw)
    assert np.array_equal(seg, seg1) and np.array_equal(probs, probs1)


@pytest.mark.parametrize("kind", ["identity", "sigmoid", "own_softmax"])
def test_predict_3d_applies_inference_apply_nonlin(kind):
    """predict_3D applies what ``inference_apply_nonlin`` holds (reference neural_network.py:531-560; the constructor default
    is the identity, :80) -- identity and sigmoid through their own modes of e2e_nonlin_flip_acc, a caller's own softmax
    lambda through the fused softmax -- against the oracle's tiled prediction with the same function."""
    import torch.nn.functional as F
    net, _, params = tiny_net()
    spec = oracle.make_spec(TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], 2, TINY["max_feat"])
    fns = {"identity": lambda t: t, "sigmoid": torch.sigmoid, "own_softmax": lambda t: F.softmax(t, dim=1)}
    if kind != "identity":
        net.inference_apply_nonlin = fns[kind]          # identity: the attribute is left at the constructor's default
    net.eval()
    net.do_ds = False
    vol = seeded_input((TINY["cin"], 13, 50, 70), seed=72).numpy()
    seg, probs = net.predict_3D(vol, do_mirroring=True, mirror_axes=(0, 1, 2), use_sliding_window=True, step_size=0.5,
                                patch_size=TINY["patch"], use_gaussian=True, verbose=False)
    with torch.no_grad():
        rseg, rprobs = oracle.predict_tiled(lambda t: fns[kind](oracle.forward(spec, params, t, do_ds=False)), vol, TINY["k"],
                                            TINY["patch"], 0.5, True, (0, 1, 2), True)
    scale = max(1.0, float(np.abs(rprobs).max()))
    assert np.abs(probs - rprobs).max() <= (1e-4 if kind == "identity" else 2e-5) * scale
    assert (seg != rseg).mean() <= 1e-3
    if kind == "identity":
        assert probs.min() < 0 and np.abs(probs.sum(0) - 1).max() > 1e-2      # logits, not probabilities


def test_predict_3d_refuses_an_unknown_nonlin():
    net, _, _ = tiny_net()
    net.inference_apply_nonlin = torch.tanh
    net.eval()
    vol = seeded_input((TINY["cin"], 13, 50, 70), seed=72).numpy()
    with pytest.raises(NotImplementedError, match="inference_apply_nonlin"):
        net.predict_3D(vol, do_mirroring=False, use_sliding_window=True, patch_size=TINY["patch"], verbose=False)


@pytest.mark.parametrize("tag", ["raw", "clip"])
def test_gradient_growth_bit_exact_masks_vs_reference_golden(tag):
    """Masking(growth_mode='gradient').truncate_weights on the device (e2e_dsff_grad_score -> e2e_dsff_kth_value ->
    e2e_dsff_grow_above) against the masks the reference's truncate_weights produced from the same weights, masks and
    weight.grad (core_channel.py:556-611, :771-790; tests/golden/grad_growth.npz) -- 'raw': parameter.grad as it is;
    'clip': the trainer's route, unscaled gradients + the squared global norm the clip coefficient is derived from."""
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    from tests.test_oracle_golden import _grad_growth_fixture
    g, shapes, params, grads, names, before, after, num_death = _grad_growth_fixture(tag)
    net, _, _ = tiny_net()
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)

    class A:
        adv, fix, update_frequency, final_density = False, False, 1, 0.05
    random.seed(5)
    mask = Masking(opt, death_rate=0.3, death_mode='magnitude', death_rate_decay=CosineDecay(0.3, 10), growth_mode='gradient',
                   redistribution_mode='none', args=A())
    mask.add_module(net, sparse_init='uniform', density=0.4)
    assert mask.names == names
    for n in names:                      # the reference's initial masks (the same draws under the same seed)
        assert np.array_equal(mask._kmask_host[n], before[n]), n
    dev_grads = {n: t.cuda().contiguous() for n, t in grads.items()}
    if tag == "raw":
        for n, p_ in net.named_parameters():
            p_.grad = dev_grads[n]
    else:
        sq = torch.zeros(1, dtype=torch.float64, device="cuda")
        sq[0] = sum(float((t.double() ** 2).sum()) for t in grads.values())
        np.testing.assert_allclose(float(sq.sqrt()), g["clip_total_norm"][0], rtol=1e-6)
        mask.set_gradients(dev_grads, sq, float(g["clip_max_norm"][0]))
    mask.truncate_weights()
    for n in names:
        assert mask.num_death[n] == num_death[n]
        assert np.array_equal(mask._kmask_host[n], after[n]), n
        assert np.array_equal(mask.kmasks[n].cpu().numpy(), after[n]), n
        w = net.get_parameter(n)
        assert float((w * (1 - mask.masks[n])).abs().max()) == 0.0


def test_masking_refuses_unimplemented_modes():
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking
    for kw in (dict(growth_mode='momentum'), dict(death_mode='SET', growth_mode='random')):
        with pytest.raises(NotImplementedError):
            Masking(None, **kw)


def test_packed_weights_follow_the_parameters():
    """Round 6: the fp16 two-piece weights of the matrix-pipe convs and the operand-range words are cached per plan and rebuilt when
    the parameters change -- seen through torch's version counters (optimizer steps, load_state_dict, any in-place op on a
    parameter), a new storage (`p.data = ...`, what the reference's Masking.apply_mask does, core_channel.py:431) and the
    native-write epoch of this package's own kernels; in-place writes through `parameter.data` are invisible to all three and need
    `network.weights_changed()`.  The cached state must never serve a forward with other weights than the module holds."""
    from e2enet_medical_amd import engine as E
    patch = (16, 64, 64)                                  # L0 planes 64 x 64, base 32: the 64 -> 32 / 32 -> 32 layers run on conv133_mm_h2
    net = build_net(patch, 2, 32, 3, [(2, 2, 2)] * 3 + [(1, 2, 2)] * 2)
    shapes, params = load_closed_form(net)
    spec = oracle.make_spec(2, 32, 3, [(2, 2, 2)] * 3 + [(1, 2, 2)] * 2)
    x = seeded_input((1, 2) + patch, seed=5)
    net.eval()
    eng = net.engine(x.cuda())
    assert any(op.use_mm() for op in eng.conv_ops.values())

    last = {}

    def check(tag):
        # the bar is the noise class of this graph (InstanceNorms over 8 voxels at the deepest level: two fp32 evaluations differ by
        # 3e-4), the signal is two orders above it: every step below must move the reference logits by more than 20 bars
        with torch.no_grad():
            got = net(x.cuda())[0].cpu()
            ref = oracle.forward(spec, {n: p.detach().cpu() for n, p in net.named_parameters()}, x)[0]
        err = float((got - ref).abs().max())
        assert err <= 1e-3, (tag, err)
        if "ref" in last:
            moved = float((ref - last["ref"]).abs().max())
            assert moved >= 0.1 or tag == "fused optimizer step", (tag, "the step did not change the logits enough to prove anything", moved)
        last["ref"] = ref
    check("initial")
    name = "loc0.4.1.blocks.0.conv.weight"
    w = net.get_parameter(name)
    # (a uniform factor on a conv weight is cancelled by the InstanceNorm behind it: the changes below are not uniform)
    with torch.no_grad():
        w[::2].mul_(-1.0)                                # in-place on the parameter: version counter
    check("in-place op")
    w.data = w.data.flip(1)                              # new storage (the reference's apply_mask idiom)
    check("new storage")
    sd = {k: (torch.roll(v, 3, 0) if k == name else v.clone()) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    check("load_state_dict")
    key0 = eng._weights_key()
    w.data[1::2].mul_(-1.0)                              # through .data: nobody can see this one
    assert eng._weights_key() == key0
    net.weights_changed()
    check("weights_changed()")
    epoch = E.PARAM_EPOCH
    opt = torch.optim.SGD(net.parameters(), 1e-2, momentum=0.99, nesterov=True, weight_decay=3e-5)
    from e2enet_medical_amd.training.fused_optim import FusedClipSGD
    fused = FusedClipSGD(opt, list(net.named_parameters()), 12.0)
    net.train()
    outs = eng.forward(x.cuda(), True)
    tg = [seeded_labels((1, 1) + tuple(o.shape[2:]), 3, seed=60 + i).cuda() for i, o in enumerate(outs)]
    eng.loss_backward(tg, oracle.ds_weights(5), batch_dice=False)
    fused.step(eng.grads, None)                          # raw-pointer writes: the native-write epoch
    assert E.PARAM_EPOCH == epoch + 1
    net.eval()
    check("fused optimizer step")


def test_trainer_surface_runs_iterations():
    """nnUNetTrainer_simple surface: plans dict -> initialize -> run_iteration with Masking; loss finite, masks kept."""
    from e2enet_medical_amd.training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    plans = {'plans_per_stage': {0: {'batch_size': 2, 'patch_size': [16, 32, 32], 'num_pool_per_axis': [3, 5, 5],
                                     'pool_op_kernel_sizes': [[2, 2, 2]] * 3 + [[1, 2, 2]] * 2,
                                     'conv_kernel_sizes': [[3, 3, 3]] * 6, 'do_dummy_2D_data_aug': False}},
             'base_num_features': 32, 'num_modalities': 1, 'num_classes': 2, 'all_classes': [1, 2],
             'transpose_forward': [0, 1, 2], 'transpose_backward': [0, 1, 2], 'conv_per_stage': 2}
    tr = nnUNetTrainer_simple(plans, 0, output_folder=None, batch_dice=False, Tconv='shiftConvPP', max_num_epochs=2,
                              num_batches_per_epoch=2)
    tr.base_num_features_override = 8
    tr.synthetic_data = True
    torch.manual_seed(0)
    net, opt = tr.initialize(True)

    class A:
        adv = False
        fix = False
        update_frequency = 2
        final_density = 0.05
    random.seed(0)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 4), growth_mode='random',
                   redistribution_mode='none', args=A())
    mask.add_module(net, sparse_init='uniform', density=0.2)
    nnz0 = {n: int(m.sum().item()) for n, m in mask.masks.items()}
    losses = [float(tr.run_iteration(tr.tr_gen, True, mask=mask)) for _ in range(3)]
    assert all(np.isfinite(losses))
    assert {n: int(m.sum().item()) for n, m in mask.masks.items()} == nnz0        # prune/grow conserves the kernel count
    for n, m in mask.masks.items():                                               # dead kernels stay exactly zero
        assert float((net.get_parameter(n).detach() * (1 - m)).abs().max()) == 0.0
    v = float(tr.run_iteration(tr.val_gen, False))
    assert np.isfinite(v)


def test_checkpoint_with_dsff_state_resumes_bit_exact(tmp_path):
    """a checkpoint written with the Masking state (packed kernel maps, schedule position, growth RNG)
    resumes to the same weights and masks as the uninterrupted run (the reference re-draws the masks on resume)."""
    from e2enet_medical_amd.training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    plans = {'plans_per_stage': {0: {'batch_size': 2, 'patch_size': [16, 32, 32], 'num_pool_per_axis': [3, 5, 5],
                                     'pool_op_kernel_sizes': [[2, 2, 2]] * 3 + [[1, 2, 2]] * 2,
                                     'conv_kernel_sizes': [[3, 3, 3]] * 6, 'do_dummy_2D_data_aug': False}},
             'base_num_features': 32, 'num_modalities': 1, 'num_classes': 2, 'all_classes': [1, 2],
             'transpose_forward': [0, 1, 2], 'transpose_backward': [0, 1, 2], 'conv_per_stage': 2}

    class A:
        adv = False
        fix = False
        update_frequency = 2
        final_density = 0.05

    def make():
        tr = nnUNetTrainer_simple(plans, 0, output_folder=None, batch_dice=False, Tconv='shiftConvPP', max_num_epochs=2,
                                  num_batches_per_epoch=2)
        tr.base_num_features_override = 8
        torch.manual_seed(0)
        tr.synthetic_data = True
        net, opt = tr.initialize(True)
        random.seed(0)
        mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 8),
                       growth_mode='random', redistribution_mode='none', args=A())
        mask.add_module(net, sparse_init='uniform', density=0.2)
        return tr, net, mask

    tr, net, mask = make()
    batches = [next(tr.tr_gen) for _ in range(5)]
    for b in batches[:3]:                       # 3 iterations: one prune/grow at step 2
        tr.run_iteration(iter([b]), True, mask=mask)
    fname = str(tmp_path / "ckpt.model")
    tr.save_checkpoint(fname, mask=mask)
    for b in batches[3:]:                       # uninterrupted: 2 more (another prune/grow at step 4)
        tr.run_iteration(iter([b]), True, mask=mask)
    want_w = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    want_m = {n: m.cpu().clone() for n, m in mask.kmasks.items()}

    tr2, net2, mask2 = make()
    random.seed(12345)                          # a different RNG position: must be overwritten by the checkpoint
    tr2.load_checkpoint(fname, train=True, mask=mask2)
    assert mask2.steps == 3
    for b in batches[3:]:
        tr2.run_iteration(iter([b]), True, mask=mask2)
    for n in want_m:
        assert torch.equal(mask2.kmasks[n].cpu(), want_m[n]), n
    for k, v in net2.state_dict().items():
        assert torch.equal(v.detach().cpu(), want_w[k]), k


def test_overlapped_gradient_allreduce_single_rank_rccl():
    """Data-parallel path on one rank: the engine's bucket hook + asynchronous RCCL all-reduces over the flat gradient
    buffer must leave exactly the gradients of the plain backward pass (sum over one rank), and every parameter
    gradient must be a view into Engine.grad_flat."""
    import torch.distributed as dist
    from e2enet_medical_amd import parallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)      # "nccl" is RCCL on ROCm
    try:
        net, shapes, params = tiny_net()
        x = seeded_input((2, TINY["cin"]) + TINY["patch"], seed=79).cuda()
        eng = net.engine(x)
        outs = eng.forward(x, True)
        targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=95 + i).cuda() for i, o in enumerate(outs)]
        w = oracle.ds_weights(5)
        eng.loss_backward(targets, w, batch_dice=False)
        want = eng.grad_flat.clone()
        lo, hi = eng.grad_flat.data_ptr(), eng.grad_flat.data_ptr() + 4 * eng.grad_flat.numel()
        assert all(lo <= g.data_ptr() < hi for g in eng.grads.values())
        buckets = []
        ov = parallel.OverlappedGradAllReduce(eng, force=True)
        hook = eng.grad_bucket_hook
        eng.grad_bucket_hook = lambda a, b: (buckets.append((a, b)), hook(a, b))
        eng.forward(x, True)
        eng.loss_backward(targets, w, batch_dice=False)
        ov.finish()
        eng.grad_bucket_hook = None
        assert torch.equal(eng.grad_flat, want)
        assert buckets and buckets[0][0] == 0 and buckets[-1][1] == eng.grad_flat.numel()
        assert all(b[1] == c[0] for b, c in zip(buckets, buckets[1:]))          # contiguous, in completion order
    finally:
        if created:
            dist.destroy_process_group()


def test_btcv_like_anisotropic_config_vs_oracle():
    """BASELINE config 3 shape family: 1 modality, 14 classes, anisotropic pooling [[1,2,2],[2,2,2]x3,[1,2,2]]
    (depth stride 1 with in-plane stride 2, transposed convs with kernel (1,2,2)), batch 2, non-cubic patch."""
    pools = [(1, 2, 2), (2, 2, 2), (2, 2, 2), (2, 2, 2), (1, 2, 2)]
    patch, cin, base, k = (8, 64, 96), 1, 8, 14
    net = build_net(patch, cin, base, k, pools, 40)
    shapes, params = load_closed_form(net)
    spec = oracle.make_spec(cin, base, k, pools, 2, 40)
    assert list(onet.param_shapes(spec).keys()) == list(shapes.keys())
    x = seeded_input((2, cin) + patch, seed=5)
    eng = net.engine(x.cuda())
    outs = eng.forward(x.cuda(), True)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), k, seed=90 + i) for i, o in enumerate(outs)]
    w = oracle.ds_weights(5)
    loss = eng.loss_backward([t.cuda() for t in targets], w, batch_dice=False)
    leaves = {n: p.clone().requires_grad_(True) for n, p in params.items()}
    ref = oracle.forward(spec, leaves, x)
    ref_loss = oracle.deep_supervision_loss(ref, targets, w, False)
    ref_loss.backward()
    for o, r in zip(outs, ref):
        assert o.shape == r.shape and (o.cpu() - r.detach()).abs().max() <= 1e-4
    assert abs(loss.item() - ref_loss.item()) < 5e-5
    for n in shapes:
        rg = leaves[n].grad
        err = (eng.grads[n].cpu() - rg).abs().max().item()
        assert err <= 2e-4 * max(1.0, rg.abs().max().item()) + 1e-6, (n, err)


def test_width48_density_quirk_masks_bit_exact():
    """Reference quirk: tensors with shape[0] == 48 get density 0.2 whatever --density says (core_channel.py:147-151)."""
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    g = golden("masks.npz")
    net = build_net((64, 64, 64), 4, 48, 4, [(2, 2, 2)] * 5)
    opt = torch.optim.SGD(net.parameters(), 1e-2, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        update_frequency = 1200
        final_density = 0.05
    for dens in (0.1, 0.5):
        random.seed(0)
        mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 10),
                       growth_mode='random', redistribution_mode='none', args=A())
        mask.add_module(net, sparse_init='uniform', density=dens)
        tag = "b48_d%s" % dens
        assert list(mask.masks.keys()) == [str(s) for s in g[tag + "_names"]]
        assert [sha_of(pack_kernel_mask(m.cpu())) for m in mask.masks.values()] == [str(s) for s in g[tag + "_sha"]]
        assert [int(m.sum().item()) for m in mask.masks.values()] == list(g[tag + "_nnz"])


def test_full_size_128_properties():
    """BASELINE size (4 x 128^3, base 32, density 0.2): size-independent properties instead of an oracle run --
    (a) skipping dead kernels via the liveness bits == dense execution of the same masked weights,
    (b) liveness derived from the zero kernels of the weights == liveness from the masks (bit identical),
    (c) run-to-run determinism, (d) every InstanceNorm'ed tensor really has zero mean / unit variance,
    (e) flip equivariance does NOT hold for the depth-shifted net (sanity: the shift is applied)."""
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    torch.manual_seed(0)
    net = build_net((128, 128, 128), 4, 32, 4, [(2, 2, 2)] * 5)
    opt = torch.optim.SGD(net.parameters(), 1e-2, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        update_frequency = 1200
        final_density = 0.05
    random.seed(0)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 10), growth_mode='random',
                   redistribution_mode='none', args=A())
    mask.add_module(net, sparse_init='uniform', density=0.2)
    total = sum(m.numel() for m in mask.masks.values())
    nnz = sum(int(m.sum().item()) for m in mask.masks.values())
    assert total == 12061696 and abs(nnz / total - 0.2) < 1e-4                # 12.06 M masked params at 32 ch
    net.eval()
    x = seeded_input((1, 4, 128, 128, 128), seed=3).cuda()
    with torch.no_grad():
        a = [o.clone() for o in net(x)]
        b = [o.clone() for o in net(x)]
        assert all(torch.equal(p, q) for p, q in zip(a, b))                     # (c)
        net.enable_auto_sparsity(True)
        c = net(x)
        assert all(torch.equal(p, q) for p, q in zip(a, c))                     # (b)
        net.set_kernel_masks(None)
        dense = net(x)
        assert max((p - q).abs().max().item() for p, q in zip(a, dense)) <= 2e-5   # (a): only the summation order differs
        eng = net.engine(x)
        for name in ("conv_blocks_context.0.blocks.1", "loc0.4.1.blocks.0", "loc2.0.0.blocks.0"):
            op = [o for o in eng.conv_ops.values() if o.prefix == name][0]
            y = op.out.data[0]
            mu = y.double().mean(dim=(1, 2, 3))
            var = y.double().var(dim=(1, 2, 3), unbiased=False)
            assert (op.out.mean[:y.shape[0]].double() - mu).abs().max().item() < 1e-5 * max(1.0, mu.abs().max().item())
            rstd = 1.0 / torch.sqrt(var + 1e-5)
            assert ((op.out.rstd[:y.shape[0]].double() - rstd) / rstd).abs().max().item() < 1e-5   # (d)
        flipped = net(torch.flip(x, (2,)))
        assert (torch.flip(flipped[0], (2,)) - dense[0]).abs().max().item() > 1e-3                   # (e)
    assert all(torch.isfinite(o).all() for o in a)


def test_training_trajectory_nine_iterations_vs_oracle():
    """Nine full training iterations (forward, DS loss, backward, clip 12, SGD-Nesterov, mask step) with prune/grow events
    at iterations 3, 6 and 9, engine against the CPU oracle run side by side from the same start: per-iteration losses,
    clip norms and death rates, mask indices after every update (bit exact), weights at the end.  (The two-iteration
    fixture from the reference pins the oracle; this pins the engine's drift over a longer run with three updates.)"""
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    from e2enet_medical_amd.training.fused_optim import FusedClipSGD
    net, shapes, params0 = tiny_net(SPARSE_PATCH)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        update_frequency = 3
        final_density = 0.05
    spec = oracle.make_spec(TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], 2, TINY["max_feat"])
    names = oracle.masked_names(spec)
    # engine side and oracle side consume the same Python `random` stream: seed before each side's draws
    random.seed(11)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 12), growth_mode='random',
                   redistribution_mode='none', args=A())
    mask.add_module(net, sparse_init='uniform', density=0.4)
    rs_engine = random.getstate()
    random.seed(11)
    oparams = {n: p.clone() for n, p in params0.items()}
    mom = {}
    ostate = oracle.DsffState(oparams, names, 0.4, 0.5, 12, 3, momentum_buffers=mom)
    rs_oracle = random.getstate()
    for n in names:
        assert torch.equal(mask.masks[n].cpu(), ostate.masks[n]), n

    x = seeded_input((2, TINY["cin"]) + SPARSE_PATCH, seed=21)
    xg = x.cuda()
    w = oracle.ds_weights(5)
    fused = FusedClipSGD(opt, list(net.named_parameters()), 12.0)
    eng = net.engine(xg)
    targets = None
    for it in range(9):
        outs = eng.forward(xg, True)
        if targets is None:
            targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=30 + i) for i, o in enumerate(outs)]
            tg = [t.cuda() for t in targets]
        loss = eng.loss_backward(tg, w, batch_dice=False).item()
        fused.step(eng.grads, mask.masks)
        tn = fused.total_norm()
        will_update = (it + 1) % 3 == 0
        if will_update:
            pre_w = {n: net.get_parameter(n).detach().cpu().clone() for n in names}
            pre_m = {n: mask.masks[n].cpu().clone() for n in names}
        random.setstate(rs_engine)
        updated = mask.step(masks_already_applied=True)
        rs_after = random.getstate()
        if will_update:
            # the engine's prune/grow decision replayed by the oracle's algorithm on the engine's OWN pre-update weights
            # and the same random stream: bit-identical mask indices at every update
            random.setstate(rs_engine)
            rep, nd = {}, {}
            for n in names:
                rep[n], nd[n] = oracle.kernel_death(pre_m[n], pre_w[n] * pre_m[n], mask.death_rate)
            for n in names:
                rep[n] = oracle.kernel_growth(rep[n], nd[n])
            for n in names:
                assert torch.equal(mask.masks[n].cpu(), rep[n]), "replayed update at iteration %d: %s" % (it, n)
        rs_engine = rs_after

        leaves = {n: p.detach().clone().requires_grad_(True) for n, p in oparams.items()}
        ref_loss = oracle.deep_supervision_loss(oracle.forward(spec, leaves, x), targets, w, False)
        ref_loss.backward()
        ref_tn = oracle.clip_and_sgd_step(oparams, {n: leaves[n].grad for n in leaves}, mom, 1e-2).item()
        ostate.params = oparams
        random.setstate(rs_oracle)
        ref_updated = ostate.step()
        rs_oracle = random.getstate()

        assert abs(loss - ref_loss.item()) <= 2e-4 * max(1.0, abs(ref_loss.item())), "loss at iteration %d: %g vs %g" % (it, loss, ref_loss.item())
        assert abs(tn - ref_tn) <= 2e-3 * ref_tn, "clip norm at iteration %d" % it
        assert mask.death_rate == ostate.death_rate
        assert bool(updated) == bool(ref_updated) == ((it + 1) % 3 == 0)
        if updated:
            # Engine run against oracle run.  Kernels regrown at the previous update carry tiny, similar weights: at later
            # updates the magnitude threshold can fall between two of them that differ by less than fp32 noise, and the two
            # runs then kill a different one.  The first update has no such ties and must agree bit for bit; after a later
            # one the runs may be legitimately different networks and the side-by-side comparison ends (the replay above
            # still pins every update of the engine).
            diff = {n: int((mask.masks[n].cpu()[:, :, 0, 0, 0] != ostate.masks[n][:, :, 0, 0, 0]).sum().item()) for n in names}
            total = sum(diff.values())
            if it < 3:
                assert total == 0, "mask indices after the first update: %s" % {n: d for n, d in diff.items() if d}
            else:
                # (one swapped pair in the death set shifts the candidate list the growth draws index into, so a tensor
                #  that differs at all differs in tens of entries; most tensors must still agree)
                assert sum(1 for d in diff.values() if d) <= len(names) // 4, "update at iteration %d: %s" % (it, diff)
                for n in names:          # same number of live kernels either way
                    assert int(mask.masks[n].sum().item()) == int(ostate.masks[n].sum().item()), n
                if total:
                    compared_until = it
                    break
    else:
        compared_until = 8
    assert compared_until >= 5, "the runs must agree at least through the second update"
    if compared_until < 8:
        return
    for n, p in net.named_parameters():
        ref = oparams[n]
        assert (p.detach().cpu() - ref).abs().max().item() <= 5e-4 * max(1e-2, ref.abs().max().item()), n


def test_full_size_128_training_properties():
    """BASELINE training workload (2 x 4 x 128^3, base 32, K 4, density 0.2), properties that need no oracle run:
    (a) backward is exactly linear in the logit gradients under scaling by a power of two (bit-identical x 0.5), and
        run-to-run deterministic;
    (b) the data gradient with dead kernels skipped == dense execution of the same masked weights;
    (c) ten iterations on one batch with two prune/grow events: finite, decreasing loss; live-kernel counts preserved;
        weights and momentum outside the masks exactly zero after every step."""
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    from e2enet_medical_amd.training.fused_optim import FusedClipSGD
    torch.manual_seed(0)
    net = build_net((128, 128, 128), 4, 32, 4, [(2, 2, 2)] * 5)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        update_frequency = 4
        final_density = 0.05
    random.seed(0)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 10), growth_mode='random',
                   redistribution_mode='none', args=A())
    mask.add_module(net, sparse_init='uniform', density=0.2)
    names = list(mask.masks.keys())
    x = seeded_input((2, 4, 128, 128, 128), seed=5).cuda()
    eng = net.engine(x)
    outs = eng.forward(x, True)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), 4, seed=40 + i).cuda() for i, o in enumerate(outs)]
    w = oracle.ds_weights(5)

    # (a) explicit logit gradients g, then g / 2
    eng.prepare_backward()
    gl = [torch.from_numpy(np.random.RandomState(7 + i).standard_normal(tuple(o.shape)).astype(np.float32)).cuda() * 1e-3 for i, o in enumerate(outs)]
    eng.backward(gl)
    g1 = {n: v.clone() for n, v in eng.grads.items()}
    eng.forward(x, True)
    eng.backward(gl)
    for n in g1:
        assert torch.equal(g1[n], eng.grads[n]), "determinism: " + n
    eng.forward(x, True)
    eng.backward([g * 0.5 for g in gl])
    for n in g1:
        assert torch.equal(g1[n] * 0.5, eng.grads[n]), "linearity: " + n

    # (b) liveness-skipping data gradient against dense execution of the same (masked) weights by the SAME kernel, conv133_kernel
    # (the matrix-core conv path and the load-balanced plans are switched off for this comparison: without kernel maps every
    # served layer would go to conv133_dense_kernel, and a plan walks the planes in another order)
    from e2enet_medical_amd import engine as engine_mod

    def noise_class(a, b, what):
        num = sum((a[n].double() - b[n].double()).pow(2).sum().item() for n in b)
        den = sum(v.double().pow(2).sum().item() for v in b.values())
        typical = (den / len(b)) ** 0.5           # (tensors whose gradient is analytically zero are measured against the typical tensor)
        worst = max(((a[n] - v).norm() / max(v.norm().item(), 1e-2 * typical)).item() for n, v in b.items())
        print("[%s] global rel-L2 %.3e, worst tensor %.3e" % (what, (num / den) ** 0.5, worst))
        assert (num / den) ** 0.5 <= 5e-3 and worst <= 5e-2, what
    engine_mod.DENSE_ENABLED = False
    try:
        eng.forward(x, True)
        eng.backward(gl)
        assert any(op.sp_fwd is not None for op in eng.conv_ops.values()), "the full-resolution masked layers run on plans"
        planned = {n: v.clone() for n, v in eng.grads.items()}
        engine_mod.SPARSE2 = False
        mask._push_liveness()                       # (the plan picks the maps up at its next forward)
        eng.forward(x, True)
        assert all(op.sp_fwd is None for op in eng.conv_ops.values())
        eng.backward(gl)
        sparse = {n: v.clone() for n, v in eng.grads.items()}
        net.set_kernel_masks(None)
        eng.forward(x, True)
        eng.backward(gl)
        for n, v in eng.grads.items():
            ref = v
            assert (sparse[n] - ref).abs().max().item() <= 2e-4 * max(ref.abs().max().item(), 1e-6), "sparse vs dense backward: " + n
        # (b1) the load-balanced plans (another plane order per chunk, another assignment of planes to waves) against the plain
        # walk: equality in the fp32 noise class of this graph (InstanceNorms over 64 voxels at the bottleneck)
        noise_class(planned, sparse, "planned kernel vs plain sparse walk backward")
    finally:
        engine_mod.DENSE_ENABLED = True
        engine_mod.SPARSE2 = True
    mask._push_liveness()
    # (b2) the same masked weights through the matrix-core conv kernel (bf16 x 3 operands) wherever it is served: another
    # summation order, so equality only in the fp32 noise class of this graph (InstanceNorms over 64 voxels at the bottleneck)
    net.set_kernel_masks(None)
    eng.forward(x, True)
    eng.backward(gl)
    noise_class(sparse, {n: v.clone() for n, v in eng.grads.items()}, "dense-kernel vs sparse-walk backward")
    mask._push_liveness()

    # (c) ten iterations
    fused = FusedClipSGD(opt, list(net.named_parameters()), 12.0)
    nnz0 = {n: int(mask.masks[n].sum().item()) for n in names}
    losses = []
    for it in range(10):
        eng.forward(x, True)
        loss = eng.loss_backward(targets, w, batch_dice=False)
        fused.step(eng.grads, mask.masks)
        updated = mask.step(masks_already_applied=True)
        assert bool(updated) == ((it + 1) % 4 == 0)
        losses.append(loss.item())
        for n in names[::5]:
            m = mask.masks[n]
            assert int(m.sum().item()) == nnz0[n], n
            assert float((net.get_parameter(n).detach() * (1 - m)).abs().max()) == 0.0, n
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0] - 0.02, losses


def test_whole_net_128_vs_oracle():
    """BASELINE config 2 at the benchmarked size against the CPU oracle (reference: unetpp_d.py:447-488 at 128^3): B = 1,
    the benchmark's network (He init under torch.manual_seed(0), DSFF masks at density 0.2 under random.seed(0)), its input
    and targets: all four logit heads within 1e-4 of the fp32 oracle, loss within 5e-5, Dice of the argmax maps >= 1 - 1e-3
    (metrics.py:106-121); then EVERY parameter gradient against the fp64 oracle under the engine's own branch decisions.  One oracle
    forward + loss at this size takes ~10-20 s on the host, the two forced-branch passes ~2 min."""
    import bench
    net, opt, mask, fused = bench.build(torch.device("cuda"))
    x, targets = bench.synthetic_batch(torch.device("cuda"), bench.PATCH, 1, seed=100)
    eng = net.engine(x)
    outs = eng.forward(x, True)
    w = oracle.ds_weights(5)
    loss = eng.loss_backward(targets, w, batch_dice=False)
    spec = oracle.make_spec(bench.CIN, bench.BASE, bench.K, bench.POOLS)
    params = {n: p.detach().cpu().clone() for n, p in net.named_parameters()}
    for n, m in mask.masks.items():                 # the weights the engine ran with are the masked ones
        assert float((params[n] * (1 - m.cpu())).abs().max()) == 0.0
    torch.set_num_threads(max(1, min(torch.get_num_threads(), 16)))
    with torch.no_grad():
        ref = oracle.forward(spec, params, x.cpu())
        ref_loss = oracle.deep_supervision_loss(ref, [t.cpu() for t in targets], w, False)
    for i, (o, r) in enumerate(zip(outs, ref)):
        err = (o.cpu() - r).abs().max().item()
        assert o.shape == r.shape and err <= 1e-4, "head %d: max|dlogit| %.3e" % (i, err)
    assert abs(loss.item() - ref_loss.item()) <= 5e-5
    seg, rseg = outs[0].argmax(1).cpu().numpy(), ref[0].argmax(1).numpy()
    for label in range(1, bench.K):
        assert oracle.hard_dice(seg, rseg, label) >= 1 - 1e-3
    # gradients at the benchmarked size: every parameter gradient against the fp64 oracle evaluated with the engine's own LeakyReLU /
    # pooling decisions (tests/helpers.py: rounding only, no kink noise) -- two more oracle passes of 128^3 on the host (~2 min)
    from tests.helpers import check_grads_same_branches
    shapes = {n: tuple(p.shape) for n, p in net.named_parameters()}
    check_grads_same_branches(eng, spec, params, x.cpu(), [t.cpu() for t in targets], w, shapes)


@pytest.mark.parametrize("mode", ["explicit", "loss"])
def test_engine_forward_and_backward_are_graph_capturable(mode):
    """include/e2e_hip.h promises asynchronous, allocation-free, graph-capturable entry points: capture a whole forward
    and backward pass of the tiny net into a HIP graph through torch.cuda.CUDAGraph and replay it on new input.
    "explicit": backward from given logit gradients -- logits and every parameter gradient bit-identical to the eager
    launches.  "loss": forward + fused loss kernels + backward -- logits bit-identical, loss and gradients at 1e-6 (the
    Dice sums are fp64 atomics whose order is not fixed, eager or not).  This test is why the library zeroes its small
    accumulators with a kernel (e2e::zero_async) and not with hipMemsetAsync: with memset nodes in the graph the second
    replay returned NaN gradients whenever eager work ran between two replays."""
    net, shapes, _ = tiny_net()
    xa = seeded_input((2, TINY["cin"]) + TINY["patch"], seed=61).cuda()
    xb = seeded_input((2, TINY["cin"]) + TINY["patch"], seed=62).cuda()
    eng = net.engine(xa)
    outs = eng.forward(xa, True)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=70 + i).cuda() for i, o in enumerate(outs)]
    gl = [torch.from_numpy(np.random.RandomState(9 + i).standard_normal(tuple(o.shape)).astype(np.float32)).cuda() * 1e-2
          for i, o in enumerate(outs)]
    w = oracle.ds_weights(5)

    def body(x):
        o = eng.forward(x, True)
        if mode == "loss":
            return o, eng.loss_backward(targets, w, batch_dice=False)
        eng.backward(gl)
        return o, None
    eng.loss_backward(targets, w, batch_dice=False)            # eager warm-up: every buffer exists before the capture
    eager = {}
    for tag, x in (("a", xa), ("b", xb)):
        o, loss = body(x)
        eager[tag] = ([t.clone() for t in o], None if loss is None else loss.clone(), {n: g.clone() for n, g in eng.grads.items()})
    xin = xa.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body(xin)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gouts, gloss = body(xin)
    for tag, x in (("a", xa), ("b", xb), ("a", xa)):
        o_ref, l_ref, g_ref = eager[tag]
        xin.copy_(x)
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(gouts, o_ref):
            assert torch.equal(a, b)
        if mode == "loss":
            assert abs(gloss.item() - l_ref.item()) <= 1e-6
        for n in g_ref:
            if mode == "explicit":
                assert torch.equal(eng.grads[n], g_ref[n]), n
            else:
                assert (eng.grads[n] - g_ref[n]).abs().max().item() <= 1e-6 * max(1.0, g_ref[n].abs().max().item()), (tag, n)


def test_small_plan_graph_replay_matches_eager(monkeypatch):
    """Engine._run: a small plan replays its forward and its loss + backward op lists as HIP graphs from the third call on.
    Replays must be bit-identical to eager execution, follow weight updates (stable parameter pointers), switch correctly
    between deep-supervision modes, and be dropped when new liveness tables arrive (prune / grow)."""
    net, shapes, params = tiny_net()
    w = oracle.ds_weights(5)

    def run(mode, steps):
        monkeypatch.setenv("E2E_GRAPHS", mode)
        net._engines.clear()
        with torch.no_grad():
            for n, p in net.named_parameters():
                p.copy_(params[n])
        net.set_kernel_masks(None)
        res = []
        for it in range(steps):
            x = seeded_input((2, TINY["cin"]) + TINY["patch"], seed=500 + it).cuda()
            eng = net.engine(x)
            outs = [o.clone() for o in eng.forward(x, True)]
            targets = [seeded_labels((2, 1) + tuple(o.shape[2:]), TINY["k"], seed=600 + it + i).cuda() for i, o in enumerate(outs)]
            loss = eng.loss_backward(targets, w, batch_dice=False).clone()
            grads = {n: g.clone() for n, g in eng.grads.items()}
            with torch.no_grad():                         # a weight update between iterations (plain SGD: pointers stay)
                for n, p in net.named_parameters():
                    p.add_(grads[n], alpha=-0.05)
            if it == 3:                                    # new liveness tables: graphs must be rebuilt
                names = oracle.masked_names(oracle.make_spec(TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], 2, TINY["max_feat"]))
                random.seed(9)
                masks = oracle.uniform_kernel_masks(shapes, names, 0.5)
                with torch.no_grad():
                    for n in names:
                        net.get_parameter(n).mul_(masks[n].cuda())
                net.set_kernel_masks({n: (masks[n].reshape(masks[n].shape[0], masks[n].shape[1], -1).sum(-1) > 0).to(torch.uint8) for n in names})
            full = eng.forward(x, False).clone() if it % 2 == 0 else None        # inference pass without deep supervision in between
            res.append((outs, loss, grads, full))
        return res, net.engine(x)
    eager, _ = run("0", 8)
    graphed, eng = run("1", 8)
    assert len(eng._graphs) >= 2, "the graphs were not built"
    for (o1, l1, g1, f1), (o2, l2, g2, f2) in zip(eager, graphed):
        assert all(torch.equal(a, b) for a, b in zip(o1, o2)) and torch.equal(l1, l2)
        assert all(torch.equal(g1[n], g2[n]) for n in g1)
        assert (f1 is None) == (f2 is None) and (f1 is None or torch.equal(f1, f2))


@pytest.mark.gpu
def test_two_lane_issue_matches_single_stream(monkeypatch):
    """Engine._exec: the deep levels on the 'light' HIP stream and the weight gradients on a third one must give bit-identical
    logits, loss and gradients to the single-stream pass (same op order, same accumulation order into shared gradient
    buffers), over several iterations (stream joins between passes) and with the data-parallel bucket hook installed."""
    from e2enet_medical_amd import engine as E
    monkeypatch.setenv("E2E_GRAPHS", "0")
    patch = (32, 64, 64)
    net, shapes, params = tiny_net(patch)
    w = oracle.ds_weights(5)

    def run(lanes, wg, hook):
        monkeypatch.setattr(E, "LANES", lanes)
        monkeypatch.setattr(E, "WGRAD_STREAM", wg)
        monkeypatch.setattr(E, "LANE_DIVS", (8, 512))         # three lanes: level 0 | levels 1-2 | levels >= 3 of this small patch
        monkeypatch.setattr(E, "WGRAD_STREAM_MAX_ELEMS", 1 << 22)
        net._engines.clear()
        res = []
        for it in range(3):
            x = seeded_input((2, TINY["cin"]) + patch, seed=700 + it).cuda()
            eng = net.engine(x)
            seen = []
            eng.grad_bucket_hook = (lambda lo, hi: seen.append((lo, hi, eng.grad_flat[lo:hi].clone()))) if hook else None
            outs = [o.clone() for o in eng.forward(x, True)]
            targets = [seeded_labels((2, 1) + tuple(o.shape[2:]), TINY["k"], seed=800 + it + i).cuda() for i, o in enumerate(outs)]
            loss = eng.loss_backward(targets, w, batch_dice=False).clone()
            grads = {n: g.clone() for n, g in eng.grads.items()}
            for lo, hi, snap in seen:                             # a bucket handed to the hook is final
                assert torch.equal(snap, eng.grad_flat[lo:hi]), "bucket [%d,%d) changed after its hook" % (lo, hi)
            res.append((outs, loss, grads))
        if lanes:
            assert set(eng._lane_of) == {0, 1, 2} and eng._lane_streams is not None
            assert any(eng._deps_fwd) and any(eng._deps_bwd)
        return res
    base = run(False, False, False)
    for cfg in ((True, True, False), (True, False, False), (False, True, False), (True, True, True)):
        other = run(*cfg)
        for (o1, l1, g1), (o2, l2, g2) in zip(base, other):
            assert all(torch.equal(a, b) for a, b in zip(o1, o2)) and torch.equal(l1, l2), cfg
            bad = [n for n in g1 if not torch.equal(g1[n], g2[n])]
            assert not bad, (cfg, bad[:4])


def test_fused_instancenorm_backward_sums_match_the_two_pass_form():
    """Round 4: the op that writes a gradient buffer LAST also forms the two per-instance sums of the InstanceNorm + LeakyReLU
    backward of that buffer's producer (autograd of unetpp_d.py:99-100, :111) as per-block / per-tile records, and the producer's
    e2e_in_lrelu_bwd adds the records up (fixed order) and runs its apply pass only.  Level 1 (default): the pooling backward,
    issued behind the other consumers' data gradients for this purpose; level 2: also conv133_sparse_kernel<1> (depth-shifted
    channels at the volume border -- slices that launch does not touch -- included).  Same arithmetic, another summation order
    of the fp32 partials below the fp64 sums (and, with three writers, of the fp32 accumulation into the buffer): every
    parameter gradient agrees with the two-pass form to 5e-5 of its scale."""
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    from e2enet_medical_amd import engine as engine_mod
    torch.manual_seed(3)
    net = build_net((16, 64, 64), 2, 16, 3, [(2, 2, 2)] * 3 + [(1, 2, 2)] * 2, 64)
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)

    class A:
        adv = False
        fix = False
        update_frequency = 1000
        final_density = 0.05
    random.seed(4)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 10), growth_mode='random',
                   redistribution_mode='none', args=A())
    mask.add_module(net, sparse_init='uniform', density=0.25)
    x = seeded_input((2, 2, 16, 64, 64), seed=9).cuda()
    w = oracle.ds_weights(5)
    grads, targets = {}, None
    try:
        for level in (1, 2, 0):
            engine_mod.FUSE_IN_SUMS = level
            net._engines.clear()                                # the issue order and the record buffers belong to the plan
            eng = net.engine(x)
            outs = eng.forward(x, True)
            if targets is None:
                targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), 3, seed=60 + i).cuda() for i, o in enumerate(outs)]
            eng.loss_backward(targets, w, batch_dice=False)
            by_pool = sum(1 for op in eng.conv_ops.values() if op.own_sums is not None and isinstance(op.out.last_writer, engine_mod.PoolOp))
            by_conv = sum(len(op.sp_bwd.fused_srcs) for op in eng.conv_ops.values() if op.sp_bwd is not None and op.sp_bwd.table is not None)
            print("[fused sums level %d] %d buffers by the pooling backward, %d by the planned data gradient" % (level, by_pool, by_conv))
            assert (by_pool > 0) == (level >= 1) and (by_conv > 0) == (level >= 2)
            assert all(not op.out.sums_ready for op in eng.conv_ops.values())                    # every producer consumed its sums
            grads[level] = {n: v.clone() for n, v in eng.grads.items()}
    finally:
        engine_mod.FUSE_IN_SUMS = 1
        net._engines.clear()
    for level in (1, 2):
        worst = (0.0, None)
        for n, v in grads[0].items():
            if n.endswith(".conv.bias"):            # analytically zero (a bias in front of an InstanceNorm): both sides hold rounding noise
                continue
            scale = max(v.abs().max().item(), 1e-6)
            err = (grads[level][n] - v).abs().max().item() / scale
            if err > worst[0]:
                worst = (err, n)
        print("[fused InstanceNorm-backward sums, level %d, vs two passes] worst tensor %.3e of its scale (%s)" % ((level,) + worst))
        assert worst[0] <= 5e-5, (level, worst)
