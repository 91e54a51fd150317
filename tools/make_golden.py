#!/usr/bin/env python
"""Generate tests/golden/*.npz by importing the REFERENCE from /root/reference.

Runs only in the build container (the reference never travels to the GPU box).
Nothing from the reference is copied: this script imports it, feeds seeded /
closed-form inputs and stores inputs' recipes + outputs as data fixtures.

    python tools/make_golden.py            # all fixtures
    python tools/make_golden.py shift net  # a subset
"""
import os
import sys
import types
import random

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

from tests.helpers import (closed_form_params, seeded_input, seeded_labels,   # noqa: E402
                           pack_kernel_mask, sha_of)


# ----------------------------------------------------------------------------- stubs
def _pad_nd_image(image, new_shape=None, mode="constant", kwargs=None, return_slicer=False,
                  shape_must_be_divisible_by=None):
    """Restatement of batchgenerators==0.24 pad_nd_image (third party, absent here)."""
    if kwargs is None:
        kwargs = {'constant_values': 0}
    old_shape = np.array(image.shape[-len(new_shape):])
    num_axes_nopad = len(image.shape) - len(new_shape)
    new_shape = np.array([max(new_shape[i], old_shape[i]) for i in range(len(new_shape))])
    if shape_must_be_divisible_by is not None:
        div = np.array(shape_must_be_divisible_by) if not isinstance(shape_must_be_divisible_by, int) \
            else np.array([shape_must_be_divisible_by] * len(new_shape))
        for i in range(len(new_shape)):
            if new_shape[i] % div[i] != 0:
                new_shape[i] += div[i] - new_shape[i] % div[i]
    difference = new_shape - old_shape
    pad_below = difference // 2
    pad_above = difference // 2 + difference % 2
    pad_list = [[0, 0]] * num_axes_nopad + list([list(i) for i in zip(pad_below, pad_above)])
    if not (all(i == 0 for i in pad_below) and all(i == 0 for i in pad_above)):
        res = np.pad(image, pad_list, mode, **kwargs)
    else:
        res = image
    if not return_slicer:
        return res
    pad_list = np.array(pad_list)
    pad_list[:, 1] = np.array(res.shape) - pad_list[:, 1]
    return res, list(slice(*i) for i in pad_list)


def install_stubs():
    ut = types.ModuleType('batchgenerators.augmentations.utils')
    ut.pad_nd_image = _pad_nd_image
    for n, m in [('batchgenerators', types.ModuleType('batchgenerators')),
                 ('batchgenerators.augmentations', types.ModuleType('batchgenerators.augmentations')),
                 ('batchgenerators.augmentations.utils', ut),
                 ('medpy', types.ModuleType('medpy')),
                 ('medpy.metric', types.ModuleType('medpy.metric'))]:
        sys.modules[n] = m
    sys.modules['medpy'].metric = sys.modules['medpy.metric']
    torch.Tensor.cuda = lambda self, *a, **k: self      # Masking calls .cuda() unconditionally


install_stubs()
from torch import nn                                                            # noqa: E402
from e2enet.network_architecture.unetpp_d import (Generic_UNetPlusPlus, InitWeights_He,   # noqa: E402
                                                  torch_shift, ConvDropoutNormNonlin)
from e2enet.network_architecture.neural_network import SegmentationNetwork     # noqa: E402
from e2enet.training.network_training.sparselearning.core_channel import (     # noqa: E402
    Masking, CosineDecay)
from e2enet.training.loss_functions.dice_loss import DC_and_CE_loss            # noqa: E402
from e2enet.training.loss_functions.deep_supervision import MultipleOutputLoss2  # noqa: E402
from e2enet.evaluation.metrics import dice as ref_dice                         # noqa: E402


def build_ref_net(patch, cin, base, k, pools, max_feat=None, seed=None):
    if seed is not None:
        torch.manual_seed(seed)
    return Generic_UNetPlusPlus(patch, cin, base, k, len(pools), 2, 2, nn.Conv3d, nn.InstanceNorm3d,
                                {'eps': 1e-5, 'affine': True}, nn.Dropout3d, {'p': 0, 'inplace': True},
                                nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                                lambda x: x, InitWeights_He(1e-2), pools, None, False, True, True,
                                max_num_features=max_feat)


def load_closed_form(net):
    shapes = {n: tuple(p.shape) for n, p in net.named_parameters()}
    params = closed_form_params(shapes)
    with torch.no_grad():
        for n, p in net.named_parameters():
            p.copy_(params[n])
    return shapes


class _Args:
    adv = False
    fix = False
    update_frequency = 1
    final_density = 0.05


def make_masking(net, density, death_rate=0.5, t_max=10, update_frequency=1, seed=0):
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    args = _Args()
    args.update_frequency = update_frequency
    decay = CosineDecay(death_rate, t_max)
    random.seed(seed)
    mask = Masking(opt, death_rate=death_rate, death_mode='magnitude', death_rate_decay=decay,
                   growth_mode='random', redistribution_mode='none', args=args)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        mask.add_module(net, sparse_init='uniform', density=density)
    return mask, opt


# ----------------------------------------------------------------------------- fixtures
def gen_shift():
    out = {}
    for c in (1, 2, 3, 4, 5, 7, 12, 32, 64, 160, 896):
        d, h, w = 7, 2, 3
        x = seeded_input((2, c, d, h, w), seed=100 + c)
        y = torch_shift(5, 2, 3)(x)
        out["c%d" % c] = y.numpy()
    # a one-slice-deep volume (every shifted group reads zeros)
    x = seeded_input((1, 10, 1, 2, 2), seed=7)
    out["d1_c10"] = torch_shift(5, 2, 3)(x).numpy()
    x = seeded_input((1, 10, 2, 2, 2), seed=8)
    out["d2_c10"] = torch_shift(5, 2, 3)(x).numpy()
    np.savez_compressed(os.path.join(OUT, "shift.npz"), **out)


def gen_block():
    out = {}
    for tag, stride, cin, cout, shape in (("s1", (1, 1, 1), 8, 6, (6, 8, 8)),
                                         ("s2", (2, 2, 2), 8, 12, (6, 8, 8)),
                                         ("s122", (1, 2, 2), 5, 7, (5, 10, 6)),
                                         ("odd", (1, 1, 1), 13, 9, (4, 7, 9))):
        kw = {'kernel_size': (1, 3, 3), 'stride': stride, 'padding': (0, 1, 1), 'dilation': 1, 'bias': True}
        blk = ConvDropoutNormNonlin(cin, cout, nn.Conv3d, kw, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True},
                                    nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU,
                                    {'negative_slope': 1e-2, 'inplace': True})
        shapes = {n: tuple(p.shape) for n, p in blk.named_parameters()}
        params = closed_form_params(shapes)
        with torch.no_grad():
            for n, p in blk.named_parameters():
                p.copy_(params[n])
        x = seeded_input((2, cin) + shape, seed=11)
        x.requires_grad_(True)
        y = blk(x)
        gy = seeded_input(tuple(y.shape), seed=12)
        y.backward(gy)
        out[tag + "_y"] = y.detach().numpy()
        out[tag + "_dx"] = x.grad.numpy()
        for n, p in blk.named_parameters():
            out[tag + "_d_" + n] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "block.npz"), **out)


TINY = dict(patch=(16, 32, 32), cin=2, base=8, k=3, pools=[[2, 2, 2]] * 3 + [[1, 2, 2]] * 2, max_feat=32)


def gen_net_tiny():
    net = build_ref_net(TINY["patch"], TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], TINY["max_feat"])
    shapes = load_closed_form(net)
    x = seeded_input((2, TINY["cin"]) + TINY["patch"], seed=21)
    outs = net(x)
    targets = []
    for i, o in enumerate(outs):
        targets.append(seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=30 + i))
    w = np.array([1 / (2 ** i) for i in range(5)])
    w[-1] = 0
    w = w / w.sum()
    loss_fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), w)
    loss = loss_fn(outs, targets)
    loss.backward()
    out = {"loss": np.float64(loss.item()), "ds_weights": w}
    for i, o in enumerate(outs):
        out["logits%d" % i] = o.detach().numpy()
    names = list(shapes.keys())
    out["names"] = np.array(names)
    out["grad_l2"] = np.array([net.get_parameter(n).grad.double().norm().item() for n in names])
    for n in ("conv_blocks_context.0.blocks.0.conv.weight", "loc0.4.1.blocks.0.conv.weight", "up2.1.weight",
              "seg_outputs.0.weight", "loc3.0.0.blocks.0.instnorm.weight", "conv_blocks_context.5.1.blocks.0.conv.bias",
              "loc0.0.0.blocks.0.conv.weight"):
        out["grad::" + n] = net.get_parameter(n).grad.numpy()
    # eval / no-deep-supervision path returns the full-res logits only
    net.do_ds = False
    with torch.no_grad():
        out["logits_nods_sum"] = np.float64(net(x).double().sum().item())
    np.savez_compressed(os.path.join(OUT, "net_tiny.npz"), **out)


# the sparse training fixture runs on a patch whose bottleneck is 2x2x2 = 8 voxels: with TINY's 2x1x1 bottleneck the
# 2-voxel InstanceNorm amplifies last-ulp differences of the SECOND iteration beyond any useful tolerance
SPARSE_PATCH = (16, 64, 64)


def gen_net_sparse_tiny():
    """Tiny net with DSFF masks (density 0.3) applied: forward + two full train-steps with a prune/grow."""
    net = build_ref_net(SPARSE_PATCH, TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], TINY["max_feat"])
    shapes = load_closed_form(net)
    mask, opt = make_masking(net, density=0.3, death_rate=0.5, t_max=10, update_frequency=2, seed=5)
    out = {"names": np.array(list(mask.masks.keys()))}
    for n, m in mask.masks.items():
        out["mask0::" + n] = pack_kernel_mask(m)
    x = seeded_input((2, TINY["cin"]) + SPARSE_PATCH, seed=21)
    out["patch"] = np.array(SPARSE_PATCH)
    w = np.array([1 / (2 ** i) for i in range(5)])
    w[-1] = 0
    w = w / w.sum()
    loss_fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), w)
    import io, contextlib
    losses = []
    for it in range(2):                         # nnUNetTrainer_simple.run_iteration, non-AMP branch
        opt.zero_grad()
        outs = net(x)
        targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=30 + i)
                   for i, o in enumerate(outs)]
        loss = loss_fn(outs, targets)
        loss.backward()
        if it == 0:
            out["logits0_it0"] = outs[0].detach().numpy()[:, :, :, ::2, ::2]          # (1, 2, 2)-subsampled
            out["logits0_it0_sum"] = np.float64(outs[0].detach().double().sum().item())
            out["grad_l2_it0"] = np.array([p.grad.double().norm().item() for _, p in net.named_parameters()])
        tn = torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        opt.step()
        if it == 1:
            # weights (after the optimizer step, before Masking.step) that the prune/grow decision is taken on,
            # and the momentum buffers it masks: lets the GPU test replay exactly this update
            for n in mask.masks:
                out["pre_prune::" + n] = net.get_parameter(n).detach().numpy().copy()
        with contextlib.redirect_stdout(io.StringIO()):
            mask.step()
        losses.append(loss.item())
        out["total_norm_it%d" % it] = np.float64(tn.item())
        out["death_rate_it%d" % it] = np.float64(mask.death_rate)
    out["losses"] = np.array(losses)
    for n, m in mask.masks.items():                  # after the prune/grow at step 2
        out["mask2::" + n] = pack_kernel_mask(m)
    sd = net.state_dict()
    out["param_names"] = np.array(list(shapes.keys()))
    out["param_sum_after"] = np.array([sd[n].double().sum().item() for n in shapes])
    out["param_abs_after"] = np.array([sd[n].double().abs().sum().item() for n in shapes])
    for n in ("loc0.4.1.blocks.0.conv.weight", "up4.0.weight", "conv_blocks_context.0.blocks.0.conv.weight"):
        out["param_after::" + n] = sd[n].numpy()
    np.savez_compressed(os.path.join(OUT, "net_sparse_tiny.npz"), **out)


def gen_net64():
    """64^3, base 32, Cin 4, K 4, DSFF density 0.2 (random.seed(0)), closed-form weights."""
    pools = [[2, 2, 2]] * 5
    net = build_ref_net((64, 64, 64), 4, 32, 4, pools)
    load_closed_form(net)
    mask, _ = make_masking(net, density=0.2, seed=0)
    net.eval()
    x = seeded_input((1, 4, 64, 64, 64), seed=41)
    with torch.no_grad():
        outs = net(x)
    out = {}
    for i, o in enumerate(outs):
        o = o.double()
        out["sum%d" % i] = np.float64(o.sum().item())
        out["abs%d" % i] = np.float64(o.abs().sum().item())
    out["slice_d32"] = outs[0][0, :, 32].numpy()
    out["slice_h5"] = outs[0][0, :, :, 5].numpy()
    out["logits1"] = outs[1].numpy()[0, :, ::4]
    out["mask_names"] = np.array(list(mask.masks.keys()))
    out["mask_sha"] = np.array([sha_of(pack_kernel_mask(m)) for m in mask.masks.values()])
    out["mask_nnz"] = np.array([int(m.sum().item()) for m in mask.masks.values()])
    np.savez_compressed(os.path.join(OUT, "net64.npz"), **out)


HIPPO = dict(patch=(40, 56, 40), cin=1, k=3, pools=[[2, 2, 2]] * 3 + [[1, 1, 1]] * 2)


def _ds_loss(batch_dice=False):
    w = np.array([1 / (2 ** i) for i in range(5)])
    w[-1] = 0
    w = w / w.sum()
    return MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': batch_dice, 'smooth': 1e-5, 'do_bg': False}, {}), w)


def gen_net_hippo():
    """BASELINE config 1: Hippocampus-shaped plumbing case (5-pool plan [[2,2,2]]*3 + [[1,1,1]]*2, patch
    40x56x40, 1 modality, 3 classes, density 1.0 = no masks), B = 1.  Base 32: forward + loss + backward; base 48 (the
    reference trainer's hard-coded width): forward only."""
    out = {}
    net = build_ref_net(HIPPO["patch"], HIPPO["cin"], 32, HIPPO["k"], HIPPO["pools"])
    shapes = load_closed_form(net)
    x = seeded_input((1, HIPPO["cin"]) + HIPPO["patch"], seed=81)
    outs = net(x)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), HIPPO["k"], seed=90 + i) for i, o in enumerate(outs)]
    loss = _ds_loss()(outs, targets)
    loss.backward()
    out["loss"] = np.float64(loss.item())
    out["out_shapes"] = np.array([list(o.shape) for o in outs])
    for i, o in enumerate(outs):
        od = o.detach()
        out["b32_sum%d" % i] = np.float64(od.double().sum().item())
        out["b32_abs%d" % i] = np.float64(od.double().abs().sum().item())
        out["b32_logits%d" % i] = od.numpy()[:, :, ::2, ::2, ::2] if i == 0 else od.numpy()
    names = list(shapes.keys())
    out["names"] = np.array(names)
    out["grad_l2"] = np.array([net.get_parameter(n).grad.double().norm().item() for n in names])
    # full gradients of small tensors; the first 8 rows of the large ones that sit on the [1,1,1] stages
    for n in ("conv_blocks_context.0.blocks.0.conv.weight", "loc0.4.1.blocks.0.conv.weight", "up0.0.weight", "up1.0.weight",
              "seg_outputs.3.weight", "conv_blocks_context.4.blocks.0.conv.weight", "loc1.0.0.blocks.0.instnorm.weight"):
        out["grad::" + n] = net.get_parameter(n).grad.numpy()[:8]
    net48 = build_ref_net(HIPPO["patch"], HIPPO["cin"], 48, HIPPO["k"], HIPPO["pools"])
    load_closed_form(net48)
    net48.eval()
    net48.do_ds = False
    with torch.no_grad():
        o = net48(x)
    out["b48_sum"] = np.float64(o.double().sum().item())
    out["b48_abs"] = np.float64(o.double().abs().sum().item())
    out["b48_logits"] = o.numpy()[:, :, ::2, ::2, ::2]
    np.savez_compressed(os.path.join(OUT, "net_hippo.npz"), **out)


def gen_net_amos():
    """BASELINE config 5: AMOS-shaped net (1 modality, 16 classes, base 32, 64^3 patch) at DSFF density 0.1 and 0.5
    (random.seed(0)), closed-form weights: forward + deep-supervision loss + backward."""
    pools = [[2, 2, 2]] * 5
    out = {}
    x = seeded_input((1, 1, 64, 64, 64), seed=141)
    for dens in (0.1, 0.5):
        tag = "d%s" % dens
        net = build_ref_net((64, 64, 64), 1, 32, 16, pools)
        shapes = load_closed_form(net)
        mask, _ = make_masking(net, density=dens, seed=0)
        outs = net(x)
        targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), 16, seed=150 + i) for i, o in enumerate(outs)]
        loss = _ds_loss()(outs, targets)
        loss.backward()
        out[tag + "_loss"] = np.float64(loss.item())
        for i, o in enumerate(outs):
            od = o.detach().double()
            out[tag + "_sum%d" % i] = np.float64(od.sum().item())
            out[tag + "_abs%d" % i] = np.float64(od.abs().sum().item())
        out[tag + "_slice_d31"] = outs[0].detach().numpy()[0, :, 31, ::2, ::2]
        out[tag + "_logits3"] = outs[3].detach().numpy()
        names = list(shapes.keys())
        out["names"] = np.array(names)
        out[tag + "_grad_l2"] = np.array([net.get_parameter(n).grad.double().norm().item() for n in names])
        for n in ("loc4.0.0.blocks.0.conv.weight", "up0.4.weight", "seg_outputs.0.weight"):
            out[tag + "_grad::" + n] = net.get_parameter(n).grad.numpy()
        out[tag + "_mask_sha"] = np.array([sha_of(pack_kernel_mask(m)) for m in mask.masks.values()])
    np.savez_compressed(os.path.join(OUT, "net_amos.npz"), **out)


def gen_net_w48():
    """Width 48 -- the width the reference trainer hard-codes (nnUNetTrainer_simple.py:296) -- end to end: 64^3 patch, 4 modalities,
    4 classes, DSFF density 0.2 (random.seed(0); includes the `shape[0] == 48 => density 0.2` quirk of Masking.init), closed-form
    weights: forward with deep supervision + loss + backward.  Channel counts 48 / 96 / 192 / 320, concats of 96 / 144 / 240 / 480 ...:
    ragged 32-blocks in every matrix-pipe kernel."""
    pools = [[2, 2, 2]] * 5
    out = {}
    x = seeded_input((1, 4, 64, 64, 64), seed=241)
    net = build_ref_net((64, 64, 64), 4, 48, 4, pools)
    shapes = load_closed_form(net)
    mask, _ = make_masking(net, density=0.2, seed=0)
    outs = net(x)
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), 4, seed=250 + i) for i, o in enumerate(outs)]
    loss = _ds_loss()(outs, targets)
    loss.backward()
    out["loss"] = np.float64(loss.item())
    out["out_shapes"] = np.array([list(o.shape) for o in outs])
    for i, o in enumerate(outs):
        od = o.detach().double()
        out["sum%d" % i] = np.float64(od.sum().item())
        out["abs%d" % i] = np.float64(od.abs().sum().item())
    out["slice_d31"] = outs[0].detach().numpy()[0, :, 31, ::2, ::2]
    out["slice_h7"] = outs[0].detach().numpy()[0, :, ::2, 7, ::2]
    out["logits2"] = outs[2].detach().numpy()
    out["logits3"] = outs[3].detach().numpy()
    names = list(shapes.keys())
    out["names"] = np.array(names)
    out["grad_l2"] = np.array([net.get_parameter(n).grad.double().norm().item() for n in names])
    for n in ("conv_blocks_context.0.blocks.0.conv.weight", "loc0.4.1.blocks.0.conv.weight", "loc1.2.0.blocks.0.conv.weight", "up0.4.weight",
              "up2.0.weight", "seg_outputs.0.weight", "loc2.0.0.blocks.0.instnorm.weight"):
        g = net.get_parameter(n).grad.numpy()
        out["grad::" + n] = g[:8] if g.ndim > 1 else g
    out["mask_names"] = np.array(list(mask.masks.keys()))
    out["mask_sha"] = np.array([sha_of(pack_kernel_mask(m)) for m in mask.masks.values()])
    out["mask_nnz"] = np.array([int(m.sum().item()) for m in mask.masks.values()])
    np.savez_compressed(os.path.join(OUT, "net_w48.npz"), **out)


def gen_masks():
    """Initial uniform masks at the BASELINE widths + death-rate schedule + L1 association order."""
    out = {}
    pools = [[2, 2, 2]] * 5
    for base in (32, 48):
        net = build_ref_net((64, 64, 64), 4, base, 4, pools, seed=0)
        for dens in (0.1, 0.2, 0.5):
            mask, _ = make_masking(net, density=dens, seed=0)
            tag = "b%d_d%s" % (base, dens)
            out[tag + "_names"] = np.array(list(mask.masks.keys()))
            out[tag + "_sha"] = np.array([sha_of(pack_kernel_mask(m)) for m in mask.masks.values()])
            out[tag + "_nnz"] = np.array([int(m.sum().item()) for m in mask.masks.values()])
            if base == 32 and dens == 0.2:
                out[tag + "_loc4.0"] = pack_kernel_mask(mask.masks["loc4.0.0.blocks.0.conv.weight"])
                out[tag + "_up0.0"] = pack_kernel_mask(mask.masks["up0.0.weight"])
    decay = CosineDecay(0.5, 10)
    seq = []
    for _ in range(12):
        decay.step()
        seq.append(decay.get_dr())
    out["death_rate_T10"] = np.array(seq, dtype=np.float64)
    decay = CosineDecay(0.5, 250 * 1000)
    seq = []
    for _ in range(5):
        decay.step()
        seq.append(decay.get_dr())
    out["death_rate_T250k"] = np.array(seq, dtype=np.float64)
    # kernel L1 association order on closed-form weights (three chained sums, core_channel.py:652-655)
    from tests.helpers import closed_form_tensor
    for tag, shp in (("l1_133", (320, 896, 1, 3, 3)), ("l1_222", (64, 32, 2, 2, 2)), ("l1_122", (16, 24, 1, 2, 2))):
        wt = closed_form_tensor(shp, 3, "conv")
        s = torch.sum(torch.sum(torch.sum(torch.abs(wt), dim=-1), dim=-1), dim=-1)
        out[tag] = s.numpy()
    np.savez_compressed(os.path.join(OUT, "masks.npz"), **out)


def gen_grad_growth():
    """growth_mode='gradient' (Masking.kernel_grad_growth, core_channel.py:771-790) through the reference's own
    truncate_weights: closed-form weights and closed-form weight.grad on the tiny net (its up-sampling kernels are (2,2,2) and
    (1,2,2): the two-sum score keeps the kernel's depth extent), once with the gradients as they are and once after
    torch.nn.utils.clip_grad_norm_ scaled them in place (what nnUNetTrainer_simple.py:573 does before mask.step())."""
    from tests.helpers import closed_form_tensor
    out = {}
    pools = [[2, 2, 2]] * 3 + [[1, 2, 2]] * 2
    for tag, max_norm in (("raw", None), ("clip", 0.75)):
        net = build_ref_net((16, 32, 32), 2, 8, 3, pools, 32, seed=0)
        load_closed_form(net)
        opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
        args = _Args()
        random.seed(5)
        import io, contextlib
        mask = Masking(opt, death_rate=0.3, death_mode='magnitude', death_rate_decay=CosineDecay(0.3, 10),
                       growth_mode='gradient', redistribution_mode='none', args=args)
        with contextlib.redirect_stdout(io.StringIO()):
            mask.add_module(net, sparse_init='uniform', density=0.4)
        names = list(mask.masks.keys())
        out[tag + "_names"] = np.array(names)
        for n in names:
            out[tag + "_before::" + n] = pack_kernel_mask(mask.masks[n])
        # gradients: closed form per parameter (index offset 200), heavy-tailed so that the scores are well separated
        for i, (n, p_) in enumerate(net.named_parameters()):
            g = closed_form_tensor(tuple(p_.shape), 200 + i, "conv" if p_.dim() > 1 else "bias")
            p_.grad = (g * (1.0 + 3.0 * g.abs())).clone()
        if max_norm is not None:
            total = torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm)
            out[tag + "_total_norm"] = np.array([float(total)], dtype=np.float64)
            out[tag + "_max_norm"] = np.array([max_norm], dtype=np.float64)
            assert float(total) > max_norm
        with contextlib.redirect_stdout(io.StringIO()):
            mask.truncate_weights()
        for n in names:
            out[tag + "_after::" + n] = pack_kernel_mask(mask.masks[n])
            out[tag + "_num_death::" + n] = np.array([mask.num_death[n]])
    np.savez_compressed(os.path.join(OUT, "grad_growth.npz"), **out)


def gen_loss():
    out = {}
    for tag, batch_dice in (("sample", False), ("batch", True)):
        k = 4
        shapes = [(2, k, 8, 12, 10), (2, k, 4, 6, 5), (2, k, 2, 3, 5), (2, k, 1, 3, 5)]
        logits = [seeded_input(s, seed=50 + i).mul(2.0).requires_grad_(True) for i, s in enumerate(shapes)]
        targets = [seeded_labels((s[0], 1) + s[2:], k, seed=60 + i) for i, s in enumerate(shapes)]
        w = np.array([1 / (2 ** i) for i in range(5)])
        w[-1] = 0
        w = w / w.sum()
        loss_fn = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': batch_dice, 'smooth': 1e-5, 'do_bg': False}, {}), w)
        loss = loss_fn(logits, targets)
        loss.backward()
        out[tag + "_loss"] = np.float64(loss.item())
        for i, l in enumerate(logits):
            out[tag + "_g%d" % i] = l.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)


def gen_sliding():
    out = {}
    for ps in ((64, 64, 64), (128, 128, 128), (16, 32, 32), (40, 56, 40)):
        g = SegmentationNetwork._get_gaussian(ps, 1. / 8)
        tag = "g%dx%dx%d" % ps
        c = [i // 2 for i in ps]
        out[tag + "_line0"] = g[:, c[1], c[2]]
        out[tag + "_line2"] = g[c[0], c[1], :]
        out[tag + "_diag"] = np.array([g[i * ps[0] // 16, i * ps[1] // 16, i * ps[2] // 16] for i in range(16)])
        out[tag + "_stats"] = np.array([g.min(), g.max(), g.astype(np.float64).sum()], dtype=np.float64)
        if ps == (16, 32, 32):
            out[tag + "_full"] = g
    # full predict_3D on a volume that needs padding in x, several tiles in y and z, 8-fold mirroring
    net = build_ref_net(TINY["patch"], TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], TINY["max_feat"])
    load_closed_form(net)
    net.inference_apply_nonlin = lambda x: torch.nn.functional.softmax(x, 1)
    net.eval()
    net.do_ds = False
    vol = seeded_input((TINY["cin"], 13, 50, 70), seed=71).numpy()
    for tag, kw in (("tta", dict(do_mirroring=True, mirror_axes=(0, 1, 2))),
                    ("notta", dict(do_mirroring=False, mirror_axes=(0, 1, 2))),
                    ("tta01", dict(do_mirroring=True, mirror_axes=(0, 1)))):
        seg, probs = net.predict_3D(vol, use_sliding_window=True, step_size=0.5, patch_size=TINY["patch"],
                                    use_gaussian=True, all_in_gpu=False, verbose=False, mixed_precision=False, **kw)
        out["pred_%s_seg" % tag] = seg.astype(np.int8)
        out["pred_%s_probs_sum" % tag] = probs.astype(np.float64).sum(axis=(1, 2, 3))
        out["pred_%s_probs_slice" % tag] = probs[:, 6, ::2, ::2]
    np.savez_compressed(os.path.join(OUT, "sliding.npz"), **out)


def gen_dice():
    out = {}
    rng = np.random.RandomState(3)
    a = rng.randint(0, 4, (12, 14, 9))
    b = rng.randint(0, 4, (12, 14, 9))
    out["a"] = a.astype(np.int8)
    out["b"] = b.astype(np.int8)
    out["dice"] = np.array([ref_dice(a == l, b == l) for l in range(1, 4)])
    out["dice_small"] = np.float64(ref_dice(np.array([0, 1, 1, 0]), np.array([0, 1, 0, 0])))
    np.savez_compressed(os.path.join(OUT, "dice.npz"), **out)


def gen_evaluator():
    """The reference's aggregate_scores / Evaluator (e2enet/evaluation/evaluator.py:37-51, :216-226, :321-400; metrics.py) on three
    random label-map pairs, one of them with a label absent from both maps (NaN rules) and one with an all-foreground label."""
    import json
    import tempfile
    fo = types.ModuleType('batchgenerators.utilities.file_and_folder_operations')

    def save_json(obj, file, indent=4, sort_keys=True):
        with open(file, 'w') as f:
            json.dump(obj, f, sort_keys=sort_keys, indent=indent)
    fo.save_json, fo.join, fo.subfiles = save_json, os.path.join, lambda *a, **k: []
    sys.modules['batchgenerators.utilities'] = types.ModuleType('batchgenerators.utilities')
    sys.modules['batchgenerators.utilities.file_and_folder_operations'] = fo
    sys.modules.setdefault('SimpleITK', types.ModuleType('SimpleITK'))
    from e2enet.evaluation import evaluator as ev
    ev.Pool = lambda n: types.SimpleNamespace(map=lambda f, it: list(map(f, it)), close=lambda: None, join=lambda: None)
    rng = np.random.RandomState(11)
    out = {}
    pairs = []
    for c in range(3):
        a = rng.randint(0, 4, (9, 11, 7))
        b = np.where(rng.rand(9, 11, 7) < 0.7, a, rng.randint(0, 4, (9, 11, 7)))
        if c == 1:
            a[a == 3] = 0
            b[b == 3] = 0                    # label 3 absent from both maps
        if c == 2:
            a[:] = 2
            b[:4] = 2                        # label 2 fills the test map
        out["test%d" % c], out["ref%d" % c] = a.astype(np.int8), b.astype(np.int8)
        pairs.append((a, b))
    with tempfile.TemporaryDirectory() as tmp:
        jf = os.path.join(tmp, "summary.json")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scores = ev.aggregate_scores(pairs, evaluator=ev.Evaluator, labels=[0, 1, 2, 3], json_output_file=jf, json_name="n",
                                         json_task="t", num_threads=1)
        with open(jf) as f:
            js = json.load(f)
    out["summary_keys"] = np.array(sorted(js.keys()))
    out["metric_names"] = np.array(list(scores["all"][0]["0"].keys()))
    out["all"] = np.array([[[float(scores["all"][c][str(l)][m]) for m in out["metric_names"]] for l in range(4)] for c in range(3)])
    out["mean"] = np.array([[float(scores["mean"][str(l)][m]) for m in out["metric_names"]] for l in range(4)])
    np.savez_compressed(os.path.join(OUT, "evaluator.npz"), **out)


def gen_init():
    """Reference He init under torch.manual_seed(1234): per-tensor checksums (RNG draw order)."""
    out = {}
    for tag, (patch, cin, base, k, pools, mf) in {
            "tiny": (TINY["patch"], TINY["cin"], TINY["base"], TINY["k"], TINY["pools"], TINY["max_feat"]),
            "b32": ((64, 64, 64), 4, 32, 4, [[2, 2, 2]] * 5, None)}.items():
        net = build_ref_net(patch, cin, base, k, pools, mf, seed=1234)
        sd = net.state_dict()
        out[tag + "_names"] = np.array(list(sd.keys()))
        out[tag + "_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        out[tag + "_sum"] = np.array([v.double().sum().item() for v in sd.values()])
        out[tag + "_abs"] = np.array([v.double().abs().sum().item() for v in sd.values()])
        out[tag + "_param_order"] = np.array([n for n, _ in net.named_parameters()])
    np.savez_compressed(os.path.join(OUT, "init.npz"), **out)



VARIANT = dict(patch=(16, 16, 64), cin=2, base=8, k=3, pools=[[2, 2, 2], [2, 2, 2], [1, 2, 2], [2, 1, 2], [1, 1, 2]], max_feat=32)


def gen_net_variants():
    """the ablation networks unetpp_d_313.py / unetpp_d_331.py (conv kernel (3,1,3) / (3,3,1), shift switched
    off in their source).  Anisotropic pooling plan so that a wrong axis permutation cannot hide; B = 2; closed-form
    weights: forward + deep-supervision loss + backward; plus the He-init checksums under torch.manual_seed(1234)."""
    import importlib
    out = {}
    V = VARIANT
    for var in ("313", "331"):
        mod = importlib.import_module("e2enet.network_architecture.unetpp_d_" + var)

        def build(seed=None):
            if seed is not None:
                torch.manual_seed(seed)
            return mod.Generic_UNetPlusPlus(V["patch"], V["cin"], V["base"], V["k"], len(V["pools"]), 2, 2, nn.Conv3d,
                                            nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                                            {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True},
                                            True, False, lambda x: x, mod.InitWeights_He(1e-2), V["pools"], None, False, True,
                                            True, max_num_features=V["max_feat"])
        net = build()
        shapes = load_closed_form(net)
        x = seeded_input((2, V["cin"]) + V["patch"], seed=121)
        outs = net(x)
        targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), V["k"], seed=130 + i) for i, o in enumerate(outs)]
        loss = _ds_loss()(outs, targets)
        loss.backward()
        out[var + "_loss"] = np.float64(loss.item())
        for i, o in enumerate(outs):
            od = o.detach()
            out[var + "_sum%d" % i] = np.float64(od.double().sum().item())
            out[var + "_logits%d" % i] = od.numpy()[..., ::2, ::2] if i == 0 else od.numpy()
        names = list(shapes.keys())
        out[var + "_names"] = np.array(names)
        out[var + "_shapes"] = np.array([str(shapes[n]) for n in names])
        out[var + "_grad_l2"] = np.array([net.get_parameter(n).grad.double().norm().item() for n in names])
        for n in ("conv_blocks_context.0.blocks.0.conv.weight", "loc0.4.1.blocks.0.conv.weight", "up0.0.weight", "up2.1.weight",
                  "up4.0.weight", "seg_outputs.0.weight", "loc3.0.0.blocks.0.instnorm.weight", "loc0.0.0.blocks.0.conv.weight"):
            out[var + "_grad::" + n] = net.get_parameter(n).grad.numpy()
        sd = build(seed=1234).state_dict()
        out[var + "_init_names"] = np.array(list(sd.keys()))
        out[var + "_init_sum"] = np.array([v.double().sum().item() for v in sd.values()])
        out[var + "_init_abs"] = np.array([v.double().abs().sum().item() for v in sd.values()])
        # position-sensitive checksum (a permuted tensor has the same sum): sum of value * (flat index + 1) / numel
        out[var + "_init_pos"] = np.array([(v.double().flatten() * (torch.arange(v.numel(), dtype=torch.float64) + 1)).sum().item() / v.numel()
                                           for v in sd.values()])
    np.savez_compressed(os.path.join(OUT, "net_variants.npz"), **out)


def gen_net_nodff():
    """the 'shiftConvPP_nodff' ablation (unetpp_d_nodff.py:171-353, selected at
    nnUNetTrainer_simple.py:326-335): a plain U-Net (no nested dense fusion) of the same shift-conv blocks with shift size 3;
    five deep-supervision outputs.  TINY plan, B = 2, closed-form weights: forward + deep-supervision loss + backward, the
    He-init checksums under torch.manual_seed(1234), and the tensors the reference's Masking selects."""
    import importlib
    mod = importlib.import_module("e2enet.network_architecture.unetpp_d_nodff")
    T = TINY

    def build(seed=None):
        if seed is not None:
            torch.manual_seed(seed)
        return mod.Generic_UNetPlusPlus(T["patch"], T["cin"], T["base"], T["k"], len(T["pools"]), 2, 2, nn.Conv3d,
                                        nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                                        {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True},
                                        True, False, lambda x: x, mod.InitWeights_He(1e-2), T["pools"], None, False, True,
                                        True, max_num_features=T["max_feat"])
    out = {}
    net = build()
    shapes = load_closed_form(net)
    x = seeded_input((2, T["cin"]) + T["patch"], seed=221)
    outs = net(x)
    assert len(outs) == 5
    targets = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), T["k"], seed=230 + i) for i, o in enumerate(outs)]
    from e2enet.training.loss_functions.dice_loss import DC_and_CE_loss
    from e2enet.training.loss_functions.deep_supervision import MultipleOutputLoss2
    # the trainer's weights for net_numpool = 5 outputs (nnUNetTrainer_simple.py:207-213): last one masked to 0
    w = np.array([1 / (2 ** i) for i in range(5)])
    w[-1] = 0
    w = w / w.sum()
    loss = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), w)(outs, targets)
    loss.backward()
    out["ds_weights"] = w
    out["loss"] = np.float64(loss.item())
    out["out_shapes"] = np.array([list(o.shape) for o in outs])
    for i, o in enumerate(outs):
        out["logits%d" % i] = o.detach().numpy()[..., ::2, ::2] if i == 0 else o.detach().numpy()
        out["sum%d" % i] = np.float64(o.detach().double().sum().item())
    names = list(shapes.keys())
    out["names"] = np.array(names)
    out["shapes"] = np.array([str(shapes[n]) for n in names])
    out["grad_l2"] = np.array([0.0 if net.get_parameter(n).grad is None else net.get_parameter(n).grad.double().norm().item() for n in names])
    for n in ("conv_blocks_context.0.blocks.0.conv.weight", "conv_blocks_localization.4.1.blocks.0.conv.weight", "tu.0.weight",
              "tu.4.weight", "seg_outputs.4.weight", "seg_outputs.1.weight", "conv_blocks_localization.2.0.blocks.0.instnorm.weight",
              "conv_blocks_context.5.1.blocks.0.conv.weight"):
        out["grad::" + n] = net.get_parameter(n).grad.numpy()
    sd = build(seed=1234).state_dict()
    out["init_names"] = np.array(list(sd.keys()))
    out["init_sum"] = np.array([v.double().sum().item() for v in sd.values()])
    out["init_abs"] = np.array([v.double().abs().sum().item() for v in sd.values()])
    # names the reference's Masking picks on this network, and its uniform masks at density 0.3 under random.seed(0)
    net2 = build(seed=7)
    mask, _ = make_masking(net2, 0.3)
    out["masked_names"] = np.array(list(mask.masks.keys()))
    out["mask_sha"] = np.array([sha_of(pack_kernel_mask(m)) for m in mask.masks.values()])
    np.savez_compressed(os.path.join(OUT, "net_nodff.npz"), **out)


def gen_export():
    """save_segmentation_nifti_from_softmax (segmentation_export.py:27-160) on volumes that need no resampling, with the
    SimpleITK writer, skimage and the batchgenerators file helpers stubbed (absent here): the uint8 array handed to the
    NIfTI writer is captured.  Ensemble inputs: three seeded 'fold' softmax volumes (summed and averaged with the numpy
    expressions of predict.py:282-296 by the oracle and the GPU path; the reference has no callable for that step)."""
    import os.path as osp
    captured = {}

    class _Img:
        def SetSpacing(self, *a): pass
        def SetOrigin(self, *a): pass
        def SetDirection(self, *a): pass
    sitk = types.ModuleType('SimpleITK')
    sitk.GetImageFromArray = lambda arr: (captured.__setitem__('arr', np.array(arr)), _Img())[1]
    sitk.WriteImage = lambda img, fname: None
    sys.modules['SimpleITK'] = sitk
    ff = types.ModuleType('batchgenerators.utilities.file_and_folder_operations')
    ff.isfile, ff.join, ff.isdir = osp.isfile, osp.join, osp.isdir
    ff.os = os
    ff.save_pickle = lambda *a, **k: None
    for n in ('subfiles', 'subdirs', 'maybe_mkdir_p', 'load_pickle', 'write_pickle', 'save_json', 'load_json'):
        setattr(ff, n, lambda *a, **k: None)
    sys.modules['batchgenerators.utilities'] = types.ModuleType('batchgenerators.utilities')
    sys.modules['batchgenerators.utilities.file_and_folder_operations'] = ff
    sys.modules['batchgenerators.augmentations.utils'].resize_segmentation = lambda *a, **k: None
    sk = types.ModuleType('skimage'); skt = types.ModuleType('skimage.transform'); skt.resize = lambda *a, **k: None
    sys.modules['skimage'] = sk; sys.modules['skimage.transform'] = skt
    for n in ('skimage.measure', 'skimage.morphology'):
        sys.modules[n] = types.ModuleType(n)
    from e2enet.inference.segmentation_export import save_segmentation_nifti_from_softmax
    rng = np.random.RandomState(11)
    out = {}
    k, shp = 5, (9, 14, 11)                                  # network axis order
    folds = [rng.rand(k, *shp).astype(np.float32) for _ in range(3)]
    folds = [f / f.sum(0, keepdims=True) for f in folds]
    for i, f in enumerate(folds):
        out["fold%d" % i] = f
    total = folds[0].copy()
    for f in folds[1:]:
        total += f
    total /= len(folds)                                      # predict.py:295-296
    for tag, tb, regions in (("plain", [0, 1, 2], None), ("transposed", [2, 0, 1], None), ("regions", [1, 0, 2], (1, 3, 2))):
        sm = total.transpose([0] + [i + 1 for i in tb])      # predict.py:298-301
        size = sm.shape[1:]
        props = {'size_after_cropping': np.array(size), 'original_size_of_raw_data': np.array([size[0] + 3, size[1] + 1, size[2] + 4]),
                 'crop_bbox': [[2, 2 + size[0]], [0, size[1]], [3, 3 + size[2]]], 'original_spacing': np.array([1., 1., 1.]),
                 'spacing_after_resampling': np.array([1., 1., 1.]), 'itk_spacing': (1., 1., 1.), 'itk_origin': (0., 0., 0.),
                 'itk_direction': (1., 0., 0., 0., 1., 0., 0., 0., 1.)}
        save_segmentation_nifti_from_softmax(sm.copy(), "/tmp/_x.nii.gz", props, 1, regions, None, None, None, None, None, 0,
                                             verbose=False)
        out[tag + "_seg"] = captured['arr'].astype(np.uint8)
        out[tag + "_tb"] = np.array(tb)
        if regions is not None:
            out[tag + "_regions"] = np.array(regions)
    np.savez_compressed(os.path.join(OUT, "export.npz"), **out)



def gen_dataloader():
    """DataLoader3D (e2enet/training/dataloading/dataset_loading.py:163-388) on three synthetic cases under np.random.seed:
    three batches each for 'edge' and 'constant' data padding, a third of every batch forced onto foreground."""
    import tempfile
    from tests.helpers import synthetic_cases
    dl = types.ModuleType('batchgenerators.dataloading.data_loader')

    class SlimDataLoaderBase:                  # batchgenerators 0.24 base class (third party): holds the dataset and batch size
        def __init__(self, data, batch_size, number_of_threads_in_multithreaded=None):
            self._data, self.batch_size, self.thread_id = data, batch_size, 0
    dl.SlimDataLoaderBase = SlimDataLoaderBase
    sys.modules['batchgenerators.dataloading'] = types.ModuleType('batchgenerators.dataloading')
    sys.modules['batchgenerators.dataloading.data_loader'] = dl
    ff = types.ModuleType('batchgenerators.utilities.file_and_folder_operations')
    import pickle
    ff.os, ff.isfile, ff.join, ff.isdir = os, os.path.isfile, os.path.join, os.path.isdir
    ff.load_pickle = lambda f, mode='rb': pickle.load(open(f, mode))
    for n in ('subfiles', 'subdirs', 'maybe_mkdir_p', 'write_pickle', 'save_pickle', 'save_json', 'load_json'):
        setattr(ff, n, lambda *a, **k: None)
    sys.modules['batchgenerators.utilities'] = types.ModuleType('batchgenerators.utilities')
    sys.modules['batchgenerators.utilities.file_and_folder_operations'] = ff
    from e2enet.training.dataloading.dataset_loading import DataLoader3D
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        ds = synthetic_cases(tmp)
        for tag, mode, kw in (("edge", "edge", None), ("const", "constant", {'constant_values': 0})):
            np.random.seed(1234)
            loader = DataLoader3D(ds, (16, 18, 20), (12, 14, 16), 4, False, oversample_foreground_percent=0.33,
                                  pad_mode=mode, pad_kwargs_data=kw)
            for it in range(3):
                b = loader.generate_train_batch()
                out["%s_data%d" % (tag, it)] = b['data'].astype(np.float32)
                out["%s_seg%d" % (tag, it)] = b['seg'].astype(np.int8)
                out["%s_keys%d" % (tag, it)] = np.array([str(k) for k in b['keys']])
            out[tag + "_rng_after"] = np.random.randint(0, 2 ** 31 - 1, 4)          # the stream position after three batches
    np.savez_compressed(os.path.join(OUT, "dataloader.npz"), **out)


ALL = dict(dataloader=gen_dataloader, export=gen_export, shift=gen_shift, block=gen_block, net=gen_net_tiny, sparse=gen_net_sparse_tiny, net64=gen_net64,
           hippo=gen_net_hippo, amos=gen_net_amos, w48=gen_net_w48,
           variants=gen_net_variants, nodff=gen_net_nodff, masks=gen_masks, grad_growth=gen_grad_growth, evaluator=gen_evaluator, loss=gen_loss, sliding=gen_sliding, dice=gen_dice, init=gen_init)

if __name__ == "__main__":
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    todo = sys.argv[1:] or list(ALL)
    for name in todo:
        print("generating", name, flush=True)
        ALL[name]()
    print("done")
