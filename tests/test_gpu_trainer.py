"""Trainer surface, data-parallel wiring on a single-rank RCCL group, and the reference's literal
iteration body on the autograd path."""
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from tests.helpers import golden, seeded_input, seeded_labels, pack_kernel_mask
from tests.test_gpu_net import tiny_net, TINY, SPARSE_PATCH

pytestmark = pytest.mark.gpu

PLANS = {'plans_per_stage': {0: {'batch_size': 2, 'patch_size': [16, 32, 32], 'num_pool_per_axis': [3, 5, 5],
                                 'pool_op_kernel_sizes': [[2, 2, 2]] * 3 + [[1, 2, 2]] * 2,
                                 'conv_kernel_sizes': [[3, 3, 3]] * 6, 'do_dummy_2D_data_aug': False}},
         'base_num_features': 32, 'num_modalities': 1, 'num_classes': 2, 'all_classes': [1, 2],
         'transpose_forward': [0, 1, 2], 'transpose_backward': [0, 1, 2], 'conv_per_stage': 2}


class _Args:
    adv = False
    fix = False
    update_frequency = 2
    final_density = 0.05


def _trainer(out=None, epochs=2, batch_dice=False, fold=0):
    from e2enet_medical_amd.training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
    tr = nnUNetTrainer_simple(PLANS, fold, output_folder=out, batch_dice=batch_dice, Tconv='shiftConvPP',
                              max_num_epochs=epochs, num_batches_per_epoch=2)
    tr.base_num_features_override = 8
    tr.num_val_batches_per_epoch = 2
    torch.manual_seed(0)
    tr.synthetic_data = True
    net, opt = tr.initialize(True)
    return tr, net, opt


def _masking(net, opt, t_max=8, density=0.2, seed=0):
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    random.seed(seed)
    mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, t_max),
                   growth_mode='random', redistribution_mode='none', args=_Args())
    mask.add_module(net, sparse_init='uniform', density=density)
    return mask


@pytest.fixture
def rccl_single_rank():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)      # "nccl" is RCCL on ROCm
    os.environ["E2E_FORCE_DIST"] = "1"
    yield
    os.environ.pop("E2E_FORCE_DIST", None)
    if created:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ run_training surface
def test_run_training_writes_reference_named_checkpoints_and_resumes(tmp_path):
    """run_training: epoch loop with validation + online evaluation, poly LR, moving averages, {Tconv}_model_best /
    _final_checkpoint files (+ .pkl), latest removed after the final save (reference :1020-1025); the final checkpoint's
    epoch equals the loss history's length; load_final / load_latest / load_best find the files; a resumed trainer gets a
    finite learning rate; checkpoints written by the loop carry the DSFF state."""
    base = str(tmp_path)
    tr, net, opt = _trainer(base, epochs=3)
    out = tr.output_folder
    assert out == os.path.join(base, "fold_0")                  # update_fold in the constructor (reference :111)
    tr.save_every = 1
    mask = _masking(net, opt)
    tr.run_training(mask)
    assert len(tr.all_tr_losses) == 3 and len(tr.all_val_losses) == 3 and len(tr.all_val_eval_metrics) == 3
    assert all(np.isfinite(tr.all_tr_losses)) and all(0.0 <= m <= 1.0 for m in tr.all_val_eval_metrics)
    assert tr.train_loss_MA is not None and tr.val_eval_criterion_MA is not None and tr.best_val_eval_criterion_MA is not None
    final = os.path.join(out, "shiftConvPP_model_final_checkpoint.model")
    assert os.path.isfile(final) and os.path.isfile(final + ".pkl")
    assert not os.path.exists(os.path.join(out, "shiftConvPP_model_latest.model"))
    ck = torch.load(final, map_location="cpu", weights_only=False)
    assert ck['epoch'] == len(ck['plot_stuff'][0]) == 3
    assert 'dsff_state' in ck and ck['dsff_state']['steps'] == mask.steps
    # resume: same folder layout the reference's model_restore uses (output_folder ends with fold_0)
    tr2, net2, opt2 = _trainer(base, epochs=6)
    mask2 = _masking(net2, opt2, seed=99)
    tr2.load_final_checkpoint(train=True, mask=mask2)
    assert tr2.epoch == 3
    tr2.maybe_update_lr(tr2.epoch)
    assert np.isfinite(opt2.param_groups[0]['lr']) and opt2.param_groups[0]['lr'] > 0
    for n in mask.kmasks:
        assert torch.equal(mask.kmasks[n], mask2.kmasks[n])
    for k, v in net.state_dict().items():
        assert torch.equal(v, net2.state_dict()[k]), k
    tr3, _, _ = _trainer(base)
    tr3.load_latest_checkpoint(train=False)           # falls through to the final checkpoint (prefixed name)
    assert tr3.epoch == 3
    tr3.load_best_checkpoint(train=False)
    # update_fold (model_restore.py:142): swap folds for ensembling
    tr3.update_fold(1)
    assert tr3.fold == 1 and tr3.output_folder.endswith("fold_1") and tr3.output_folder_pretrained == tr3.output_folder
    with pytest.raises(RuntimeError):
        tr3.load_final_checkpoint()
    with pytest.raises(NotImplementedError):         # preprocessing stays with the reference package
        tr3.preprocess_patient(["a.nii.gz"])
    with pytest.raises(FileNotFoundError):           # validate() runs on the engine, but this trainer was fed synthetic batches
        tr3.validate()


def test_online_evaluation_counts_match_torch():
    """run_online_evaluation (reference :371-405) on the HIP kernel against the reference's torch expressions."""
    tr, net, opt = _trainer()
    batch = next(tr.val_gen)
    tr.network.eval()
    tr.run_iteration(iter([batch]), False, True)
    eng = net.engine(batch['data'].cuda())
    logits = eng.heads[0].out.data
    target = batch['target'][0].cuda()[:, 0]
    seg = F.softmax(logits, 1).argmax(1)
    k = logits.shape[1]
    tp = np.array([((seg == c).float() * (target == c).float()).sum().item() for c in range(1, k)])
    fp = np.array([((seg == c).float() * (target != c).float()).sum().item() for c in range(1, k)])
    fn = np.array([((seg != c).float() * (target == c).float()).sum().item() for c in range(1, k)])
    assert np.array_equal(np.array(tr.online_eval_tp[-1]), tp)
    assert np.array_equal(np.array(tr.online_eval_fp[-1]), fp)
    assert np.array_equal(np.array(tr.online_eval_fn[-1]), fn)
    tr.finish_online_evaluation()
    want = np.mean([2 * a / (2 * a + b + c) for a, b, c in zip(tp, fp, fn)])
    assert abs(tr.all_val_eval_metrics[-1] - want) < 1e-6          # the reference keeps the counts in float32


def test_online_evaluation_public_signature_and_deferred_prefetch_error():
    """(a) run_online_evaluation(output, target) with foreign logits (the reference's public signature, K != input channels)
    counts straight from those tensors and builds no activation plan for them; (b) an exception raised by the generator
    while run_iteration fetches one batch ahead surfaces on the NEXT call, after the applied step's loss was returned."""
    tr, net, opt = _trainer()
    plans_before = len(net._engines)
    g = torch.Generator().manual_seed(5)
    logits = torch.randn((2, 7, 6, 10, 12), generator=g).cuda()
    target = torch.randint(0, 7, (2, 1, 6, 10, 12), generator=g).float().cuda()
    tr.run_online_evaluation([logits], [target])
    assert len(net._engines) == plans_before
    seg = logits.argmax(1)
    tgt = target[:, 0]
    tp = np.array([((seg == c) & (tgt == c)).sum().item() for c in range(1, 7)], dtype=np.float32)
    fn = np.array([((seg != c) & (tgt == c)).sum().item() for c in range(1, 7)], dtype=np.float32)
    assert np.array_equal(np.array(tr.online_eval_tp[-1]), tp) and np.array_equal(np.array(tr.online_eval_fn[-1]), fn)

    batch = next(tr.tr_gen)

    def gen():
        yield batch
        raise ValueError("augmenter failed")
    it = gen()
    tr.prefetch_batches = True
    loss = tr.run_iteration(it, True)                    # step applied, the one-ahead fetch fails quietly
    assert np.isfinite(loss)
    with pytest.raises(ValueError, match="augmenter failed"):
        tr.run_iteration(it, True)
    tr.drain_prefetched()
    assert not tr._prefetched and not tr._prefetch_error


def test_validation_loss_value_matches_training_loss():
    tr, net, opt = _trainer(batch_dice=True)
    batch = next(tr.tr_gen)
    v = float(tr.run_iteration(iter([batch]), False))
    eng = net.engine(batch['data'].cuda())
    assert eng.grads == {} or not eng._backward_ready          # a validation batch allocates no gradient buffers
    t = float(tr.run_iteration(iter([batch]), True))
    assert abs(v - t) < 1e-6


# ------------------------------------------------------------------------------------------------ ADVICE round 1
def test_masks_then_checkpoint_load_uses_the_loaded_weights():
    """simple_main order: Masking (fresh random masks) first, then a checkpoint is loaded (:163-177).  The reference's
    conv is dense, so the loaded weights act in full until the next apply_mask; the engine's liveness tables must not
    hide kernels that now hold weights."""
    tr, net, opt = _trainer()
    dense = {k: v.detach().clone() for k, v in net.state_dict().items()}      # dense He-init weights
    x = seeded_input((2, 1, 16, 32, 32), seed=3).cuda()
    net.eval()
    with torch.no_grad():
        want = [o.clone() for o in net(x)]
    mask = _masking(net, opt)                                                 # zeroes 80 % of the fusion kernels
    with torch.no_grad():
        sparse = [o.clone() for o in net(x)]
    assert (sparse[0] - want[0]).abs().max() > 1e-3
    ck = {'epoch': 0, 'state_dict': dense, 'optimizer_state_dict': None, 'plot_stuff': ([], [], [], []),
          'best_stuff': (None, None, None)}
    tr.load_checkpoint_ram(ck, train=True)
    with torch.no_grad():
        got = net(x)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    mask.apply_mask()                                                         # the next mask application re-prunes
    with torch.no_grad():
        again = net(x)
    assert torch.equal(again[0], sparse[0])
    # inference-mode load: auto sparsity takes precedence over stale masks
    tr.load_checkpoint_ram(ck, train=False)
    with torch.no_grad():
        got = net(x)
    assert torch.equal(got[0], want[0])


def test_backward_after_second_forward_raises():
    net, _, _ = tiny_net()
    x = seeded_input((1, TINY["cin"]) + TINY["patch"], seed=1).cuda()
    net.train()
    out1 = net(x)
    net(x * 2)
    with pytest.raises(RuntimeError, match="another forward"):
        out1[0].sum().backward()


def test_out_of_range_label_poisons_the_loss():
    net, _, _ = tiny_net()
    x = seeded_input((1, TINY["cin"]) + TINY["patch"], seed=1).cuda()
    eng = net.engine(x)
    outs = eng.forward(x, True)
    targets = [seeded_labels((1, 1) + tuple(o.shape[2:]), TINY["k"], seed=5 + i).cuda() for i, o in enumerate(outs)]
    assert np.isfinite(eng.loss_value(targets, oracle.ds_weights(5)).item())
    targets[0][0, 0, 0, 0, 0] = -1.0
    assert np.isnan(eng.loss_value(targets, oracle.ds_weights(5)).item())


# ------------------------------------------------------------------------------------------------ reference iteration body
def test_reference_literal_iteration_body_on_the_autograd_path():
    """The reference's run_iteration body, verbatim (nnUNetTrainer_simple.py:566-576): output = network(data);
    l = loss(output, target); l.backward(); clip_grad_norm_(parameters, 12); optimizer.step(); mask.step() -- against the
    reference's own two iterations (golden net_sparse_tiny: losses, clip norms, death rates, masks after the prune/grow)."""
    from e2enet_medical_amd.training.loss_functions.dice_loss import DC_and_CE_loss
    from e2enet_medical_amd.training.loss_functions.deep_supervision import MultipleOutputLoss2
    g = golden("net_sparse_tiny.npz")
    network, shapes, _ = tiny_net(SPARSE_PATCH)
    optimizer = torch.optim.SGD(network.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
    random.seed(5)
    mask = Masking(optimizer, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 10),
                   growth_mode='random', redistribution_mode='none', args=_Args())
    mask.add_module(network, sparse_init='uniform', density=0.3)
    loss = MultipleOutputLoss2(DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {}), oracle.ds_weights(5))
    data = seeded_input((2, TINY["cin"]) + SPARSE_PATCH, seed=21).cuda()
    network.train()
    for it in range(2):
        optimizer.zero_grad()
        output = network(data)
        target = [seeded_labels((o.shape[0], 1) + tuple(o.shape[2:]), TINY["k"], seed=30 + i).cuda() for i, o in enumerate(output)]
        l = loss(output, target)
        l.backward()
        tn = torch.nn.utils.clip_grad_norm_(network.parameters(), 12)
        optimizer.step()
        mask.step()
        assert abs(l.item() - g["losses"][it]) <= 5e-5
        assert abs(tn.item() - float(g["total_norm_it%d" % it])) <= 1e-3 * float(g["total_norm_it%d" % it])
        assert mask.death_rate == float(g["death_rate_it%d" % it])
    names = [str(s) for s in g["names"]]
    nnz = {n: int(mask.masks[n].sum().item()) for n in names}
    assert nnz == {n: int(np.unpackbits(g["mask2::" + n]).sum()) * int(np.prod(mask.masks[n].shape[-3:])) for n in names}
    sd = network.state_dict()
    got_abs = np.array([sd[n].double().abs().sum().item() for n in shapes])
    np.testing.assert_allclose(got_abs, g["param_abs_after"], rtol=5e-3, atol=1e-3)


# ------------------------------------------------------------------------------------------------ data parallel, one rank
def test_trainer_data_parallel_path_single_rank_rccl(rccl_single_rank):
    """The product's data-parallel iteration (bucketed RCCL gradient all-reduce under the backward pass, global batch
    dice, mask broadcast after the prune/grow) on a one-rank group gives exactly the single-process result."""
    import torch.distributed as dist
    os.environ.pop("E2E_FORCE_DIST")
    tr0, net0, opt0 = _trainer(batch_dice=True)
    mask0 = _masking(net0, opt0)
    batches = [next(tr0.tr_gen) for _ in range(3)]
    l0 = [float(tr0.run_iteration(iter([b]), True, mask=mask0)) for b in batches]
    os.environ["E2E_FORCE_DIST"] = "1"
    tr1, net1, opt1 = _trainer(batch_dice=True)
    mask1 = _masking(net1, opt1)
    mask1.force_sync = True
    calls = {"bcast": 0, "allreduce": 0}
    orig_b, orig_a = dist.broadcast, dist.all_reduce
    dist.broadcast = lambda *a, **k: (calls.__setitem__("bcast", calls["bcast"] + 1), orig_b(*a, **k))[1]
    dist.all_reduce = lambda *a, **k: (calls.__setitem__("allreduce", calls["allreduce"] + 1), orig_a(*a, **k))[1]
    try:
        l1 = [float(tr1.run_iteration(iter([b]), True, mask=mask1)) for b in batches]
    finally:
        dist.broadcast, dist.all_reduce = orig_b, orig_a
    assert l0 == l1
    assert calls["bcast"] == 1                                   # one prune/grow (update_frequency 2, 3 iterations)
    assert calls["allreduce"] >= 3 * (1 + 4)                     # per iteration: >= 1 gradient bucket + 4 batch-dice sums
    for k, v in net0.state_dict().items():
        assert torch.equal(v, net1.state_dict()[k]), k
    for n in mask0.kmasks:
        assert torch.equal(mask0.kmasks[n], mask1.kmasks[n])
    # validation batch with online evaluation under DP: counts summed over the ranks
    tr1.run_iteration(tr1.val_gen, False, True)
    assert len(tr1.online_eval_tp) == 1


@pytest.mark.parametrize("tag,kw", [("tta", dict(do_mirroring=True, mirror_axes=(0, 1, 2)))])
def test_sharded_predict_3d_branch_single_rank_rccl(rccl_single_rank, tag, kw):
    """The sharded branch of predict_3D (partition -> RCCL all-gather -> ordered overlap-add) executed for real on a
    one-rank group: bit-identical to the unsharded loop, and within the bars of the reference golden."""
    from e2enet_medical_amd.utilities.nd_softmax import softmax_helper
    g = golden("sliding.npz")
    net, _, _ = tiny_net()
    net.inference_apply_nonlin = softmax_helper
    net.eval()
    net.do_ds = False
    vol = seeded_input((TINY["cin"], 13, 50, 70), seed=71).numpy()
    args = dict(use_sliding_window=True, step_size=0.5, patch_size=TINY["patch"], use_gaussian=True, all_in_gpu=False,
                verbose=False, mixed_precision=False, **kw)
    seg0, probs0 = net.predict_3D(vol, **args)
    net.shard_tiles(0, 1, None, force=True)
    seg1, probs1 = net.predict_3D(vol, **args)
    assert np.array_equal(seg0, seg1) and np.array_equal(probs0, probs1)
    ref_seg = g["pred_%s_seg" % tag].astype(np.int64)
    assert (seg1 != ref_seg).mean() < 1e-3
    assert np.abs(probs1[:, 6, ::2, ::2] - g["pred_%s_probs_slice" % tag]).max() <= 2e-5
    # the other exchange (round 6): per-rank partial volumes + ONE RCCL all-reduce (weight map accumulated through the NULL-patch
    # mode of e2e_sw_accumulate); a different summation order is allowed, 1e-6 is the bar of the gloo test
    net.shard_tiles(0, 1, None, force=True, exchange="allreduce")
    seg2, probs2 = net.predict_3D(vol, **args)
    assert np.abs(probs2 - probs0).max() <= 1e-6
    assert (seg2 != seg0).mean() < 1e-4


# ------------------------------------------------------------------------------------------------ N1: ensemble + export
@pytest.mark.parametrize("tag", ["plain", "transposed", "regions"])
def test_export_kernels_match_reference_golden(tag):
    """e2e_ensemble_accumulate + e2e_export_argmax_u8 against the reference's own output (golden export.npz): fold sum in
    order, float32 division, transpose_backward, first-maximum argmax / region thresholds, crop-box placement."""
    from e2enet_medical_amd._lib import lib
    from e2enet_medical_amd.inference.predict import export_segmentation
    g = golden("export.npz")
    folds = [torch.from_numpy(g["fold%d" % i]).cuda() for i in range(3)]
    total = folds[0].clone()
    for i, f in enumerate(folds[1:]):
        lib().ensemble_accumulate(total.data_ptr(), f.data_ptr(), total.numel(), 0, 3 if i == 1 else 0, 0)
    assert np.array_equal(total.cpu().numpy(), oracle.ensemble_softmax([g["fold%d" % i] for i in range(3)]))
    tb = [int(v) for v in g[tag + "_tb"]]
    size = [total.shape[1 + i] for i in tb]
    props = {'size_after_cropping': np.array(size), 'original_size_of_raw_data': np.array([size[0] + 3, size[1] + 1, size[2] + 4]),
             'crop_bbox': [[2, 2 + size[0]], [0, size[1]], [3, 3 + size[2]]]}
    regions = tuple(int(v) for v in g[tag + "_regions"]) if tag + "_regions" in g.files else None
    seg = export_segmentation(total, props, tb, regions)
    assert seg.dtype == np.uint8 and np.array_equal(seg, g[tag + "_seg"])


@pytest.mark.parametrize("tb,new_shape,spacing,expect_axis", [
    ([0, 1, 2], (19, 40, 33), (1.0, 1.0, 1.0), None),          # isotropic: trilinear over all axes
    ([2, 0, 1], (25, 17, 50), (1.2, 1.0, 0.9), None),          # with transpose_backward, up- and down-sampling mixed
    ([0, 1, 2], (9, 44, 35), (5.0, 0.8, 0.8), 0),              # anisotropic: nearest along axis 0, bilinear in the plane
    ([1, 0, 2], (30, 36, 7), (0.7, 0.7, 3.0), 2),
    ([0, 1, 2], (12, 31, 28), (4.0, 1.0, 1.0), 0),             # the separate axis keeps its size: slices only
])
def test_export_resamples_softmax_to_original_grid(tb, new_shape, spacing, expect_axis):
    """N1, segmentation_export.py:84-104: a softmax volume whose grid differs from size_after_cropping is resampled on the
    device (order 1; order 0 along one separate low-resolution axis) before argmax and crop-box placement.  Checked against
    the oracle's scipy restatement of resample_data_or_seg / skimage resize (parity unpinned: oracle/export.py)."""
    from e2enet_medical_amd.inference.predict import export_segmentation, resample_softmax, resample_plan
    rng = np.random.RandomState(5)
    logits = rng.standard_normal((4, 12, 28, 31)).astype(np.float32) * 2
    soft = np.exp(logits) / np.exp(logits).sum(0, keepdims=True)
    soft = soft.astype(np.float32)
    new_shape = tuple(new_shape)
    props = {'size_after_cropping': np.array(new_shape), 'original_size_of_raw_data': np.array([new_shape[0] + 2, new_shape[1] + 1, new_shape[2]]),
             'crop_bbox': [[1, 1 + new_shape[0]], [0, new_shape[1]], [0, new_shape[2]]],
             'original_spacing': np.array(spacing), 'spacing_after_resampling': np.array([1.0, 1.0, 1.0])}
    sep, axis = resample_plan(props)
    assert axis == expect_axis and sep == (expect_axis is not None)
    dev = torch.from_numpy(soft).cuda()
    got = resample_softmax(dev, new_shape, tb, axis).cpu().numpy()
    want = oracle.export.resample_softmax(soft.transpose([0] + [i + 1 for i in tb]), new_shape, axis)
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.abs(got - want).max() <= 1e-6
    assert (got != want).mean() <= 1e-3                     # same arithmetic, same order: bit-identical but for rare last-bit cases
    seg = export_segmentation(dev, props, tb, None)
    ref = oracle.export_segmentation(soft, props, tb, None, lowres_axis=axis)
    assert seg.shape == ref.shape and (seg != ref).mean() <= 1e-5
    with pytest.raises(NotImplementedError):
        export_segmentation(dev, props, tb, None, order=3)


def test_predict_cases_fold_ensemble_on_device():
    """inference.predict.predict_cases: two 'folds' (checkpoints), sliding window with mirroring per fold, ensemble and
    export on the device == the host route (predict_3D per fold -> numpy sum / n -> oracle export)."""
    from e2enet_medical_amd.inference.predict import predict_cases
    tr, net, opt = _trainer()
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    torch.manual_seed(123)
    sd1 = {k: (v + 0.05 * torch.randn_like(v)) for k, v in sd0.items()}
    params = [{'epoch': 0, 'state_dict': sd, 'optimizer_state_dict': None, 'plot_stuff': ([], [], [], []),
               'best_stuff': (None, None, None)} for sd in (sd0, sd1)]
    tr.plans = dict(tr.plans)                        # (PLANS is the module's shared dict: no leak into later tests)
    tr.plans['transpose_forward'], tr.plans['transpose_backward'] = [1, 2, 0], [2, 0, 1]
    vol = seeded_input((1, 20, 40, 45), seed=9).numpy()
    props = {'size_after_cropping': np.array([40, 45, 20])[[0, 1, 2]], 'original_size_of_raw_data': np.array([22, 41, 50]),
             'crop_bbox': None}
    # host route
    host = []
    for p in params:
        tr.load_checkpoint_ram(p, False)
        host.append(tr.predict_preprocessed_data_return_seg_and_softmax(vol, do_mirroring=True, mirror_axes=(0, 1, 2),
                                                                       use_sliding_window=True, step_size=0.5, use_gaussian=True,
                                                                       verbose=False)[1])
    total = oracle.ensemble_softmax(host)
    tb = tr.plans['transpose_backward']
    size = [total.shape[1 + i] for i in tb]
    props = {'size_after_cropping': np.array(size), 'original_size_of_raw_data': np.array([size[0] + 2, size[1], size[2] + 5]),
             'crop_bbox': [[1, 1 + size[0]], [0, size[1]], [4, 4 + size[2]]]}
    want = oracle.export_segmentation(total, props, tb, None)
    got = {}
    done = predict_cases(tr, params, [("case0.nii.gz", (vol, props))], lambda seg, fn, dct: got.__setitem__(fn, seg))
    assert done == ["case0.nii.gz"] and np.array_equal(got["case0.nii.gz"], want)


# ------------------------------------------------------------------------------------------------ N4: Tconv ablations
@pytest.mark.parametrize("tconv,variant,shift", [("shiftConvPP_313", "313", 1), ("shiftConvPP_331", "331", 1),
                                                 ("shiftConvPP_noshift", "133", 1)])
def test_trainer_builds_and_steps_the_tconv_ablations(tmp_path, tconv, variant, shift):
    """initialize_network dispatches on Tconv like the reference (:303-346); one training iteration on the fast path
    (data and targets moved to the engine's axis order by run_iteration) gives the loss of the same network evaluated by
    the oracle in the reference's axis order, and the checkpoint written afterwards has the reference's tensor shapes."""
    from e2enet_medical_amd.training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
    plans = dict(PLANS)
    plans['plans_per_stage'] = {0: dict(PLANS['plans_per_stage'][0], patch_size=[16, 16, 64],
                                        pool_op_kernel_sizes=[[2, 2, 2], [2, 2, 2], [1, 2, 2], [2, 1, 2], [1, 1, 2]])}
    tr = nnUNetTrainer_simple(plans, 0, output_folder=str(tmp_path), Tconv=tconv, max_num_epochs=1, num_batches_per_epoch=1)
    tr.base_num_features_override = 8
    torch.manual_seed(0)
    tr.synthetic_data = True
    net, opt = tr.initialize(True)
    assert net.conv_variant == variant and net._cfg.shift_size == shift
    pools = [tuple(k) for k in plans['plans_per_stage'][0]['pool_op_kernel_sizes']]
    spec = oracle.make_spec(1, 8, 3, pools, 2, 320, shift_size=shift, conv_variant=variant)
    sd = {n: v.detach().cpu().clone() for n, v in net.state_dict().items()}
    assert {n: tuple(v.shape) for n, v in sd.items()} == oracle.param_shapes(spec)
    x = seeded_input((2, 1, 16, 16, 64), seed=171)
    with torch.no_grad():
        ref_outs = oracle.forward(spec, sd, x)
    targets = [seeded_labels((2, 1) + tuple(o.shape[2:]), 3, seed=180 + i) for i, o in enumerate(ref_outs)]
    ref_loss = oracle.deep_supervision_loss(ref_outs, targets, oracle.ds_weights(5), tr.batch_dice).item()

    def gen():
        while True:
            yield {'data': x.clone(), 'target': [t.clone() for t in targets]}
    loss = tr.run_iteration(gen(), do_backprop=True, run_online_evaluation=False)
    assert abs(float(loss) - ref_loss) < 2e-5
    fname = os.path.join(tr.output_folder, "ck.model")
    tr.save_checkpoint(fname)
    saved = torch.load(fname, map_location="cpu", weights_only=False)['state_dict']
    assert {n: tuple(v.shape) for n, v in saved.items()} == oracle.param_shapes(spec)


def test_trainer_builds_ds_targets_on_device_from_full_resolution_labels(tmp_path):
    """A generator that yields only the full-resolution label map (N3: the DownsampleSegForDSTransform2 step moved to the
    device) gives the same iteration loss as one that yields the oracle's list of downsampled targets."""
    tr, net, opt = _trainer(str(tmp_path), epochs=1)
    x = seeded_input((2, 1, 16, 32, 32), seed=191)
    full = seeded_labels((2, 1, 16, 32, 32), 3, seed=192)
    scales = tr.deep_supervision_scales[:4]
    lists = [torch.from_numpy(a) for a in oracle.downsample_seg_for_ds(full.numpy(), scales)]

    def gen(targets):
        while True:
            yield {'data': x.clone(), 'target': targets}
    a = tr.run_iteration(gen(full.clone()), do_backprop=False)
    b = tr.run_iteration(gen([t.clone() for t in lists]), do_backprop=False)
    assert float(a) == float(b)


def test_run_iteration_prefetches_the_next_batch_without_changing_the_sequence(tmp_path):
    """run_iteration fetches batch i+1 and uploads it on a copy stream under the kernels of batch i: the losses are those
    of the synchronous path batch for batch, the generator is consumed exactly one batch ahead, a second generator does
    not see the first one's batch, and an exhausted generator ends cleanly."""
    batches = []
    for i in range(4):
        x = seeded_input((2, 1, 16, 32, 32), seed=300 + i)
        t = [seeded_labels((2, 1) + s, 3, seed=310 + i) for s in ((16, 32, 32), (8, 16, 16), (4, 8, 8), (2, 4, 4))]
        batches.append({'data': x.numpy(), 'target': [v.numpy() for v in t]})     # numpy, like the reference's augmenters

    def run(prefetch):
        tr, net, opt = _trainer(str(tmp_path / ("p%d" % prefetch)), epochs=1)
        tr.prefetch_batches = bool(prefetch)
        taken = []

        def gen(tag):
            for i, b in enumerate(batches):
                taken.append((tag, i))
                yield b
        g1, g2 = gen("a"), gen("b")
        out = [float(tr.run_iteration(g1, do_backprop=True)) for _ in range(2)]
        n_after_two = len([t for t in taken if t[0] == "a"])
        out.append(float(tr.run_iteration(g2, do_backprop=False)))          # another generator: its own first batch
        out += [float(tr.run_iteration(g1, do_backprop=True)) for _ in range(2)]
        with pytest.raises(StopIteration):
            tr.run_iteration(g1, do_backprop=True)                           # 4 batches consumed, nothing prefetched
        return out, n_after_two
    sync, n_sync = run(0)
    pre, n_pre = run(1)
    assert sync == pre
    assert n_sync == 2 and n_pre == 3


def _write_cases(folder, n_cases=6, shape=(20, 44, 40), mods=1):
    """preprocessed cases in the reference's on-disk form: <case>.npy ([modalities..., seg]) + <case>.pkl (properties with
    class_locations), closed forms"""
    import pickle
    from collections import OrderedDict
    os.makedirs(folder, exist_ok=True)
    for ci in range(n_cases):
        shp = tuple(s + 2 * (ci % 3) for s in shape)
        j = np.arange(int(np.prod(shp)), dtype=np.float64).reshape(shp)
        data = [np.sin(0.21 * j + 0.7 * ci + m).astype(np.float32) * (1 + m) for m in range(mods)]
        zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shp], indexing="ij")
        seg = (((zz // 3 + yy // 5 + xx // 4 + ci) % 5) < 2).astype(np.float32) + (((zz + yy + xx) % 17) == 0).astype(np.float32)
        seg = np.minimum(seg, 2.0)
        seg[:2] = -1                                           # nnU-Net marks voxels outside the nonzero mask with -1
        np.save(os.path.join(folder, "case_%02d.npy" % ci), np.stack(data + [seg]).astype(np.float32))
        locs = OrderedDict((c, np.argwhere(seg == c)) for c in (1, 2))
        with open(os.path.join(folder, "case_%02d.pkl" % ci), "wb") as f:
            pickle.dump(OrderedDict(class_locations=locs, name="case_%02d" % ci), f)


@pytest.mark.parametrize("dummy_2d", [False, True])
def test_trainer_initialize_feeds_real_cases_through_loader_and_device_augmenter(tmp_path, dummy_2d):
    """initialize(training=True) as the reference wires it (nnUNetTrainer_simple.py:216-239): load_dataset + do_split over the stage
    folder, DataLoader3D x 2, get_moreDA_augmentation (here: DeviceAugmenter on the GPU) -- also for a `do_dummy_2D_data_aug` plan,
    the anisotropic-patch mode BASELINE config 3's shape switches on -- and run_iteration trains on those batches.  Without the
    folder initialize() raises instead of silently training on noise."""
    from e2enet_medical_amd.training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
    from e2enet_medical_amd.training.data_augmentation.data_augmentation_moreDA import _DeviceGenerator
    plans = dict(PLANS, data_identifier="nnUNetData_plans_v2.1")
    plans['plans_per_stage'] = {0: dict(PLANS['plans_per_stage'][0], do_dummy_2D_data_aug=dummy_2d)}
    root = tmp_path / "Task998"
    tr0 = nnUNetTrainer_simple(plans, 0, output_folder=str(tmp_path / "out0"), dataset_directory=str(root), Tconv='shiftConvPP')
    tr0.base_num_features_override = 8
    with pytest.raises(FileNotFoundError):
        tr0.initialize(True)
    _write_cases(str(root / "nnUNetData_plans_v2.1_stage0"), mods=PLANS['num_modalities'])
    tr = nnUNetTrainer_simple(plans, 0, output_folder=str(tmp_path / "out"), dataset_directory=str(root), batch_dice=False,
                              Tconv='shiftConvPP', max_num_epochs=1, num_batches_per_epoch=2)
    tr.base_num_features_override = 8
    torch.manual_seed(0)
    np.random.seed(0)
    net, opt = tr.initialize(True)
    assert isinstance(tr.tr_gen, _DeviceGenerator) and isinstance(tr.val_gen, _DeviceGenerator)
    assert len(tr.dataset_tr) + len(tr.dataset_val) == 6 and len(tr.dataset_val) >= 1
    assert bool(tr.data_aug_params["dummy_2D"]) == dummy_2d
    patch = tuple(int(v) for v in tr.patch_size)
    if dummy_2d:
        assert int(tr.basic_generator_patch_size[0]) == patch[0]
    b = next(tr.tr_gen)
    assert tuple(b["data"].shape) == (tr.batch_size, tr.num_input_channels) + patch and b["data"].is_cuda
    assert tuple(b["target"][0].shape) == (tr.batch_size, 1) + patch and float(b["target"][0].min()) >= 0
    assert len(b["target"]) == tr._num_ds_outputs()
    v = next(tr.val_gen)
    assert tuple(v["data"].shape) == (tr.batch_size, tr.num_input_channels) + patch
    losses = [float(tr.run_iteration(tr.tr_gen, True)) for _ in range(3)]
    assert all(np.isfinite(losses)), losses
    assert np.isfinite(float(tr.run_iteration(tr.val_gen, False, True)))


def test_validate_exports_and_scores_the_validation_split(tmp_path):
    """nnUNetTrainer_simple.validate (reference :1309-1479): every validation case -> sliding-window prediction -> device export to
    the case's original geometry (crop box, one resampled case) -> writer; scored against the ground-truth volumes with the restated
    confusion-matrix metrics; summary.json in the reference's structure.  The exported label maps equal the host route (predict ->
    oracle export); the Dice in summary.json equals oracle.hard_dice on the same volumes."""
    import json
    import pickle
    from tests.helpers import write_synthetic_task
    from e2enet_medical_amd.training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
    ddir, plans = write_synthetic_task(str(tmp_path / "pre"), plans=dict(PLANS, transpose_forward=[0, 1, 2], transpose_backward=[0, 1, 2]))
    tr = nnUNetTrainer_simple(plans, 0, output_folder=str(tmp_path / "out"), dataset_directory=ddir, batch_dice=False,
                              Tconv='shiftConvPP', max_num_epochs=1, num_batches_per_epoch=2)
    tr.base_num_features_override = 8
    torch.manual_seed(0)
    np.random.seed(0)
    tr.initialize(True)
    assert tr.gt_niftis_folder == os.path.join(ddir, "gt_segmentations") and len(tr.dataset_val) >= 1
    written = {}
    scores = tr.validate(do_mirroring=True, save_softmax=True, writer=lambda seg, path, props: written.__setitem__(path, seg.copy()))
    out = os.path.join(tr.output_folder, "validation_raw")
    js = json.load(open(os.path.join(out, "summary.json")))
    assert sorted(js.keys()) == ['author', 'description', 'id', 'name', 'results', 'task', 'timestamp']
    assert js["name"] == "nnUNetTrainer_simple val tiled True" and js["task"] == "Task998_Synth" and js["description"] == ""
    assert os.path.isfile(os.path.join(out, "validation_args.json"))
    assert len(js["results"]["all"]) == len(tr.dataset_val) == len(written)
    for k, rec in zip(tr.dataset_val.keys(), js["results"]["all"]):
        props = pickle.load(open(tr.dataset[k]['properties_file'], 'rb'))
        path = os.path.join(out, k + ".nii.gz")
        seg = written[path]
        assert rec["test"] == path and seg.dtype == np.uint8 and tuple(seg.shape) == tuple(props['original_size_of_raw_data'])
        gt = np.load(os.path.join(ddir, "gt_segmentations", k + ".npy"))
        for label in (1, 2):
            d = oracle.hard_dice(seg, gt, label)
            assert abs(rec[str(label)]["Dice"] - d) < 1e-12
        # host route: the same prediction through the oracle's export
        data = np.load(tr.dataset[k]['data_file'][:-4] + ".npy")
        _, sm = tr.predict_preprocessed_data_return_seg_and_softmax(data[:-1], do_mirroring=True, mirror_axes=(0, 1, 2), verbose=False)
        ref_seg = oracle.export_segmentation(sm, props, plans['transpose_backward'])
        assert (seg != ref_seg).mean() <= 1e-4
        sm16 = np.load(os.path.join(out, k + ".npz"))["softmax"]
        assert sm16.dtype == np.float16 and sm16.shape == sm.shape
    assert set(scores["mean"].keys()) == {"0", "1", "2"}
    assert os.path.isdir(os.path.join(tr.output_folder_base, "gt_niftis"))


def test_simple_main_and_simple_predict_take_the_reference_argv(tmp_path, monkeypatch):
    """python -m e2enet_medical_amd.simple_main / .simple_predict with the reference's argv (simple_main.py:34-105, simple_predict.py:
    26-128): two epochs on a synthetic task folder with DSFF, the checkpoints the reference names ({Tconv}_model_final_checkpoint.model
    + .pkl, plans.pkl), -c continues with the saved masks, --validation_only writes summary.json, simple_predict predicts the folder
    from the written checkpoint (case sharding by --part_id / --num_parts) and reproduces validate()'s label maps."""
    from tests.helpers import write_synthetic_task
    from e2enet_medical_amd import simple_main, simple_predict
    pre, res = tmp_path / "pre", tmp_path / "res"
    monkeypatch.setenv("nnUNet_preprocessed", str(pre))
    monkeypatch.setenv("RESULTS_FOLDER", str(res))
    monkeypatch.setenv("nnUNet_raw_data_base", str(tmp_path / "raw"))
    ddir, plans = write_synthetic_task(str(pre), plans=dict(PLANS, transpose_forward=[0, 1, 2], transpose_backward=[0, 1, 2]))
    common = ["--task", "998", "--fold", "0", "--Tconv", "shiftConvPP", "--base_num_features", "8", "--sparse", "True", "--density",
              "0.5", "--update_frequency", "2", "--death-rate", "0.3"]
    random.seed(3)
    torch.manual_seed(3)
    np.random.seed(3)
    tr = simple_main.main(common + ["--max_num_epochs", "2", "--num_batches_per_epoch", "2"])
    fold_dir = os.path.join(str(res), "nnUNet", "3d_fullres", "Task998_Synth", "nnUNetTrainerV2__nnUNetPlansv2.1", "fold_0")
    assert tr.output_folder == fold_dir
    for f in ("shiftConvPP_model_final_checkpoint.model", "shiftConvPP_model_final_checkpoint.model.pkl"):
        assert os.path.isfile(os.path.join(fold_dir, f)), f
    assert os.path.isfile(os.path.join(os.path.dirname(fold_dir), "plans.pkl"))
    assert len(tr.all_tr_losses) == 2 and all(np.isfinite(tr.all_tr_losses))
    ck = torch.load(os.path.join(fold_dir, "shiftConvPP_model_final_checkpoint.model"), weights_only=False)
    assert 'dsff_state' in ck and ck['dsff_state']['steps'] == 4
    # -c: one more epoch from the written checkpoint, masks and schedule position restored
    tr2 = simple_main.main(common + ["--max_num_epochs", "3", "--num_batches_per_epoch", "2", "-c"])
    assert len(tr2.all_tr_losses) == 3 and tr2._mask.steps == 6
    # --validation_only: loads the final checkpoint, validates, writes summary.json
    tr3 = simple_main.main(common + ["--validation_only", "True", "--val_folder", "val_cli"])
    vdir = os.path.join(fold_dir, "val_cli")
    assert os.path.isfile(os.path.join(vdir, "summary.json"))
    val_keys = list(tr3.dataset_val.keys())
    # simple_predict over the stage folder (preprocessed cases), two parts like two processes would
    stage = os.path.join(ddir, plans['data_identifier'] + "_stage0")
    outp = str(tmp_path / "pred")
    done = []
    for part in (0, 1):
        done += simple_predict.main(["-i", stage, "-o", outp, "-t", "998", "-f", "0", "--Tconv", "shiftConvPP", "--part_id", str(part),
                                     "--num_parts", "2"])
    assert sorted(os.path.basename(d) for d in done) == ["case_%02d.nii.gz" % i for i in range(6)]
    assert os.path.isfile(os.path.join(outp, "plans.pkl"))
    for k in val_keys:
        a = np.load(os.path.join(outp, k + ".npy"))
        b = np.load(os.path.join(vdir, k + ".npy"))
        assert a.dtype == np.uint8 and np.array_equal(a, b), k
    with pytest.raises(NotImplementedError, match="preprocessing"):
        os.makedirs(str(tmp_path / "rawcases"))
        open(os.path.join(str(tmp_path / "rawcases"), "c_0000.nii.gz"), "wb").close()
        simple_predict.main(["-i", str(tmp_path / "rawcases"), "-o", outp, "-t", "998", "-f", "0", "--Tconv", "shiftConvPP",
                             "--overwrite_existing"])
