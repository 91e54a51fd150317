"""CPU-side tests: the C-ABI library loads and exports every symbol include/e2e_hip.h declares (no compute calls),
host logic of the drop-in surface, init RNG parity with the reference, and the N>1 exchange paths on gloo."""
import os
import re
import sys
import subprocess
import numpy as np
import pytest
import torch
from torch import nn

from tests.helpers import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from e2enet_medical_amd._lib import lib, SIGNATURES, LIB_PATH
    hdr = open(os.path.join(ROOT, "include", "e2e_hip.h")).read()
    declared = set(re.findall(r"\b(e2e_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert declared == set(SIGNATURES.keys()), declared ^ set(SIGNATURES.keys())
    handle = lib()                                   # dlopen + getattr on every symbol
    assert os.path.samefile(handle.path, LIB_PATH)
    from e2enet_medical_amd._lib import ABI_VERSION
    assert handle.abi_version() == ABI_VERSION
    assert handle.conv133_num_partials(128, 128, 128, 1, 1) == 128 * 8 * 4          # 16x32 output tiles
    assert handle.loss_ws_bytes(2, 4) == (2 * 4 * 3 + 1) * 8


def _build(patch, cin, base, k, pools, mf, seed=None):
    from e2enet_medical_amd.network_architecture.unetpp_d import Generic_UNetPlusPlus
    from e2enet_medical_amd.network_architecture.initialization import InitWeights_He
    if seed is not None:
        torch.manual_seed(seed)
    return Generic_UNetPlusPlus(patch, cin, base, k, 5, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True},
                                nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU,
                                {'negative_slope': 1e-2, 'inplace': True}, True, False, lambda x: x, InitWeights_He(1e-2),
                                pools, None, False, True, True, max_num_features=mf)


@pytest.mark.parametrize("tag,args", [("tiny", ((16, 32, 32), 2, 8, 3, [[2, 2, 2]] * 3 + [[1, 2, 2]] * 2, 32)),
                                      ("b32", ((64, 64, 64), 4, 32, 4, [[2, 2, 2]] * 5, None))])
def test_module_tree_and_init_match_reference_bit_for_bit(tag, args):
    """state_dict names/shapes (checkpoint wire format), named_parameters order (Masking draw order) and the
    He-normal initial weights under torch.manual_seed(1234) equal the reference's."""
    g = golden("init.npz")
    net = _build(*args, seed=1234)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(s) for s in g[tag + "_names"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g[tag + "_shapes"]]
    assert [n for n, _ in net.named_parameters()] == [str(s) for s in g[tag + "_param_order"]]
    assert np.array_equal(np.array([v.double().sum().item() for v in sd.values()]), g[tag + "_sum"])
    assert np.array_equal(np.array([v.double().abs().sum().item() for v in sd.values()]), g[tag + "_abs"])


def test_network_attributes_and_errors():
    net = _build((16, 32, 32), 2, 8, 3, [[2, 2, 2]] * 3 + [[1, 2, 2]] * 2, 32)
    assert net.conv_op == nn.Conv3d and net.num_classes == 3 and net.do_ds and net._deep_supervision
    assert list(net.input_shape_must_be_divisible_by) == [8, 32, 32]
    assert len(net.td) == 0 and len(net.down0) == 4 and len(net.up0) == 5 and len(net.loc4) == 1
    with pytest.raises(RuntimeError):                       # product path never computes on the CPU
        net(torch.zeros(1, 2, 16, 32, 32))
    with pytest.raises(ValueError):                         # reference forward() needs exactly 6 levels
        from e2enet_medical_amd.network_architecture.unetpp_d import Generic_UNetPlusPlus
        Generic_UNetPlusPlus((40, 56, 40), 1, 8, 3, 3, conv_op=nn.Conv3d, norm_op=nn.InstanceNorm3d,
                             dropout_op_kwargs={'p': 0}, convolutional_pooling=True, convolutional_upsampling=True)
    from e2enet_medical_amd.engine import shift_amounts
    import oracle
    for c in (1, 4, 12, 32, 64, 160, 896):
        assert shift_amounts(c) == oracle.shift_amounts(c)


def test_sliding_window_host_logic_matches_reference_vectors():
    from e2enet_medical_amd.network_architecture.neural_network import SegmentationNetwork, pad_nd_image
    cs = SegmentationNetwork._compute_steps_for_sliding_window
    assert cs((128, 128, 128), (146, 176, 148), 0.5) == [[0, 18], [0, 48], [0, 20]]
    assert cs((128, 128, 128), (424, 456, 456), 0.5) == [[0, 59, 118, 178, 237, 296], [0, 55, 109, 164, 219, 273, 328],
                                                         [0, 55, 109, 164, 219, 273, 328]]
    assert cs((64, 192, 192), (94, 308, 308), 0.5) == [[0, 30], [0, 58, 116], [0, 58, 116]]
    assert cs((40, 56, 40), (40, 56, 40), 0.5) == [[0], [0], [0]]
    g = golden("sliding.npz")
    m = SegmentationNetwork._get_gaussian((16, 32, 32))
    assert np.array_equal(m, g["g16x32x32_full"])
    x = np.arange(2 * 13 * 5 * 40, dtype=np.float32).reshape(2, 13, 5, 40)
    padded, slicer = pad_nd_image(x, (16, 32, 32), "constant", {'constant_values': 0}, True, None)
    assert padded.shape == (2, 16, 32, 40)
    assert [(s.start, s.stop) for s in slicer] == [(0, 2), (1, 14), (13, 18), (0, 40)]
    assert np.array_equal(padded[tuple(slicer)], x)


def test_trainer_and_masking_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from e2enet_medical_amd.training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
    plans = {'plans_per_stage': {0: {'batch_size': 1, 'patch_size': [16, 32, 32], 'num_pool_per_axis': [3, 5, 5],
                                     'pool_op_kernel_sizes': [[2, 2, 2]] * 3 + [[1, 2, 2]] * 2,
                                     'conv_kernel_sizes': [[3, 3, 3]] * 6, 'do_dummy_2D_data_aug': False}},
             'base_num_features': 32, 'num_modalities': 1, 'num_classes': 2, 'all_classes': [1, 2], 'conv_per_stage': 2}
    tr = nnUNetTrainer_simple(plans, 0, batch_dice=False, Tconv='shiftConvPP')
    tr.base_num_features_override = 8
    tr.synthetic_data = True
    tr.initialize(True)
    assert tr.ds_loss_weights.tolist() == pytest.approx([8 / 15, 4 / 15, 2 / 15, 1 / 15, 0])
    assert tr.deep_supervision_scales[:3] == [[1, 1, 1], [0.5, 0.5, 0.5], [0.25, 0.25, 0.25]]
    with pytest.raises(RuntimeError):
        tr.run_iteration(tr.tr_gen, True)
    from e2enet_medical_amd.training.loss_functions.dice_loss import DC_and_CE_loss
    with pytest.raises(RuntimeError):
        DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False}, {})(torch.zeros(1, 3, 2, 2, 2), torch.zeros(1, 1, 2, 2, 2))


def test_missing_library_is_an_import_error(tmp_path):
    code = ("import sys; sys.path.insert(0, %r); import e2enet_medical_amd._lib as L; L.LIB_PATH = %r;\n"
            "try:\n    L.lib()\nexcept ImportError as e:\n    print('IMPORT_ERROR')\n" % (ROOT, str(tmp_path / "nope.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "IMPORT_ERROR" in out.stdout, out.stderr


def test_every_e2e_knob_read_anywhere_is_listed_and_unknown_ones_are_refused():
    """INTEGRATION.md section 7 is the one list of knobs: every E2E_* variable the library or the host code reads is in
    _lib.KNOWN_ENV and in that table, nothing else is; a variable outside the list is reported when the library loads (a warning
    -- "E2E_" is a common harness prefix -- or, under E2E_STRICT_ENV=1, a refusal)."""
    import glob
    from e2enet_medical_amd import _lib
    read = set()
    for f in glob.glob(os.path.join(ROOT, "e2enet_medical_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "e2enet_medical_amd", "csrc", "*.h")):
        read |= set(re.findall(r'getenv\("(E2E_[A-Z0-9_]+)"\)', open(f).read()))
    for f in glob.glob(os.path.join(ROOT, "e2enet_medical_amd", "**", "*.py"), recursive=True) + [os.path.join(ROOT, "bench.py")]:
        read |= set(re.findall(r'environ(?:\.get\(|\[)"(E2E_[A-Z0-9_]+)"', open(f).read()))
    assert read == set(_lib.KNOWN_ENV), read ^ set(_lib.KNOWN_ENV)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = set(re.findall(r"^\| `(E2E_[A-Z0-9_]+)`", doc, flags=re.M))
    assert table == set(_lib.KNOWN_ENV), table ^ set(_lib.KNOWN_ENV)
    _lib.check_env({"E2E_CONV_MM": "0", "PATH": "/bin"})
    with pytest.warns(RuntimeWarning, match="E2E_BASE_URL"):
        _lib.check_env({"E2E_BASE_URL": "http://localhost"})
    with pytest.raises(RuntimeError, match="E2E_CT_TPX32"):
        _lib.check_env({"E2E_CT_TPX32": "1", "E2E_STRICT_ENV": "1"})
    code = "import sys; sys.path.insert(0, %r); import e2enet_medical_amd._lib as L; L.lib()" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, E2E_WG_V3_TARGET="64", E2E_STRICT_ENV="1"))
    assert out.returncode != 0 and "unknown E2E_* environment variable(s) E2E_WG_V3_TARGET" in out.stderr, out.stderr
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, E2E_WG_V3_TARGET="64"))
    assert out.returncode == 0 and "E2E_WG_V3_TARGET" in out.stderr, out.stderr


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import oracle
from oracle import network as onet
from tests.helpers import closed_form_params, seeded_input
from e2enet_medical_amd import parallel
from e2enet_medical_amd.network_architecture.neural_network import SegmentationNetwork, pad_nd_image

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.set_num_threads(2)
pools = [(2, 2, 2)] * 3 + [(1, 2, 2)] * 2
spec = oracle.make_spec(2, 8, 3, pools, 2, 32)
params = closed_form_params(onet.param_shapes(spec))
patch = (16, 32, 32)
vol = seeded_input((2, 13, 40, 50), seed=71).numpy()

def net_fn(t):
    with torch.no_grad():
        return torch.softmax(oracle.forward(spec, params, t, do_ds=False), 1)

# single-process reference (oracle), no mirroring to keep the test short
seg_ref, probs_ref = oracle.predict_tiled(net_fn, vol, 3, patch, 0.5, do_mirroring=False, use_gaussian=True)

# sharded: the product's partition / exchange / ordered overlap-add, with the oracle standing in for the GPU forward
data, slicer = pad_nd_image(vol, patch, "constant", {'constant_values': 0}, True, None)
steps = SegmentationNetwork._compute_steps_for_sliding_window(patch, data.shape[1:], 0.5)
tiles = [(a, b, c) for a in steps[0] for b in steps[1] for c in steps[2]]
g = SegmentationNetwork._get_gaussian(patch)
agg = np.zeros((3,) + data.shape[1:], np.float32); cnt = np.zeros_like(agg)
evaluated = []

def predict_tile(ti):                      # stands in for the GPU forward + mirror averaging of predict_3D
    sx, sy, sz = tiles[ti]
    evaluated.append(ti)
    return net_fn(torch.from_numpy(np.ascontiguousarray(data[None, :, sx:sx + 16, sy:sy + 32, sz:sz + 32])))[0]

def accumulate(ti, patch_t):               # stands in for e2e_sw_accumulate
    sx, sy, sz = tiles[ti]
    agg[:, sx:sx + 16, sy:sy + 32, sz:sz + 32] += patch_t.numpy() * g
    cnt[:, sx:sx + 16, sy:sy + 32, sz:sz + 32] += g

# the product's sharded tile loop (the function predict_3D itself runs when tile_world > 1): pipelined (groups of `world`
# tiles, asynchronous all-gather under the next group's compute) and the blocking form it replaced
sl = tuple([slice(0, 3)] + slicer[1:])
for pipelined in (True, False):
    agg[:] = 0; cnt[:] = 0; del evaluated[:]
    st = {}
    parallel.run_tiles_sharded(len(tiles), rank, world, None, predict_tile, accumulate, (3,) + patch, torch.device("cpu"),
                               pipelined=pipelined, stats=st)
    assert evaluated == list(range(rank, len(tiles), world))
    assert st["tiles_total"] == len(tiles) and st["tiles_local"] == len(evaluated) and st["pipelined"] == pipelined
    probs = agg[sl] / cnt[sl]
    assert np.array_equal(probs, probs_ref), "sharded overlap-add must be bit identical to the single-process order"
    assert np.array_equal(probs.argmax(0), seg_ref)

# the other exchange (SURVEY section 8e): every rank overlap-adds its own tiles into a partial volume, ONE all-reduce at the end;
# the weight map is accumulated for all tiles on every rank (count_only).  Summation order differs: <= 1e-6, not bit-identical
def count_only(ti):                        # stands in for e2e_sw_accumulate(patch = NULL)
    sx, sy, sz = tiles[ti]
    cnt[:, sx:sx + 16, sy:sy + 32, sz:sz + 32] += g
agg[:] = 0; cnt[:] = 0; del evaluated[:]
agg_t = torch.from_numpy(agg)              # (shares memory: the all-reduce lands in agg)
st = {}
parallel.run_tiles_partial(len(tiles), rank, world, None, predict_tile, accumulate, count_only, agg_t, stats=st)
assert evaluated == list(range(rank, len(tiles), world)) and st["mode"] == "allreduce_partial_volumes"
assert st["allreduce_bytes"] == agg.size * 4
probs = agg[sl] / cnt[sl]
assert np.abs(probs - probs_ref).max() <= 1e-6, np.abs(probs - probs_ref).max()
assert (probs.argmax(0) != seg_ref).mean() <= 1e-4
cnt_partial = cnt.copy()
agg[:] = 0; cnt[:] = 0
for ti in range(len(tiles)):
    count_only(ti)
assert np.array_equal(cnt, cnt_partial), "the weight map does not depend on who evaluated a tile"

# data-parallel gradient averaging + DSFF mask broadcast
grads = {"a": torch.full((5, 3), float(rank + 1)), "b": torch.arange(4, dtype=torch.float32) * (rank + 1)}
parallel.allreduce_mean_gradients(grads, ["a", "b"])
assert torch.allclose(grads["a"], torch.full((5, 3), (1 + world) / 2.0))
assert torch.allclose(grads["b"], torch.arange(4, dtype=torch.float32) * (1 + world) / 2.0)
flat = torch.arange(70, dtype=torch.float32) * (rank + 1)          # the engine's flat gradient buffer: in place
views = [flat[0:6].view(2, 3), flat[64:70]]
parallel.allreduce_mean_flat(flat)

class _Eng:                                                         # bucketed, overlapped variant (engine hook protocol)
    pass
eng = _Eng(); eng.grad_flat = torch.arange(70, dtype=torch.float32) * (rank + 1); eng.grad_bucket_hook = None
ov = parallel.OverlappedGradAllReduce(eng)
eng.grad_bucket_hook(0, 64); eng.grad_bucket_hook(64, 70)
ov.finish()
assert torch.allclose(eng.grad_flat, torch.arange(70, dtype=torch.float32) * (1 + world) / 2.0)
assert torch.allclose(flat, torch.arange(70, dtype=torch.float32) * (1 + world) / 2.0)
assert torch.allclose(views[1], torch.arange(64, 70, dtype=torch.float32) * (1 + world) / 2.0)
# DSFF masks: every rank adopts rank 0's kernel maps after a prune/grow (the product's Masking.sync_kernel_maps)
from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking
m = Masking.__new__(Masking)
m.names = ["a", "b"]
m._params = {"a": torch.zeros(3, 4, 1, 3, 3), "b": torch.zeros(2, 5, 2, 2, 2)}
maps = {"a": (np.arange(12).reshape(3, 4) %% (rank + 2) == 0).astype(np.uint8),
        "b": (np.arange(10).reshape(2, 5) %% (rank + 3) == 0).astype(np.uint8)}
got = m.sync_kernel_maps(maps)
assert np.array_equal(got["a"], (np.arange(12).reshape(3, 4) %% 2 == 0).astype(np.uint8))
assert np.array_equal(got["b"], (np.arange(10).reshape(2, 5) %% 3 == 0).astype(np.uint8))
# data-parallel batch dice: the folded (tp, fp, fn) sums of all ranks
t = torch.arange(9, dtype=torch.float64) * (rank + 1)
parallel.batch_dice_allreduce(None)(t)
assert torch.equal(t, torch.arange(9, dtype=torch.float64) * sum(range(1, world + 1)))
# collective save_checkpoint: rank 0 writes, a failed write raises on EVERY rank instead of leaving the others in a barrier
from e2enet_medical_amd.training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
tr = nnUNetTrainer_simple.__new__(nnUNetTrainer_simple)
tr.network = torch.nn.Linear(3, 2); tr.optimizer = torch.optim.SGD(tr.network.parameters(), 0.1)
tr.epoch = 4; tr.process_group = None; tr._mask = None; tr.init_args = (); tr.plans = {"k": 1}
tr.all_tr_losses = tr.all_val_losses = tr.all_val_losses_tr_mode = tr.all_val_eval_metrics = []
tr.best_epoch_based_on_MA_tr_loss = tr.best_MA_tr_loss_for_patience = tr.best_val_eval_criterion_MA = None
ck = os.path.join(%(tmp)r, "ck.model")
tr.save_checkpoint(ck)
assert os.path.isfile(ck) and os.path.isfile(ck + ".pkl")          # complete on every rank when the call returns
assert torch.load(ck, weights_only=False)["epoch"] == 5
try:
    tr.save_checkpoint(os.path.join(%(tmp)r, "no_such_dir", "ck.model"))
    raise SystemExit("a failed checkpoint write must raise on rank %%d" %% rank)
except (RuntimeError, OSError):
    pass
if rank == 1:                                                       # the `if rank == r: save` idiom: no other rank involved
    tr.save_checkpoint(os.path.join(%(tmp)r, "solo.model"), collective=False)
dist.barrier()
assert os.path.isfile(os.path.join(%(tmp)r, "solo.model"))
if rank == 0:
    print("WORKER_OK")
dist.destroy_process_group()
'''


def test_world_size_2_tile_sharding_and_dp_exchange_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT, "tmp": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29553", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    assert "WORKER_OK" in outs[0][0]


_WORKER_SUBGROUP = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
sub = dist.new_group([1, 2])               # data-parallel replicas on global ranks 1 and 2: group rank 0 is GLOBAL rank 1
if rank in (1, 2):
    from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking
    m = Masking.__new__(Masking)
    m.names = ["a"]
    m._params = {"a": torch.zeros(3, 4, 1, 3, 3)}
    m.process_group = sub
    maps = {"a": (np.arange(12).reshape(3, 4) %% (rank + 1) == 0).astype(np.uint8)}
    got = m.sync_kernel_maps(maps)         # src = 0 is a rank OF THE GROUP (dist.broadcast wants the global one)
    assert np.array_equal(got["a"], (np.arange(12).reshape(3, 4) %% 2 == 0).astype(np.uint8)), got["a"]
dist.barrier()
if rank == 0:
    print("WORKER_OK")
dist.destroy_process_group()
'''


def test_mask_broadcast_in_a_subgroup_uses_the_global_rank_gloo(tmp_path):
    """Masking.sync_kernel_maps(src=0) inside a process group that does not contain global rank 0 (three gloo ranks, group =
    ranks 1 and 2): group rank 0 is global rank 1; passing the group-relative 0 to dist.broadcast raises / hangs."""
    script = tmp_path / "worker_sub.py"
    script.write_text(_WORKER_SUBGROUP % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29557", WORLD_SIZE="3", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(3)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    assert "WORKER_OK" in outs[0][0]


# kernel-shape ablation networks (reference unetpp_d_313.py / unetpp_d_331.py)
VARIANT = dict(patch=(16, 16, 64), cin=2, base=8, k=3, pools=[[2, 2, 2], [2, 2, 2], [1, 2, 2], [2, 1, 2], [1, 1, 2]], max_feat=32)


def _build_variant(var, seed=None):
    import importlib
    mod = importlib.import_module("e2enet_medical_amd.network_architecture.unetpp_d_" + var)
    if seed is not None:
        torch.manual_seed(seed)
    V = VARIANT
    return mod.Generic_UNetPlusPlus(V["patch"], V["cin"], V["base"], V["k"], len(V["pools"]), 2, 2, nn.Conv3d, nn.InstanceNorm3d,
                                    {'eps': 1e-5, 'affine': True}, nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU,
                                    {'negative_slope': 1e-2, 'inplace': True}, True, False, lambda x: x,
                                    mod.InitWeights_He(1e-2), V["pools"], None, False, True, True, max_num_features=V["max_feat"])


@pytest.mark.parametrize("var", ["313", "331"])
def test_conv_variant_modules_state_dict_and_init_match_reference(var):
    """The ablation networks keep the reference's module path, class name, state_dict names AND shapes (conv weights
    [o,i,3,1,3] / [o,i,3,3,1], transposed-conv weights in the reference's axis order) and its He-init draws, although
    the engine runs them on axis-permuted tensors; position-weighted checksums catch a permuted tensor."""
    g = golden("net_variants.npz")
    net = _build_variant(var, seed=1234)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(s) for s in g[var + "_init_names"]]
    shapes = {str(n): str(s) for n, s in zip(g[var + "_names"], g[var + "_shapes"])}
    for n, v in sd.items():
        assert str(tuple(v.shape)) == shapes[n], n
    assert np.array_equal(np.array([v.double().sum().item() for v in sd.values()]), g[var + "_init_sum"])
    assert np.array_equal(np.array([v.double().abs().sum().item() for v in sd.values()]), g[var + "_init_abs"])
    pos = np.array([(v.double().flatten() * (torch.arange(v.numel(), dtype=torch.float64) + 1)).sum().item() / v.numel()
                    for v in sd.values()])
    assert np.array_equal(pos, g[var + "_init_pos"])
    # round trip through the checkpoint format leaves the engine-order parameters untouched
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    net.load_state_dict({k: v.clone() for k, v in sd.items()})
    for n, p in net.named_parameters():
        assert torch.equal(p, before[n]), n
    # the engine's pooling plan is the reference's with the axes permuted; divisibility stays in the reference's order
    perm = {"313": (1, 0, 2), "331": (2, 0, 1)}[var]
    assert net._cfg.pool_kernels == [tuple(k[a] for a in perm) for k in VARIANT["pools"]]
    assert list(net.input_shape_must_be_divisible_by) == [8, 8, 32]
    assert net._cfg.shift_size == 1
    x = torch.arange(2 * 3 * 4 * 5 * 6, dtype=torch.float32).view(2, 3, 4, 5, 6)
    assert torch.equal(net.from_engine_layout(net.to_engine_layout(x)), x)
    assert tuple(net.to_engine_layout(x).shape[2:]) == tuple(x.shape[2 + a] for a in perm)


def test_ds_target_index_formula_is_scipy_zoom_order0():
    """The per-axis source indices the device gather uses are exactly scipy.ndimage.zoom(order=0, mode='nearest',
    grid_mode=True) -- the routine scikit-image 0.19.3's resize(order=0) delegates to (oracle/ds_targets.py header: parity
    for this step is anchored on that third-party routine, the reference's own resize could not be run here)."""
    from scipy import ndimage
    from e2enet_medical_amd.training.data_augmentation.downsampling import zoom_nearest_indices
    import oracle
    for n_in in (1, 2, 3, 5, 7, 8, 20, 33, 40, 56, 96, 128, 160):
        for n_out in sorted({max(1, int(np.round(n_in * s))) for s in (1.0, 0.5, 0.25, 0.125, 0.0625, 0.75)}):
            ramp = np.arange(n_in, dtype=float)
            ref = ndimage.zoom(ramp, n_out / n_in, order=0, mode="nearest", grid_mode=True)
            assert ref.shape == (n_out,)
            assert np.array_equal(zoom_nearest_indices(n_in, n_out), ref.astype(np.int32)), (n_in, n_out)
    assert np.array_equal(zoom_nearest_indices(128, 64), np.arange(64) * 2 + 1)          # scale 1/2 keeps the odd voxels
    assert np.array_equal(zoom_nearest_indices(128, 32), np.arange(32) * 4 + 2)
    seg = np.random.RandomState(0).randint(0, 4, size=(2, 1, 6, 10, 12)).astype(np.float32)
    outs = oracle.downsample_seg_for_ds(seg, [[1, 1, 1], [0.5, 0.5, 0.5], [1, 0.5, 0.25]])
    assert outs[0] is seg and outs[1].shape == (2, 1, 3, 5, 6) and outs[2].shape == (2, 1, 6, 5, 3)
    assert np.array_equal(outs[1], seg[:, :, 1::2, 1::2, 1::2])


def test_oracle_resample_softmax_properties():
    """oracle.export.resample_softmax (scipy restatement of resample_data_or_seg, preprocessing.py:113-202; parity unpinned):
    identity on an unchanged grid, exact on constants and on linear ramps away from the border, nearest pick along a separate
    axis, and the separate-axis branch == slice-wise resize followed by the nearest pick."""
    import numpy as np
    from oracle.export import resample_softmax
    rng = np.random.RandomState(0)
    x = rng.rand(2, 6, 10, 12).astype(np.float32)
    assert resample_softmax(x, (6, 10, 12)) is x
    c = np.full((1, 5, 7, 9), 0.25, np.float32)
    assert np.array_equal(resample_softmax(c, (8, 14, 5)), np.full((1, 8, 14, 5), 0.25, np.float32))
    ramp = np.broadcast_to(np.arange(12, dtype=np.float32), (1, 6, 10, 12)).copy()
    up = resample_softmax(ramp, (6, 10, 24))
    want = np.clip((np.arange(24) + 0.5) * 0.5 - 0.5, 0, 11)
    assert np.abs(up[0, 3, 4] - want).max() <= 1e-6
    sep = resample_softmax(x, (3, 20, 18), lowres_axis=0)
    assert sep.shape == (2, 3, 20, 18)
    idx = np.floor(np.clip(6 / 3 * (np.arange(3) + 0.5) - 0.5, 0, 5) + 0.5).astype(int)
    full = resample_softmax(x, (6, 20, 18), lowres_axis=0)
    assert np.array_equal(sep, full[:, idx])


@pytest.mark.parametrize("graph,divs", [("unetpp", (64,)), ("unetpp", (8, 512)), ("unet", (8, 64))])
def test_lane_plan_orders_every_cross_lane_dependency(graph, divs, monkeypatch):
    """Engine._plan_lanes / _plan_lanes_backward (plan only, no kernel is launched: the plan builds on the CPU device): replay the
    issue order with vector clocks — a lane knows what was issued on it earlier and whatever an awaited event's op knew.
    Forward: every source tensor's producer must be known when its consumer is issued.  Backward: every earlier toucher of a
    gradient buffer an op touches must be known (first writer overwrites, later writers accumulate in the planned order,
    the producer's in-place dz -> dy pass comes last).  Events are only awaited after they were recorded."""
    import oracle
    from e2enet_medical_amd import engine as E
    monkeypatch.setattr(E, "LANE_DIVS", divs)
    pools = [(2, 2, 2)] * 5
    cfg = E.NetConfig(4, 8, 3, pools, 2, 64, graph=graph)
    spec = oracle.make_spec(4, 8, 3, pools, 2, 64, graph=graph) if graph != "unetpp" else oracle.make_spec(4, 8, 3, pools, 2, 64)
    eng = E.Engine(cfg, oracle.init_params(spec, 0), 2, (64, 64, 64), torch.device("cpu"))
    eng.prepare_backward()
    lanes = eng._lane_of
    assert set(lanes) == set(range(len(divs) + 1)), "every lane is used at this shape"

    def replay(order, deps, touched_of):
        known = [set() for _ in range(len(divs) + 1)]        # per lane: ops known to be complete when the lane reaches this point
        snapshot, recorded, last_toucher = {}, set(), {}
        for i in order:
            ln = lanes[i]
            for j in deps[i]:
                assert j in recorded, "op %d waits for the event of op %d before it is recorded" % (i, j)
                known[ln] |= snapshot[j] | {j}
            for res, prev in touched_of(i, last_toucher):
                for j in prev:
                    assert j in known[ln], "op %d (lane %d) is not ordered behind op %d (lane %d) on %s" % (i, ln, j, lanes[j], res)
            known[ln].add(i)                                 # in-order stream: later ops of the lane see this one complete
            snapshot[i] = set(known[ln])
            recorded.add(i)

    producer = {id(op.out): i for i, op in enumerate(eng.ops)}

    def fwd_touch(i, _last):
        return [(a.name, [producer[id(a)]] if id(a) in producer else []) for a in E.Engine._reads(eng.ops[i])]
    replay(range(len(eng.ops)), eng._deps_fwd, fwd_touch)

    def bwd_touch(i, last):
        op = eng.ops[i]
        out = []
        for a in [op.out] + [s for s in E.Engine._reads(op) if s.grad is not None]:
            out.append((a.name, list(last.get(id(a), []))))
            last.setdefault(id(a), []).append(i)
        return out
    # the backward issue order: the reverse op list with every pooling backward moved behind the other writers of its source's
    # gradient buffer, right in front of the source's producer (Engine._backward_order: it forms that producer's InstanceNorm sums)
    order = list(eng._bwd_order)
    assert sorted(order) == list(range(len(eng.ops)))
    pos = {i: k for k, i in enumerate(order)}
    for i, op in enumerate(eng.ops):
        if isinstance(op, E.PoolOp) and op.src.producer is not None:
            assert pos[i] + 1 == pos[producer[id(op.src)]], "pooling backward sits right in front of its source's producer"
            assert op.src.last_writer is op
        else:
            # everything else keeps the reverse order among itself
            assert all(pos[i] < pos[j] for j in range(i) if not isinstance(eng.ops[j], E.PoolOp)) or isinstance(op, E.PoolOp)
    replay(order, eng._deps_bwd, bwd_touch)
    assert any(eng._deps_fwd) and any(eng._deps_bwd)


def _bench(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_self_launches_n_ranks_without_torchrun():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the form the driver uses): the parent starts two
    children, one rank each, and relays exactly one JSON line from rank 0 (dry mode: gloo, no GPU work)."""
    import json
    r = _bench({"E2E_BENCH_DRY": "1"}, "--gpus", "2", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["warmup"] == 1
    assert rec["rccl"]["world"] == 2 and rec["rccl"]["allreduce_of_ones"] == 2.0
    assert {"ms_per_step_without_allreduce", "allreduce_exposed_ms", "allreduce_bytes_per_step", "device_per_rank"} <= set(rec["rccl"])   # the N > 1 record
    assert rec["ms_per_step_per_rank"] == [1.0, 2.0]                  # rank r contributed r + 1: both ranks were in the group


def test_bench_launcher_propagates_a_failed_rank_and_refuses_a_mismatched_world():
    r = _bench({"E2E_BENCH_DRY": "1", "E2E_BENCH_DRY_FAIL_RANK": "1"}, "--gpus", "2")
    assert r.returncode == 3 and "rank 1 exited with code 3" in r.stderr and not r.stdout.strip()
    r = _bench({"E2E_BENCH_DRY": "1", "WORLD_SIZE": "1"}, "--gpus", "2")       # torchrun form with the wrong --gpus
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr


@pytest.mark.parametrize("tag,mode,kw", [("edge", "edge", None), ("const", "constant", {'constant_values': 0})])
def test_dataloader3d_batches_equal_the_reference(tmp_path, tag, mode, kw):
    """DataLoader3D (reference dataset_loading.py:163-388) -- same constructor, same numpy.random draw order: under
    one seed the batches (random crops, a third of each batch forced onto a foreground voxel, 'edge' / constant borders, a case
    smaller than the patch) are the reference's, bit for bit, and so is the position of the random stream afterwards."""
    from tests.helpers import synthetic_cases
    from e2enet_medical_amd.training.dataloading.dataset_loading import DataLoader3D
    g = golden("dataloader.npz")
    ds = synthetic_cases(str(tmp_path))
    np.random.seed(1234)
    loader = DataLoader3D(ds, (16, 18, 20), (12, 14, 16), 4, False, oversample_foreground_percent=0.33, pad_mode=mode,
                          pad_kwargs_data=kw)
    assert loader.data_shape == (4, 2, 16, 18, 20) and loader.seg_shape == (4, 1, 16, 18, 20)
    assert [loader.get_do_oversample(j) for j in range(4)] == [False, False, False, True]
    seen = []
    for it in range(3):
        b = next(loader)
        assert [str(k) for k in b['keys']] == [str(k) for k in g["%s_keys%d" % (tag, it)]]
        assert np.array_equal(b['data'], g["%s_data%d" % (tag, it)]), "data of batch %d" % it
        assert np.array_equal(b['seg'], g["%s_seg%d" % (tag, it)].astype(np.float32)), "seg of batch %d" % it
        assert b['data'].dtype == np.float32 and b['seg'].dtype == np.float32
        assert b['data_pinned'].numpy().ctypes.data == b['data'].ctypes.data          # the batch IS the (pinned) staging buffer
        seen.append(b['data'].ctypes.data)
        assert [p['name'] for p in b['properties']] == [str(k) for k in b['keys']]
    assert seen[0] != seen[1] and seen[0] == seen[2]                                    # two rotating buffers
    assert np.array_equal(np.random.randint(0, 2 ** 31 - 1, 4), g[tag + "_rng_after"])


@pytest.mark.parametrize("R,Cc,dens", [(32, 64, 0.2), (64, 160, 0.2), (34, 33, 0.3), (128, 320, 0.1), (70, 90, 0.5)])
def test_sparse_plan_is_a_valid_balanced_permutation(R, Cc, dens):
    """e2e_conv133_sparse_plan (host side of the load-balanced DSFF conv, csrc/conv133_sparse.hip): for both directions the plan
    places every output plane in exactly one wave slot of its 32-plane group and every input plane in exactly one chunk slot (the
    same chunking for every group), its liveness words are the kernel map read through those slots (mask indices stay bit exact), it is a pure
    function of the map, and the work of the slowest wave summed over the chunks is close to the mean (the point of the plan)."""
    import ctypes as C
    from e2enet_medical_amd._lib import lib
    L = lib()
    km = (np.random.RandomState(R * 1000 + Cc).rand(R, Cc) < dens).astype(np.uint8)

    def plan(tr):
        Q, P = (Cc, R) if tr else (R, Cc)
        G, NC = (Q + 31) // 32, (P + 7) // 8
        qs, ps, qd = np.empty(G * 32, np.int32), np.empty(G * NC * 8, np.int32), np.empty(G * 8 * NC, np.uint32)
        wo = np.empty(G * NC * 8, np.int32)
        fl, kmx = C.c_int(0), C.c_int(0)
        L.conv133_sparse_plan(km.ctypes.data, R, Cc, tr, qs.ctypes.data, ps.ctypes.data, qd.ctypes.data, wo.ctypes.data,
                              C.addressof(kmx), C.addressof(fl))
        # the packed block of a chunk holds the live kernels wave by wave: offsets are the running popcount, kmax the largest chunk
        pc = np.array([bin(int(v)).count("1") for v in qd]).reshape(G, 8, NC)
        want = np.cumsum(pc, axis=1) - pc                            # [G, wave, chunk] exclusive prefix over the waves
        assert np.array_equal(wo.reshape(G, NC, 8), want.transpose(0, 2, 1))
        assert kmx.value == max(1, int(pc.sum(axis=1).max()))
        assert L.conv133_sparse_wpk_floats(P, Q, kmx.value) == G * NC * kmx.value * 12
        return Q, P, G, NC, qs, ps, qd, fl.value
    for tr in (0, 1):
        Q, P, G, NC, qs, ps, qd, flush = plan(tr)
        again = plan(tr)
        assert all(np.array_equal(a, b) for a, b in zip((qs, ps, qd), again[4:7])) and flush == again[7]
        assert 1 <= flush <= NC
        alive = km.T if tr else km                                   # [Q, P]
        natural = planned = ideal = 0.0
        for g in range(G):
            q = qs[g * 32:(g + 1) * 32]
            assert sorted(q[q >= 0]) == list(range(g * 32, min(Q, g * 32 + 32)))
            p = ps[g * NC * 8:(g + 1) * NC * 8]
            assert sorted(p[p >= 0]) == list(range(P))
            sub = np.zeros((32, NC * 8), np.uint8)                   # the map in slot order
            sub[np.ix_(q >= 0, p >= 0)] = alive[np.ix_(q[q >= 0], p[p >= 0])]
            words = np.zeros((8, NC), np.uint32)
            for w in range(8):
                for c in range(NC):
                    bits = sub[w * 4:w * 4 + 4, c * 8:c * 8 + 8]          # [a, cl] -> bit cl * 4 + a
                    words[w, c] = sum(int(bits[a, cl]) << (cl * 4 + a) for a in range(4) for cl in range(8))
            assert np.array_equal(words.reshape(-1), qd[g * 8 * NC:(g + 1) * 8 * NC])

            def cost(m):                                             # m [32, NC * 8] -> per (chunk, wave): kernels + 0.35 per visited plane
                k = m.reshape(8, 4, NC, 8)
                return (k.sum(axis=(1, 3)) + 0.35 * (k.sum(axis=1) > 0).sum(axis=2)).T
            nat = np.zeros((32, NC * 8), np.uint8)
            nq = min(32, Q - g * 32)
            nat[:nq, :P] = alive[g * 32:g * 32 + nq]
            natural += cost(nat).max(axis=1).sum()
            planned += cost(sub).max(axis=1).sum()
            ideal += cost(sub).sum() / 8
        assert planned <= natural + 1e-6
        for g in range(1, G):                                        # ONE chunking of the input planes for all groups of a layer
            assert np.array_equal(ps[:NC * 8], ps[g * NC * 8:(g + 1) * NC * 8])           # (they share the staged planes through L2)
        if Q % 32 == 0 and dens <= 0.3:
            assert natural / ideal >= 1.3 and (planned / ideal <= 1.2 if G == 1 else planned <= 0.87 * natural), (planned / ideal, natural / ideal)


def test_trainer_never_trains_on_noise_by_accident_and_splits_like_sklearn(tmp_path):
    """initialize(training=True) builds its generators from the preprocessed stage folder (reference nnUNetTrainer_simple.py:216-239);
    without one it raises unless synthetic_data was set on purpose.  load_dataset / do_split: the reference's file-name dict and
    its 5-fold split (sklearn KFold(5, shuffle=True, random_state=12345), restated without sklearn: compared with sklearn here)."""
    import pickle
    from collections import OrderedDict
    from e2enet_medical_amd.training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
    plans = {'plans_per_stage': {0: {'batch_size': 1, 'patch_size': [16, 32, 32], 'num_pool_per_axis': [3, 5, 5],
                                     'pool_op_kernel_sizes': [[2, 2, 2]] * 3 + [[1, 2, 2]] * 2,
                                     'conv_kernel_sizes': [[3, 3, 3]] * 6, 'do_dummy_2D_data_aug': False}},
             'base_num_features': 32, 'num_modalities': 1, 'num_classes': 2, 'all_classes': [1, 2], 'conv_per_stage': 2,
             'data_identifier': 'nnUNetData_plans_v2.1'}
    tr = nnUNetTrainer_simple(plans, 0, dataset_directory=str(tmp_path), batch_dice=False, Tconv='shiftConvPP')
    tr.base_num_features_override = 8
    if not torch.cuda.is_available():
        with pytest.raises((FileNotFoundError, RuntimeError)):          # (without a GPU the engine itself refuses first)
            tr.initialize(True)
    # ---- dataset dict and split (host logic only)
    folder = tmp_path / "nnUNetData_plans_v2.1_stage0"
    folder.mkdir()
    names = ["case_%02d" % i for i in (7, 3, 11, 0, 5, 9, 1, 8, 2, 10, 4, 6, 12)]
    for n in names:
        np.save(str(folder / (n + ".npy")), np.zeros((2, 4, 4, 4), np.float32))
        with open(str(folder / (n + ".pkl")), "wb") as f:
            pickle.dump(OrderedDict(class_locations={1: np.zeros((0, 3), int)}, name=n), f)
    tr.folder_with_preprocessed_data = str(folder)
    tr.load_dataset()
    assert list(tr.dataset.keys()) == sorted(names)
    assert tr.dataset["case_03"]["data_file"].endswith("case_03.npz") and tr.dataset["case_03"]["properties"]["name"] == "case_03"
    from sklearn.model_selection import KFold
    keys = np.sort(names)
    want = [(keys[a], keys[b]) for a, b in KFold(n_splits=5, shuffle=True, random_state=12345).split(keys)]
    for fold in range(5):
        tr.fold = fold
        tr.do_split()
        assert list(tr.dataset_tr.keys()) == sorted(want[fold][0]) and list(tr.dataset_val.keys()) == sorted(want[fold][1])
    assert (tmp_path / "splits_final.pkl").exists()
    tr.fold = 7                                                           # not in the file: seeded 80:20 split
    tr.do_split()
    assert len(tr.dataset_tr) == int(len(names) * 0.8) and len(tr.dataset_val) == len(names) - int(len(names) * 0.8)
    tr.fold = "all"
    tr.do_split()
    assert list(tr.dataset_tr.keys()) == list(tr.dataset_val.keys()) == sorted(names)
    # ---- the loader's patch of a dummy_2D plan keeps the depth and grows in-plane only (reference :718-727)
    plans2 = dict(plans)
    plans2['plans_per_stage'] = {0: dict(plans['plans_per_stage'][0], patch_size=[16, 64, 64], do_dummy_2D_data_aug=True)}
    tr2 = nnUNetTrainer_simple(plans2, 0, batch_dice=False, Tconv='shiftConvPP')
    tr2.load_plans_file()
    tr2.process_plans(tr2.plans)
    tr2.setup_DA_params()
    assert tr2.data_aug_params["dummy_2D"] and tr2.data_aug_params["rotation_x"][1] == pytest.approx(np.pi)
    assert tr2.basic_generator_patch_size[0] == 16 and tr2.basic_generator_patch_size[1] > 64 and tr2.basic_generator_patch_size[2] > 64
    tr.process_plans(tr.plans)
    tr.setup_DA_params()
    assert not tr.data_aug_params["dummy_2D"] and all(b >= p for b, p in zip(tr.basic_generator_patch_size, tr.patch_size))


def test_inference_nonlin_is_recognised_by_what_it_computes():
    """SegmentationNetwork._inference_nonlin_code: identity (the constructor default), softmax over the class axis (softmax_helper
    or a caller's own lambda) and sigmoid map to the modes of e2e_nonlin_flip_acc; anything else raises instead of being replaced by
    softmax (reference neural_network.py:80, :531-560; nnUNetTrainer_simple.py:363)."""
    import torch
    import torch.nn.functional as F
    from e2enet_medical_amd.network_architecture.neural_network import SegmentationNetwork
    from e2enet_medical_amd.utilities.nd_softmax import softmax_helper
    for k in (2, 3, 16):
        net = SegmentationNetwork()
        net.num_classes = k
        assert net._inference_nonlin_code() == 0
        for fn, code in ((softmax_helper, 1), (lambda t: F.softmax(t, 1), 1), (lambda t: torch.softmax(t, dim=1), 1),
                         (torch.sigmoid, 2), (lambda t: t.clone(), 0)):
            net.inference_apply_nonlin = fn
            assert net._inference_nonlin_code() == code
        for bad in (torch.tanh, lambda t: F.softmax(t, 2), lambda t: t * 2, lambda t: t[:, :1]):
            net.inference_apply_nonlin = bad
            with pytest.raises(NotImplementedError, match="inference_apply_nonlin"):
                net._inference_nonlin_code()


def test_evaluator_matches_reference_aggregate_scores(tmp_path):
    """e2enet_medical_amd.evaluation.evaluator (what nnUNetTrainer_simple.validate writes summary.json with) against the
    reference's aggregate_scores / Evaluator on the same label maps (tests/golden/evaluator.npz): metric names and order,
    per-case values incl. the NaN rules for absent / full labels, nan-means, summary.json keys."""
    import json
    from e2enet_medical_amd.evaluation.evaluator import aggregate_scores, DEFAULT_METRICS
    g = golden("evaluator.npz")
    cases = [(g["test%d" % c], g["ref%d" % c], "t%d" % c, "r%d" % c) for c in range(3)]
    jf = tmp_path / "summary.json"
    scores = aggregate_scores(cases, [0, 1, 2, 3], json_output_file=str(jf), json_name="n", json_task="t")
    names = [str(m) for m in g["metric_names"]]
    assert sorted(DEFAULT_METRICS) == names
    for c in range(3):
        assert scores["all"][c]["test"] == "t%d" % c and scores["all"][c]["reference"] == "r%d" % c
        for l in range(4):
            assert list(scores["all"][c][str(l)].keys()) == names
            got = np.array([float(scores["all"][c][str(l)][m]) for m in names])
            np.testing.assert_array_equal(got, g["all"][c, l])            # (NaN == NaN under assert_array_equal)
    for l in range(4):
        got = np.array([float(scores["mean"][str(l)][m]) for m in names])
        np.testing.assert_array_equal(got, g["mean"][l])
    js = json.load(open(jf))
    assert sorted(js.keys()) == [str(k) for k in g["summary_keys"]]
    assert js["name"] == "n" and js["task"] == "t" and len(js["id"]) == 12 and len(js["results"]["all"]) == 3


def test_live_traffic_summary_and_bench_selection(tmp_path, monkeypatch):
    """bench.py --live-traffic: the per-kernel summary of two rocprofv3 --pmc passes (FETCH_SIZE in KiB, doubled on gfx950; WRITE_SIZE)
    and its use for `roofline.traffic`: a summary collected by the run itself takes precedence over the committed one, a committed
    summary of another ABI version is not reported, a failed counter pass leaves the committed summary in charge."""
    import importlib.util
    import bench
    from e2enet_medical_amd._lib import ABI_VERSION
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("traffic_summary", os.path.join(root, "tools", "traffic_summary.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    kname = "void (anonymous namespace)::conv133_mm_kernel<0, 2>((anonymous namespace)::MmParams)"
    for sub, ctr, vals in (("pmc_fetch", "FETCH_SIZE", [1000.0, 3000.0]), ("pmc_write", "WRITE_SIZE", [500.0, 500.0])):
        d = tmp_path / sub / "host"
        d.mkdir(parents=True)
        with open(d / "1_counter_collection.csv", "w") as fh:
            fh.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n")
            for i, v in enumerate(vals):
                fh.write('%d,"%s",%s,%s\n' % (i, kname, ctr, v))
            fh.write('9,"%s",GRBM_GUI_ACTIVE,7\n' % kname)
    doc = ts.summarise(str(tmp_path))
    rec = doc["conv133_mm_kernel<0, 2>"]
    assert rec["launches"] == 2
    assert rec["fetch_bytes_per_launch_corrected"] == 2.0 * 2000.0 * 1024 and rec["write_bytes_per_launch"] == 500.0 * 1024
    assert rec["hbm_bytes_per_launch"] == (4000.0 + 500.0) * 1024 and doc["_meta"] == {"abi_version": ABI_VERSION}
    monkeypatch.setattr(bench, "LIVE_TRAFFIC", doc)
    assert bench.pmc_traffic("conv133_mm_kernel", "conv133_kernel") == (4000.0 + 500.0) * 1024
    assert bench.traffic_source().startswith("live:")
    stale = dict(doc, _meta={"abi_version": ABI_VERSION - 1})
    monkeypatch.setattr(bench, "LIVE_TRAFFIC", stale)
    assert bench.pmc_traffic("conv133_mm_kernel") is None          # counters of another library revision are never reported
    # a counter pass that fails (no GPU here): nothing is reported as live, the committed summary stays in charge
    monkeypatch.setattr(bench, "LIVE_TRAFFIC", None)

    class R:
        returncode, stderr = 1, "no device"
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: R())
    bench.collect_live_traffic()
    assert bench.LIVE_TRAFFIC is None and bench.traffic_source().startswith("profiles/")
