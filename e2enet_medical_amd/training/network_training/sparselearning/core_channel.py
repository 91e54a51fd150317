"""Dynamic Sparse Feature Fusion masks (``Masking``) with the statistics on the GPU.

Drop-in for reference e2enet/training/network_training/sparselearning/core_channel.py (``add_sparse_args`` :17-31,
``CosineDecay`` :32-41, ``Masking`` :59-).  Behaviour kept bit for bit where the reference defines it:
  * tensor selection by name (:324), uniform kernel-granular init with Python ``random.sample`` (:141-169, incl.
    the ``shape[0] == 48`` density quirk), ``apply_mask`` on weights and momentum (:427-434),
  * ``step`` = apply_mask -> cosine death-rate decay -> every ``update_frequency`` steps ``truncate_weights``
    (:290-317, :556-611): kernel-L1 magnitude death (:647-666) then random growth (:721-739).
Mechanism: the per-kernel L1 sums (same association order as the three chained ``torch.sum``), the exact k-th order
statistic (radix select instead of a full sort) and the death comparison run as HIP kernels; the index draws stay
on the host with Python's ``random`` so that the mask indices are bit-identical to the reference for the same seed.
death_mode='magnitude' (the CLI default, :25) with growth_mode='random' (the CLI default, :24; ``kernel_growth`` :721-739) or
growth_mode='gradient' (``kernel_grad_growth`` :771-790: the dead kernels with the largest |gradient| sums regrow -- the dense
weight gradient the clip norm needs anyway is in HBM every step) are implemented.
"""
from __future__ import print_function

import math
import random

import numpy as np
import torch
import torch.optim as optim

from ...._lib import lib, ParamEntry


def str2bool(s):
    return True if s.lower() == 'true' else False


def add_sparse_args(parser):
    parser.add_argument('--sparse', type=str2bool, default=True, help='Enable sparse mode. Default: True.')
    parser.add_argument('--adv', type=bool, default=False, help='adv sparse mode. Default: True.')
    parser.add_argument('--init-prune-epoch', type=int, default=0, help='The pruning rate / death rate.')
    parser.add_argument('--final-prune-epoch', type=int, default=1000, help='The density of the overall sparse network.')
    parser.add_argument('--fix', type=bool, default=False, help='Fix sparse connectivity during training. Default: True.')
    parser.add_argument('--sparse_init', type=str, default='uniform', help='sparse initialization: ERK, snip, Grasp')
    parser.add_argument('--growth', type=str, default='random', help='Growth mode. Choose from: momentum, random, random_unfired, and gradient.')
    parser.add_argument('--death', type=str, default='magnitude', help='Death mode / pruning mode. Choose from: magnitude, SET, threshold.')
    parser.add_argument('--redistribution', type=str, default='none', help='Redistribution mode. Choose from: momentum, magnitude, nonzeros, or none.')
    parser.add_argument('--death-rate', type=float, default=0.50, help='The pruning rate / death rate.')
    parser.add_argument('--density', type=float, default=0.3, help='The density of the overall sparse network.')
    parser.add_argument('--final_density', type=float, default=0.05, help='The density of the overall sparse network.')
    parser.add_argument('--update_frequency', type=int, default=5, metavar='N', help='how many iterations to train between parameter exploration')
    parser.add_argument('--decay-schedule', type=str, default='cosine', help='The decay schedule for the pruning rate. Default: cosine. Choose from: cosine, linear.')


class CosineDecay(object):
    """Death-rate schedule: torch CosineAnnealingLR on a dummy SGD (reference :32-41; the recursive form is kept
    because it differs from the closed form in the last ulp)."""

    def __init__(self, death_rate, T_max, eta_min=0.001, last_epoch=-1):
        self.sgd = optim.SGD(torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(1))]), lr=death_rate)
        self.cosine_stepper = torch.optim.lr_scheduler.CosineAnnealingLR(self.sgd, T_max, eta_min, last_epoch)

    def step(self):
        self.cosine_stepper.step()

    def get_dr(self):
        return self.sgd.param_groups[0]['lr']


class LinearDecay(object):
    def __init__(self, death_rate, factor=0.99, frequency=600):
        self.factor, self.steps, self.frequency = factor, 0, frequency

    def step(self):
        self.steps += 1

    def get_dr(self, death_rate):
        if self.steps > 0 and self.steps % self.frequency == 0:
            return death_rate * self.factor
        return death_rate


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Masking(object):
    def __init__(self, optimizer, death_rate=0.3, growth_death_ratio=1.0, death_rate_decay=None, death_mode='magnitude',
                 growth_mode='momentum', redistribution_mode='momentum', threshold=0.001, train_loader=None, T_max=0.,
                 args=None, verbose=False):
        if death_mode != 'magnitude' or growth_mode not in ('random', 'gradient'):
            raise NotImplementedError("MI355X Masking implements death_mode='magnitude' (core_channel.py:25) with growth_mode="
                                      "'random' (the CLI default, :24) or 'gradient' (:771-790); got %s/%s"
                                      % (death_mode, growth_mode))
        self.args = args
        self.device = torch.device("cuda")
        self.growth_mode, self.death_mode = growth_mode, death_mode
        self.growth_death_ratio = growth_death_ratio
        self.redistribution_mode = redistribution_mode
        self.death_rate_decay = death_rate_decay
        self.verbose = verbose
        self.masks = {}            # name -> float32 element mask (API parity with the reference)
        self.kmasks = {}           # name -> uint8 [dim0, dim1] kernel map on the device (authoritative)
        self._kmask_host = {}      # host mirror (numpy uint8), used for the index draws
        self.modules, self.names = [], []
        self.optimizer = optimizer
        self.name2zeros, self.num_remove, self.num_death, self.name2nonzeros = {}, {}, {}, {}
        self.death_rate = death_rate
        self.baseline_nonzero = None
        self.steps = 0
        self.explore_step = 0
        self.decay_flag = True
        self.total_nozeros = self.total_weights = 0
        self.loader = train_loader
        self.adv = getattr(args, 'adv', False)
        self.curr_density = 0.0
        self.T_max = T_max
        self.prune_every_k_steps = None if getattr(args, 'fix', False) else getattr(args, 'update_frequency', None)
        self._table = None
        self._table_keys = None
        # growth_mode='gradient': where the weight gradients are.  The reference reads ``weight.grad`` AFTER clip_grad_norm_ has
        # scaled it in place (nnUNetTrainer_simple.py:573, core_channel.py:833-835).  The trainer's fused step leaves the
        # gradients unscaled in the engine's flat buffer and hands them over with the squared global norm (set_gradients);
        # without that, ``parameter.grad`` is used as it is (the autograd route: clip_grad_norm_ already scaled it).
        self._grad_source = None

    def set_gradients(self, grads, sq_norm=None, max_norm=12.0):
        """grads: name -> gradient tensor of this iteration (device); sq_norm: device fp64 scalar holding the squared global
        gradient norm the clip coefficient max_norm / (sqrt(sq_norm) + 1e-6) is derived from (None: already clipped)."""
        self._grad_source = (grads, sq_norm, float(max_norm))

    # ------------------------------------------------------------------------------------------ setup
    def add_module(self, module, density, sparse_init='ER'):
        self.modules.append(module)
        self.module = module
        self._params = {}
        for name, tensor in module.named_parameters():
            if ('loc' in name and 'context' not in name) or 'up' in name:        # reference :324
                if 'bias' in name or 'instnorm' in name:                          # remove_weight_partial_name :329-331
                    continue
                self.names.append(name)
                self._params[name] = tensor
        self.init(mode=sparse_init, density=density)

    def init(self, mode='ERK', density=0.05, erk_power_scale=1.0):
        self.density = density
        if mode != 'uniform':
            raise NotImplementedError("only sparse_init='uniform' (the CLI default, core_channel.py:23) is implemented")
        self.baseline_nonzero = 0
        for name in self.names:                                   # named_parameters order == random draw order
            w = self._params[name]
            shp = tuple(w.shape)
            density_n = 0.2 if shp[0] == 48 else density          # reference quirk :147-151
            k_size = np.prod(shp[-3:])
            nonzeros = w.numel() * density_n
            kernel_num = round(nonzeros / k_size)
            idx_rand = random.sample(list(range(0, shp[0] * shp[1])), kernel_num)
            km = np.zeros(shp[0] * shp[1], dtype=np.uint8)
            km[np.asarray(idx_rand, dtype=np.int64)] = 1
            self._set_kmask(name, km.reshape(shp[0], shp[1]))
            self.baseline_nonzero += int(km.sum()) * int(k_size)
            if self.verbose:
                print("layer: %s, shape: %s, density: %f" % (name, shp, km.mean()))
        self._push_liveness()
        self.apply_mask()
        total = sum(m.numel() for m in self.masks.values())
        sparse = sum(int(self._kmask_host[n].sum()) * int(np.prod(self._params[n].shape[-3:])) for n in self.names)
        print('Total Model parameters:', total)
        print('Total parameters under sparsity level of {0}: {1}'.format(self.density, sparse / total))

    def _set_kmask(self, name, km_host: np.ndarray):
        w = self._params[name]
        dev = w.device
        self._kmask_host[name] = km_host
        km = torch.from_numpy(km_host).to(dev)
        self.kmasks[name] = km
        mask = self.masks.get(name)
        if mask is None or mask.device != dev:
            mask = torch.empty(w.shape, dtype=torch.float32, device=dev)
            self.masks[name] = mask
        r, cc = km_host.shape
        ks = int(np.prod(w.shape[-3:]))
        lib().dsff_expand(km.data_ptr(), mask.data_ptr(), None, None, r, cc, ks, _stream())
        self._table = None

    # ------------------------------------------------------------------------------------------ checkpoint
    def state_dict(self):
        """DSFF state for a resumable checkpoint: the reference saves none of it (its resume re-draws the masks, §5 of
        SURVEY.md).  Kernel maps are bit-packed ([dim0, dim1] -> ceil(dim0*dim1/8) bytes, 0.17 MB at 32 ch); the Python
        `random` state is what the growth draws continue from."""
        import random
        sd = {'version': 1, 'steps': self.steps, 'explore_step': self.explore_step, 'death_rate': self.death_rate,
              'decay_flag': self.decay_flag, 'density': getattr(self, 'density', None),
              'kmasks': {n: (tuple(self._kmask_host[n].shape), np.packbits(self._kmask_host[n].reshape(-1)))
                         for n in self.names},
              'random_state': random.getstate()}
        if isinstance(self.death_rate_decay, CosineDecay):
            sd['decay'] = {'kind': 'cosine', 'scheduler': self.death_rate_decay.cosine_stepper.state_dict(),
                           'lr': self.death_rate_decay.sgd.param_groups[0]['lr']}
        elif isinstance(self.death_rate_decay, LinearDecay):
            sd['decay'] = {'kind': 'linear', 'steps': self.death_rate_decay.steps}
        return sd

    def load_state_dict(self, sd):
        """Restore after add_module() on the re-created network/optimizer: masks, schedule position, RNG."""
        import random
        if sd.get('version') != 1:
            raise ValueError("unknown DSFF checkpoint version %r" % (sd.get('version'),))
        missing = [n for n in self.names if n not in sd['kmasks']]
        if missing or len(sd['kmasks']) != len(self.names):
            raise KeyError("DSFF checkpoint does not match the masked tensors of this network: missing %s" % missing[:3])
        for n in self.names:
            shape, packed = sd['kmasks'][n]
            w = self._params[n]
            if tuple(shape) != (w.shape[0], w.shape[1]):
                raise ValueError("kernel map of %s has shape %s, weight has %s" % (n, shape, tuple(w.shape[:2])))
            km = np.unpackbits(np.asarray(packed, dtype=np.uint8))[:shape[0] * shape[1]].reshape(shape).astype(np.uint8)
            self._set_kmask(n, km)
        self.steps, self.explore_step = sd['steps'], sd['explore_step']
        self.death_rate, self.decay_flag = sd['death_rate'], sd['decay_flag']
        dec = sd.get('decay')
        if dec is not None and dec['kind'] == 'cosine' and isinstance(self.death_rate_decay, CosineDecay):
            self.death_rate_decay.cosine_stepper.load_state_dict(dec['scheduler'])
            self.death_rate_decay.sgd.param_groups[0]['lr'] = dec['lr']
        elif dec is not None and dec['kind'] == 'linear' and isinstance(self.death_rate_decay, LinearDecay):
            self.death_rate_decay.steps = dec['steps']
        random.setstate(sd['random_state'])
        self.apply_mask()
        self._push_liveness()
        self.cal_nonzero_counts()

    def _push_liveness(self):
        for m in self.modules:
            if hasattr(m, "set_kernel_masks"):
                m.set_kernel_masks(dict(self.kmasks))

    def _notify_masks_applied(self):
        for m in self.modules:
            if hasattr(m, "masks_applied"):
                m.masks_applied()

    # ------------------------------------------------------------------------------------------ data parallel
    def _dist_world(self):
        """(world size, group) when the masks must be kept consistent over data-parallel ranks."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1, None
        group = getattr(self, "process_group", None)
        return dist.get_world_size(group), group

    def sync_kernel_maps(self, host_maps, src=0):
        """Data-parallel replicas draw the growth indices with Python's ``random`` on every rank; identical seeds give
        identical masks, but nothing in the reference guarantees identical seeds.  After a prune/grow the kernel maps of
        group rank ``src`` are therefore broadcast (one uint8 tensor, 1.39 M kernels = 1.4 MB at 32 ch; RCCL on GPUs, gloo in
        the CPU tests) and adopted by every rank.  ``host_maps``: name -> numpy uint8 [dim0, dim1]; returns the same
        dict holding rank src's maps."""
        import torch.distributed as dist
        world, group = self._dist_world()
        if world <= 1 and not getattr(self, "force_sync", False):
            return host_maps
        names = list(host_maps.keys())
        flat = np.concatenate([host_maps[n].reshape(-1) for n in names]).astype(np.uint8)
        backend = dist.get_backend(group)
        dev = self._params[self.names[0]].device if backend == "nccl" else torch.device("cpu")
        t = torch.from_numpy(flat).to(dev)
        # (dist.broadcast takes a GLOBAL rank: rank `src` OF THE GROUP is not global rank `src` for a sub-group)
        dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
        flat = t.cpu().numpy()
        off = 0
        for n in names:
            k = host_maps[n].size
            host_maps[n] = flat[off:off + k].reshape(host_maps[n].shape).copy()
            off += k
        return host_maps

    # ------------------------------------------------------------------------------------------ per-step
    def _build_table(self):
        entries, keep = [], []
        for name in self.names:
            w = self._params[name]
            st = self.optimizer.state.get(w, {}) if self.optimizer is not None else {}
            buf = st.get('momentum_buffer')
            entries.append(ParamEntry(w.data_ptr(), None, buf.data_ptr() if buf is not None else None,
                                      self.masks[name].data_ptr(), w.numel()))
            keep.append((w.data_ptr(), buf.data_ptr() if buf is not None else 0))
        raw = b"".join(bytes(e) for e in entries)
        dev = self._params[self.names[0]].device
        self._table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self._table_keys = keep

    def apply_mask(self):
        """weight *= mask; momentum_buffer *= mask (reference :427-434), one multi-tensor HIP launch."""
        if not self.names:
            return
        keys = []
        for name in self.names:
            w = self._params[name]
            st = self.optimizer.state.get(w, {}) if self.optimizer is not None else {}
            buf = st.get('momentum_buffer')
            keys.append((w.data_ptr(), buf.data_ptr() if buf is not None else 0))
        if self._table is None or keys != self._table_keys:
            self._build_table()
        lib().apply_mask(self._table.data_ptr(), len(self.names), _stream())
        from ....engine import note_native_param_write
        note_native_param_write()               # (weights written through raw pointers: see engine.PARAM_EPOCH)
        self._notify_masks_applied()

    def step(self, masks_already_applied=False):
        """reference :290-317.  ``masks_already_applied``: the fused optimizer kernel already multiplied weights and
        momentum by the masks this iteration."""
        if not masks_already_applied:
            self.apply_mask()
        else:
            self._notify_masks_applied()
        if self.decay_flag:
            self.death_rate_decay.step()
            self.death_rate = self.death_rate_decay.get_dr()
        else:
            self.death_rate = 0.001
            self.adv = False
        self.steps += 1
        if self.prune_every_k_steps is not None and self.steps % self.prune_every_k_steps == 0:
            self.explore_step += 1
            self.truncate_weights()
            self.cal_nonzero_counts()
            self.curr_density = self.total_nozeros / self.total_weights
            print('curr_density: {0:.4f}, final_density:{1:.4f}'.format(self.curr_density,
                                                                         getattr(self.args, 'final_density', 0.0)))
            return True
        return False

    def cal_nonzero_counts(self):
        self.total_nozeros = self.total_weights = 0
        for name in self.names:
            ks = int(np.prod(self._params[name].shape[-3:]))
            self.total_nozeros += int(self._kmask_host[name].sum()) * ks
            self.total_weights += self._params[name].numel()

    # ------------------------------------------------------------------------------------------ prune / grow
    def truncate_weights(self):
        """reference :556-611.  Death pass on the device (kernel L1 in the reference's association order, exact k-th order
        statistic, `<=` comparison); ONE packed device->host copy of all kernel maps; the growth draws on the host with
        Python's ``random`` (bit-identical indices: ``random.sample(range(n), k)`` draws exactly what the reference's
        ``random.sample(list(range(n)), k)`` draws); one broadcast under data parallelism; masks re-expanded on device."""
        L = lib()
        dev = self._params[self.names[0]].device
        nmax = max(self._kmask_host[n].size for n in self.names)
        l1 = torch.empty(nmax, dtype=torch.float32, device=dev)
        thr = torch.empty(1, dtype=torch.float32, device=dev)
        for name in self.names:                                    # death pass (reference :558-581)
            w = self._params[name]
            km_h = self._kmask_host[name]
            r, cc = km_h.shape
            kd, kh, kw = (int(v) for v in w.shape[-3:])
            k_size = kd * kh * kw
            nonzeros = float(int(km_h.sum()) * k_size)             # mask.sum().item()
            zeros = w.numel() - nonzeros
            self.name2nonzeros[name], self.name2zeros[name] = nonzeros, zeros
            prune_num = math.ceil(self.death_rate * nonzeros / k_size)
            num_zeros = math.ceil(zeros / k_size)
            L.dsff_kernel_l1(w.data_ptr(), l1.data_ptr(), r, cc, kd, kh, kw, _stream())
            L.dsff_kth_value(l1.data_ptr(), r * cc, num_zeros + prune_num - 1, thr.data_ptr(), None, _stream())
            L.dsff_death(l1.data_ptr(), thr.data_ptr(), self.kmasks[name].data_ptr(), r * cc, _stream())
            self.num_death[name] = prune_num
        packed = torch.cat([self.kmasks[n].reshape(-1) for n in self.names]).cpu().numpy()     # one D2H copy + one sync
        new_maps, off = {}, 0
        for name in self.names:                                    # growth pass (reference :583-609)
            shp = self._kmask_host[name].shape
            flat = packed[off:off + shp[0] * shp[1]].copy()
            off += shp[0] * shp[1]
            self.num_remove[name] = int(self._kmask_host[name].sum()) - int(flat.sum())
            if self.growth_mode == 'random':
                cand = np.flatnonzero(flat < 1)                    # row-major (dim0, dim1), reference :732
                idx_rand = random.sample(range(cand.shape[0]), self.num_death[name])
                flat[cand[np.asarray(idx_rand, dtype=np.int64)]] = 1
            new_maps[name] = flat.reshape(shp)
        if self.growth_mode == 'gradient':
            new_maps = self._gradient_growth(new_maps)
        new_maps = self.sync_kernel_maps(new_maps)
        for name in self.names:
            self._set_kmask(name, new_maps[name])
        self._push_liveness()
        self.apply_mask()

    def _gradient_growth(self, maps_after_death):
        """reference kernel_grad_growth (:771-790) on the device: per dead kernel (and per depth slice of the kernel: the
        reference sums the LAST TWO axes only) the sum of |weight.grad|, 0 for live ones; threshold = the (num_growth)-th
        largest score (0-based, exact order statistic); every kernel holding a score strictly above it regrows.  The
        device kernel maps already hold the death pass."""
        L = lib()
        dev = self._params[self.names[0]].device
        src = self._grad_source
        nmax = max(self._kmask_host[n].size * int(self._params[n].shape[-3]) for n in self.names)
        score = torch.empty(nmax, dtype=torch.float32, device=dev)
        thr = torch.empty(1, dtype=torch.float32, device=dev)
        for name in self.names:
            w = self._params[name]
            num_growth = self.num_death[name]
            if num_growth == 0:                                     # :773
                continue
            if src is not None and name in src[0]:
                g, sq, max_norm = src[0][name], src[1], src[2]
            else:
                g, sq, max_norm = w.grad, None, 0.0
            if g is None:
                raise RuntimeError("growth_mode='gradient' needs the weight gradient of %s: call Masking.set_gradients(...) "
                                   "(the trainer does) or run backward() so that parameter.grad exists" % name)
            assert g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.numel() == w.numel()
            r, cc = self._kmask_host[name].shape
            kd, kh, kw = (int(v) for v in w.shape[-3:])
            n = r * cc * kd
            if num_growth >= n:
                raise IndexError("kernel_grad_growth: num_growth %d >= %d scores of %s (the reference indexes "
                                 "value[num_growth], core_channel.py:786)" % (num_growth, n, name))
            km = self.kmasks[name]
            L.dsff_grad_score(g.data_ptr(), sq.data_ptr() if sq is not None else None, max_norm, km.data_ptr(), score.data_ptr(),
                              r, cc, kd, kh, kw, _stream())
            # descending order, index num_growth  ==  ascending order, index n - 1 - num_growth
            L.dsff_kth_value(score.data_ptr(), n, n - 1 - num_growth, thr.data_ptr(), None, _stream())
            L.dsff_grow_above(score.data_ptr(), thr.data_ptr(), km.data_ptr(), r, cc, kd, _stream())
        packed = torch.cat([self.kmasks[n].reshape(-1) for n in self.names]).cpu().numpy()
        out, off = {}, 0
        for name in self.names:
            shp = maps_after_death[name].shape
            out[name] = packed[off:off + shp[0] * shp[1]].copy().reshape(shp)
            off += shp[0] * shp[1]
        return out
