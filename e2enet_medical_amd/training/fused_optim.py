"""clip_grad_norm_ + SGD(nesterov) + DSFF mask in two multi-tensor HIP launches.

Replaces ``torch.nn.utils.clip_grad_norm_(params, 12); optimizer.step(); mask.apply_mask()`` of the reference training
iteration (nnUNetTrainer_simple.py:573-576, core_channel.py:427-434).  State lives in the wrapped ``torch.optim.SGD``
(``state[p]['momentum_buffer']``, ``param_groups[0]['lr']``) so checkpoints keep the reference's format.
"""
import torch

from .._lib import lib, ParamEntry


def _stream():
    return torch.cuda.current_stream().cuda_stream


class FusedClipSGD:
    def __init__(self, optimizer: torch.optim.SGD, named_params, max_norm=12.0):
        self.opt = optimizer
        self.named = list(named_params)
        g = optimizer.param_groups[0]
        assert g.get('dampening', 0) == 0, "dampening is not supported"
        self.max_norm = float(max_norm)
        self._table = None
        self._keys = None
        self._sq = None
        self.first = all('momentum_buffer' not in optimizer.state.get(p, {}) for _, p in self.named)

    def _build(self, grads, masks):
        entries, keys = [], []
        dev = self.named[0][1].device
        for name, p in self.named:
            st = self.opt.state[p]
            if 'momentum_buffer' not in st or st['momentum_buffer'] is None:
                st['momentum_buffer'] = torch.zeros_like(p)
            buf = st['momentum_buffer']
            m = masks.get(name) if masks else None
            g = grads[name]
            entries.append(ParamEntry(p.data_ptr(), g.data_ptr(), buf.data_ptr(), m.data_ptr() if m is not None else None,
                                      p.numel()))
            keys.append((p.data_ptr(), g.data_ptr(), buf.data_ptr(), m.data_ptr() if m is not None else 0))
        self._table = torch.frombuffer(bytearray(b"".join(bytes(e) for e in entries)), dtype=torch.uint8).to(dev)
        self._keys = keys
        if self._sq is None:
            self._sq = torch.zeros(1, dtype=torch.float64, device=dev)

    def step(self, grads, masks=None):
        """grads: name -> gradient tensor; masks: name -> fp32 element mask (DSFF) or None."""
        keys = []
        for name, p in self.named:
            st = self.opt.state[p]
            buf = st.get('momentum_buffer')
            m = masks.get(name) if masks else None
            keys.append((p.data_ptr(), grads[name].data_ptr(), buf.data_ptr() if buf is not None else -1,
                         m.data_ptr() if m is not None else 0))
        if self._table is None or keys != self._keys:
            self._build(grads, masks)
        g = self.opt.param_groups[0]
        L = lib()
        n = len(self.named)
        L.grad_sqnorm(self._table.data_ptr(), n, self._sq.data_ptr(), _stream())
        L.sgd_clip_mask_step(self._table.data_ptr(), n, self._sq.data_ptr(), self.max_norm, float(g['lr']),
                             float(g['weight_decay']), float(g['momentum']), 1 if g['nesterov'] else 0,
                             1 if self.first else 0, _stream())
        self.first = False
        from ..engine import note_native_param_write
        note_native_param_write()               # (parameters written through raw pointers: torch's version counters do not move)

    def total_norm(self):
        """sqrt of the last squared gradient norm (device sync)."""
        return float(self._sq.sqrt().item())
