"""Optimizer tail of one training iteration (oracle; test infrastructure only).

Restates reference nnUNetTrainer_simple.py:549-576 (non-AMP branch):
``clip_grad_norm_(all params, 12)`` -> ``SGD(lr, weight_decay 3e-5, momentum .99,
nesterov)`` (:369-370) and e2enet/training/learning_rate/poly_lr.py:16-17.
The arithmetic itself is torch's (torch.nn.utils.clip_grad_norm_, torch.optim.SGD);
this restatement spells out the same update rule on plain tensors.
"""
import torch


def poly_lr(epoch, max_epochs, initial_lr, exponent=0.9):
    return initial_lr * (1 - epoch / max_epochs) ** exponent


def clip_and_sgd_step(params, grads, momentum_buffers, lr, max_norm=12.0, weight_decay=3e-5,
                      momentum=0.99, nesterov=True):
    """params/grads: dict name->tensor (updated in place); momentum_buffers: dict, filled on
    first use with a clone of the (decayed) gradient like torch.optim.SGD.  Returns total_norm."""
    names = list(params.keys())
    norms = torch.stack([torch.linalg.vector_norm(grads[n], 2) for n in names])
    total_norm = torch.linalg.vector_norm(norms, 2)
    clip = torch.clamp(max_norm / (total_norm + 1e-6), max=1.0)
    for n in names:
        g = grads[n] * clip
        g = g.add(params[n], alpha=weight_decay)
        if n not in momentum_buffers:
            momentum_buffers[n] = g.clone()
        else:
            momentum_buffers[n].mul_(momentum).add_(g)
        buf = momentum_buffers[n]
        g = g.add(buf, alpha=momentum) if nesterov else buf
        params[n].add_(g, alpha=-lr)
    return total_norm
