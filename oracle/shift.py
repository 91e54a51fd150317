"""Restricted depth shift (oracle; test infrastructure only).

Follows reference e2enet/network_architecture/unetpp_d.py:45-59 (``torch_shift``
with shift_size=5, dim=2 as hard-set at :89-91): pad D by 2 on both sides, split
the channels with ``torch.chunk(x, 5, 1)``, roll chunk i by (i-2) along D, cat,
narrow back.  Closed form used here:

    out[n, c, d] = x[n, c, d - s(c)]   if 0 <= d - s(c) < D else 0
    s(c) = c // ceil(C / 5) - 2

(``torch.chunk`` makes ceil(C/5)-sized chunks, so C < 5 yields fewer than five
groups; C == 1 is a single group with s = -2.)
"""
import math
import torch


def shift_amounts(num_channels: int, shift_size: int = 5):
    group = math.ceil(num_channels / shift_size)
    pad = shift_size // 2
    return [c // group - pad for c in range(num_channels)]


def depth_shift(x: torch.Tensor, shift_size: int = 5) -> torch.Tensor:
    """x: [N, C, D, H, W] -> shifted copy (zero filled)."""
    n, c, d, h, w = x.shape
    s = shift_amounts(c, shift_size)
    out = torch.zeros_like(x)
    start = 0
    while start < c:
        end = start
        while end < c and s[end] == s[start]:
            end += 1
        sh = s[start]
        if sh >= 0:
            if sh < d:
                out[:, start:end, sh:] = x[:, start:end, :d - sh]
        else:
            if -sh < d:
                out[:, start:end, :d + sh] = x[:, start:end, -sh:]
        start = end
    return out
