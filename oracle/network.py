"""shiftConvPP network forward (oracle; test infrastructure only).

Functional torch-CPU restatement of reference
e2enet/network_architecture/unetpp_d.py:
  * conv block  = depth shift -> Conv3d k(1,3,3) pad(0,1,1) stride s, bias ->
    InstanceNorm3d(eps 1e-5, affine, instance statistics always) ->
    LeakyReLU(0.01)                                            (:61-111)
  * encoder stages with strided first conv ("convolutional pooling")  (:326-371)
  * UNet++ nests loc/up/down 0..4                              (:380-389, :491-550)
  * wiring and output order                                    (:447-488)
Parameters are addressed by the reference's state-dict names (wire format).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple
import math
import torch
import torch.nn.functional as F

from .shift import depth_shift


@dataclass
class NetSpec:
    in_channels: int
    base_features: int
    num_classes: int
    pool_kernels: List[Tuple[int, int, int]]
    convs_per_stage: int = 2
    max_features: int = 320
    feats: List[int] = field(default_factory=list)
    shift_size: int = 5          # reference hard-sets 5 (unetpp_d.py:89; the comment there lists 3/7/11); 1 = 'noshift' ablation
    conv_variant: str = "133"    # "313" / "331": the ablation networks unetpp_d_313.py / unetpp_d_331.py (kernel (3,1,3) /
                                 # (3,3,1), padding on the axes of size 3, and NO shift: their forward has `and False`, :102)
    graph: str = "unetpp"        # "unet": the 'shiftConvPP_nodff' ablation, unetpp_d_nodff.py (plain U-Net wiring, no nests)

    @property
    def num_pool(self):
        return len(self.pool_kernels)


CONV_KERNELS = {"133": (1, 3, 3), "313": (3, 1, 3), "331": (3, 3, 1)}


def make_spec(in_channels, base_features, num_classes, pool_kernels=None, convs_per_stage=2,
              max_features=320, shift_size=5, conv_variant="133", graph="unetpp") -> NetSpec:
    if conv_variant not in CONV_KERNELS:
        raise ValueError("conv_variant must be one of %s" % sorted(CONV_KERNELS))
    if pool_kernels is None:
        pool_kernels = [(2, 2, 2)] * 5
    pool_kernels = [tuple(int(v) for v in k) for k in pool_kernels]
    if graph not in ("unetpp", "unet"):
        raise ValueError("graph must be 'unetpp' or 'unet'")
    if graph == "unetpp" and len(pool_kernels) != 5:
        # reference forward() indexes 6 levels literally (unetpp_d.py:451-483)
        raise ValueError("shiftConvPP needs exactly 5 pooling stages")
    feats = []
    f = base_features
    for _ in range(len(pool_kernels) + 1):
        feats.append(min(f, max_features))
        f = int(round(f * 2))
        f = min(f, max_features)
    return NetSpec(in_channels, base_features, num_classes, pool_kernels, convs_per_stage,
                   max_features, feats, shift_size, conv_variant, graph)


# --------------------------------------------------------------------------- naming
def _block_names(prefix):
    return [prefix + ".conv.weight", prefix + ".conv.bias",
            prefix + ".instnorm.weight", prefix + ".instnorm.bias"]


def nest_nodes(spec: NetSpec, z: int):
    """Nodes of nest z (= diagonal k = 5 - z) as (m, level i). m indexes loc_z/up_z/down_z."""
    k = spec.num_pool - z
    return [(m, k - 1 - m) for m in range(k)]


def loc_block_prefixes(spec: NetSpec, z: int, m: int):
    """state-dict prefixes of the conv blocks inside loc{z}[m], in execution order."""
    n = spec.convs_per_stage
    if z != 0:
        return ["loc%d.%d.0.blocks.%d" % (z, m, b) for b in range(n - 1)]
    return (["loc0.%d.0.blocks.%d" % (m, b) for b in range(n - 1)] +
            ["loc0.%d.1.blocks.0" % m])


def encoder_block_prefixes(spec: NetSpec, stage: int):
    n = spec.convs_per_stage
    if stage < spec.num_pool:
        return ["conv_blocks_context.%d.blocks.%d" % (stage, b) for b in range(n)]
    return (["conv_blocks_context.%d.0.blocks.%d" % (stage, b) for b in range(n - 1)] +
            ["conv_blocks_context.%d.1.blocks.0" % stage])


def concat_channels(spec: NetSpec, level: int) -> int:
    f = spec.feats
    return 2 * f[level] + (f[level - 1] if level > 0 else 0)


def unet_loc_prefixes(spec: NetSpec, u: int):
    """unetpp_d_nodff.py:303-311: conv_blocks_localization[u] = Sequential(Stacked(2 skip -> skip, n - 1), Stacked(skip -> skip, 1))"""
    n = spec.convs_per_stage
    return (["conv_blocks_localization.%d.0.blocks.%d" % (u, b) for b in range(n - 1)] +
            ["conv_blocks_localization.%d.1.blocks.0" % u])


def param_shapes(spec: NetSpec) -> "Dict[str, Tuple[int, ...]]":
    """Ordered like the reference's named_parameters(): loc0..4, conv_blocks_context,
    up0..4, seg_outputs (module registration order, unetpp_d.py:418-438); for graph 'unet'
    conv_blocks_context, conv_blocks_localization, tu, seg_outputs (unetpp_d_nodff.py:238-242)."""
    f = spec.feats
    shapes: Dict[str, Tuple[int, ...]] = {}
    if spec.graph == "unet":
        P = spec.num_pool

        def add(prefix, cin, cout):
            shapes[prefix + ".conv.weight"] = (cout, cin) + CONV_KERNELS[spec.conv_variant]
            shapes[prefix + ".conv.bias"] = (cout,)
            shapes[prefix + ".instnorm.weight"] = (cout,)
            shapes[prefix + ".instnorm.bias"] = (cout,)
        for st in range(P + 1):
            cin = spec.in_channels if st == 0 else f[st - 1]
            for bi, p in enumerate(encoder_block_prefixes(spec, st)):
                add(p, cin if bi == 0 else f[st], f[st])
        for u in range(P):
            lvl = P - 1 - u
            for bi, p in enumerate(unet_loc_prefixes(spec, u)):
                add(p, 2 * f[lvl] if bi == 0 else f[lvl], f[lvl])
        for u in range(P):
            lvl = P - 1 - u
            shapes["tu.%d.weight" % u] = (f[lvl + 1], f[lvl]) + tuple(spec.pool_kernels[lvl])
        for u in range(P):
            shapes["seg_outputs.%d.weight" % u] = (spec.num_classes, f[P - 1 - u], 1, 1, 1)
        return shapes

    def add_block(prefix, cin, cout):
        shapes[prefix + ".conv.weight"] = (cout, cin) + CONV_KERNELS[spec.conv_variant]
        shapes[prefix + ".conv.bias"] = (cout,)
        shapes[prefix + ".instnorm.weight"] = (cout,)
        shapes[prefix + ".instnorm.bias"] = (cout,)

    for z in range(spec.num_pool):
        for m, lvl in nest_nodes(spec, z):
            cin = concat_channels(spec, lvl)
            for bi, p in enumerate(loc_block_prefixes(spec, z, m)):
                add_block(p, cin if bi == 0 else f[lvl], f[lvl])
    for st in range(spec.num_pool + 1):
        cin = spec.in_channels if st == 0 else f[st - 1]
        for bi, p in enumerate(encoder_block_prefixes(spec, st)):
            add_block(p, cin if bi == 0 else f[st], f[st])
    for z in range(spec.num_pool):
        for m, lvl in nest_nodes(spec, z):
            shapes["up%d.%d.weight" % (z, m)] = (f[lvl + 1], f[lvl]) + tuple(spec.pool_kernels[lvl])
    for h in range(4):
        shapes["seg_outputs.%d.weight" % h] = (spec.num_classes, f[h], 1, 1, 1)
    return shapes


def masked_names(spec: NetSpec):
    """Names the reference's Masking.add_module selects (core_channel.py:320-336)."""
    out = []
    for name in param_shapes(spec):
        if (("loc" in name and "context" not in name) or "up" in name) and \
                "bias" not in name and "instnorm" not in name:
            out.append(name)
    return out


def init_params(spec: NetSpec, seed: int = 0, dtype=torch.float32) -> "Dict[str, torch.Tensor]":
    """He-normal(a=0.01) conv / transposed-conv weights, zero bias, unit affine
    (rule of unetpp_d.py:28-36).  RNG draw order is this module's own (name
    order), NOT the reference's; parity fixtures carry explicit weights."""
    g = torch.Generator().manual_seed(seed)
    params = {}
    for name, shp in param_shapes(spec).items():
        if name.endswith("conv.weight") or name.startswith("up") or name.startswith("tu.") or name.startswith("seg_outputs"):
            # kaiming_normal_: fan_in = size(1) * receptive field
            fan_in = shp[1] * int(math.prod(shp[2:]))
            gain = math.sqrt(2.0 / (1 + 0.01 ** 2))
            std = gain / math.sqrt(fan_in)
            params[name] = (torch.randn(shp, generator=g, dtype=dtype) * std)
        elif name.endswith("instnorm.weight"):
            params[name] = torch.ones(shp, dtype=dtype)
        else:
            params[name] = torch.zeros(shp, dtype=dtype)
    return params


# --------------------------------------------------------------------------- forward
class Branches:
    """Branch decisions of the graph's piecewise-linear operators, taken from ANOTHER evaluation of the same graph (the tests take
    them from the engine's activations): per conv block the LeakyReLU mask (u > 0, keyed by the block prefix), per pooled node the
    arg-max index of every window (``F.max_pool3d(return_indices=True)`` layout, keyed by the prefix of the node's last block).
    ``forward(..., branches=b)`` evaluates the graph WITH these decisions: for fixed decisions the network is a smooth function of
    weights and input, so two evaluations of its gradient differ by rounding only -- none of the LeakyReLU-kink and pooling-tie
    flips that make any two fp32 evaluations of the plain graph differ by per cents behind InstanceNorms over few voxels.  No
    reference counterpart (test infrastructure)."""

    def __init__(self, recording=False):
        self.lrelu: Dict[str, torch.Tensor] = {}
        self.pool: Dict[str, torch.Tensor] = {}
        self.recording = recording          # True: forward() takes its own decisions and writes them down here
        self.taps = None                    # a dict: forward() keeps every block's pre-norm conv output in it with retain_grad()
                                            # (diagnostics: d loss / d y per block against the engine's dy buffers)


def _max_pool(x, kernel, branches, key):
    if branches is not None and branches.recording:
        out, idx = F.max_pool3d(x, kernel, return_indices=True)
        branches.pool[key] = idx
        return out
    if branches is None or key not in branches.pool:
        return F.max_pool3d(x, kernel)
    idx = branches.pool[key]
    return x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)


def conv_block(x, w, b, gamma, beta, stride=(1, 1, 1), shift_size=5, branches=None, key=None):
    """unetpp_d.py:102-111 for kernel (1,3,3); unetpp_d_313.py / unetpp_d_331.py:101-110 for the other two kernel
    shapes (their shift is switched off in the source: ``if self.conv.kernel_size == (3, 1, 3) and False``)."""
    k = tuple(w.shape[2:])
    if k == (1, 3, 3):
        x = depth_shift(x, shift_size)
    y = F.conv3d(x, w, b, stride=stride, padding=tuple(1 if v == 3 else 0 for v in k))
    if branches is not None and branches.taps is not None:
        y.retain_grad()
        branches.taps[key] = y
    y = F.instance_norm(y, weight=gamma, bias=beta, eps=1e-5)
    if branches is not None and branches.recording:
        branches.lrelu[key] = y.detach() > 0
    elif branches is not None and key in branches.lrelu:
        return torch.where(branches.lrelu[key], y, y * 0.01)
    return F.leaky_relu(y, 0.01, inplace=True)      # in place like the reference's nonlin_kwargs (unetpp_d.py:248): same values,
                                                    # one pass and one allocation fewer (14-19 % of the CPU step at 64^3)


def _run_blocks(params, prefixes, x, first_stride=(1, 1, 1), shift_size=5, branches=None):
    for bi, p in enumerate(prefixes):
        x = conv_block(x, params[p + ".conv.weight"], params[p + ".conv.bias"],
                       params[p + ".instnorm.weight"], params[p + ".instnorm.bias"],
                       stride=first_stride if bi == 0 else (1, 1, 1), shift_size=shift_size,
                       branches=branches, key=p)
    return x


def forward_unet(spec: NetSpec, params, x, do_ds=True):
    """unetpp_d_nodff.py:356-378: encoder with strided first convs, then per level transposed conv, cat((up, skip)), two conv
    blocks and a 1x1x1 head; outputs [full res, 1/2, ..., lowest] = num_pool tensors."""
    P = spec.num_pool
    skips, cur = [], x
    for st in range(P + 1):
        stride = (1, 1, 1) if st == 0 else spec.pool_kernels[st - 1]
        cur = _run_blocks(params, encoder_block_prefixes(spec, st), cur, stride, shift_size=spec.shift_size)
        if st < P:
            skips.append(cur)
    segs = []
    for u in range(P):
        lvl = P - 1 - u
        cur = F.conv_transpose3d(cur, params["tu.%d.weight" % u], stride=spec.pool_kernels[lvl])
        cur = _run_blocks(params, unet_loc_prefixes(spec, u), torch.cat((cur, skips[lvl]), 1), shift_size=spec.shift_size)
        segs.append(F.conv3d(cur, params["seg_outputs.%d.weight" % u]))
    outs = [segs[-1]] + segs[:-1][::-1]
    return outs if do_ds else outs[0]


def forward(spec: NetSpec, params, x, do_ds=True, return_nodes=False, branches=None):
    """unetpp_d.py:447-488.  Returns [full, 1/2, 1/4, 1/8] logits if do_ds else full only.  ``branches``: see Branches."""
    if spec.graph == "unet":
        assert not return_nodes and branches is None
        return forward_unet(spec, params, x, do_ds)
    P = spec.num_pool
    nodes = {}
    last = {}                               # node -> prefix of its last conv block (the key of its pooling decisions)
    cur = x
    for st in range(P + 1):
        stride = (1, 1, 1) if st == 0 else spec.pool_kernels[st - 1]
        cur = _run_blocks(params, encoder_block_prefixes(spec, st), cur, stride, shift_size=spec.shift_size, branches=branches)
        nodes[(st, 0)] = cur
        last[(st, 0)] = encoder_block_prefixes(spec, st)[-1]
        if st == 0:
            continue
        z = P - st
        for m, lvl in nest_nodes(spec, z):
            j = st - lvl
            parts = [nodes[(lvl, j - 1)],
                     F.conv_transpose3d(nodes[(lvl + 1, j - 1)], params["up%d.%d.weight" % (z, m)],
                                        stride=spec.pool_kernels[lvl])]
            if lvl > 0:
                parts.append(_max_pool(nodes[(lvl - 1, j - 1)], spec.pool_kernels[lvl - 1], branches, last[(lvl - 1, j - 1)]))
            nodes[(lvl, j)] = _run_blocks(params, loc_block_prefixes(spec, z, m), torch.cat(parts, 1), shift_size=spec.shift_size,
                                          branches=branches)
            last[(lvl, j)] = loc_block_prefixes(spec, z, m)[-1]
    outs = [F.conv3d(nodes[(h, P - h)], params["seg_outputs.%d.weight" % h]) for h in range(4)]
    res = outs if do_ds else outs[0]
    if return_nodes:
        return res, nodes
    return res
