"""Execution plan of the shiftConvPP network on the HIP kernels.

Mirrors the wiring of reference ``Generic_UNetPlusPlus.forward`` (unetpp_d.py:447-488) and ``create_nest``
(:491-550) as a static list of ops over pre-allocated HBM buffers.  PyTorch supplies device memory and streams
only; every arithmetic op is a call through the C ABI of libe2e_hip.so (``_lib``).

Design notes:
  * a conv block stores its *pre-norm* output y plus per-(n,c) InstanceNorm scale/shift; consumers apply
    ``lrelu(scale*y+shift)`` on load, so normalisation costs no extra pass over HBM;
  * ``torch.cat`` and the depth shift never materialise: a conv reads through a per-channel plane table;
  * backward is the reverse op list; gradients w.r.t. a tensor are accumulated in one buffer whose first writer
    (in backward order) overwrites and later writers add -- decided statically when the plan is built.
"""
import contextlib
import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

import numpy as np

from ._lib import lib, InChan, OutChan, ParamEntry, SparsePackJob, InSumChan, MmPackJob, RangeSrc, RangeJob

LRELU_SLOPE = 0.01
IN_EPS = 1e-5
DENSE_MIN_DENSITY = float(os.environ.get("E2E_DENSE_MIN_DENSITY", "0.5"))
WGRAD_STREAM = os.environ.get("E2E_WGRAD_STREAM", "1") != "0"      # weight gradients on a second HIP stream
WGRAD_STREAM_MAX_ELEMS = 40000000   # level 0 of 128^3 stays in line
WGRAD_LATE = True            # level-0 weight gradients behind their data gradient, on the side stream
LANES = os.environ.get("E2E_LANES", "1") != "0"                    # deep levels on their own HIP stream (Engine._exec)
# lane of an op = how many of these it passes: output voxels * div <= patch voxels (lane 0 = the caller's stream).  "64" puts
# levels >= 2 on a second stream; "64,4096" gives levels >= 4 a third one
# "auto": plans of at most 2^20 voxels (everything latency bound) use three lanes split at 1/8 and 1/512 of the patch
# (Hippocampus patch fwd+loss+bwd 9.1 -> 7.3 ms), larger plans two lanes split at 1/64 (128^3: a third lane changed nothing)
LANE_DIVS = None              # None = "auto"; a tuple pins the split (tests, tools/scratch)


def _lane_divs(voxels):
    if LANE_DIVS is not None:
        return LANE_DIVS
    return (8, 512) if voxels <= (1 << 20) else (64,)
# fused first pass of the InstanceNorm backward in the LAST writer of a gradient buffer.  1 (default): the pooling backward, which
# is issued behind the other consumers for that purpose (an HBM-bound kernel that reads y and the final dz anyway: free);
# 2: also the load-balanced conv data gradient (measured: +1.4 ms on those launches against -1.0 ms of in_bwd_reduce: a loss); 0: off
FUSE_IN_SUMS = 1
SPARSE2 = os.environ.get("E2E_CONV_SPARSE2", "1") != "0"         # load-balanced kernel for the DSFF-masked full-resolution layers
DENSE_ENABLED = True          # tests switch the matrix-core conv paths off to compare the sparse walk with itself
MM_FORWARD = MM_BACKWARD = True   # tests / diagnostics: K1m in one direction only (the other takes the next kernel in the dispatch order)
# fp16 two-piece matrix-pipe conv (conv133_mm.hip, round 5) for every layer it serves whose kernel map is at least this dense; below
# it a DSFF-masked layer runs on the load-balanced sparse walk (round 4), whose time falls with the density while K1m's does not.
# Round 6 A/B on whole training steps of the config-5 network (profiles/r06_density_switch.txt, K1m / walk): d = 0.05 32.8 / 31.1 ms,
# 0.1 33.3 / 32.4, 0.15 33.4 / 33.8, 0.2 33.7 / 35.4, 0.3 34.4 / 38.4 -- the crossover sits at d ~ 0.13.
# E2E_CONV_MM=0 switches the matrix path off in the library.
MM_MIN_DENSITY = float(os.environ.get("E2E_MM_MIN_DENSITY", "0.125"))


# Native kernels of this package write parameters through raw pointers (the fused optimizer step, Masking.apply_mask): torch's
# per-tensor version counters do not see that.  Every such writer calls note_native_param_write(); an engine re-derives what it caches
# from the weights (packed fp16 weights of the matrix-pipe conv, operand ranges) when this epoch, a parameter's storage or its
# torch version has changed since.  NOT seen: in-place writes through `parameter.data` (a view with its own version counter) or from
# foreign native code -- follow those with `network.weights_changed()`.
PARAM_EPOCH = 0


def note_native_param_write():
    global PARAM_EPOCH
    PARAM_EPOCH += 1


def shift_amounts(num_channels: int, shift_size: int = 5):
    """s(c) of the restricted depth shift (reference unetpp_d.py:45-59: torch.chunk into ceil(C/5)-sized groups,
    group i rolled by i-2)."""
    group = math.ceil(num_channels / shift_size)
    pad = shift_size // 2
    return [c // group - pad for c in range(num_channels)]


def _stream():
    return torch.cuda.current_stream().cuda_stream


@contextlib.contextmanager
def _wgrad_stream(eng, elems):
    """Run the enclosed launches on the plan's weight-gradient stream (Engine.backward sets it up and joins it)."""
    side = getattr(eng, "_wg_active", None)
    if side is None or elems > WGRAD_STREAM_MAX_ELEMS:
        yield eng.wgrad_ws
        return
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        yield eng.wgrad_ws_side              # its own workspace: in-line weight gradients may run at the same time


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _upload_structs(structs, device) -> torch.Tensor:
    raw = b"".join(bytes(s) for s in structs)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


class Act:
    """An activation tensor in HBM.  ``normed`` tensors hold the pre-norm conv output together with the
    InstanceNorm scale/shift that consumers apply on load."""

    def __init__(self, name, shape, normed, device):
        self.name = name
        self.shape = tuple(int(v) for v in shape)          # (B, C, D, H, W)
        self.normed = normed
        self.data = torch.empty(self.shape, dtype=torch.float32, device=device)
        b, c = self.shape[:2]
        if normed:
            self.scale = torch.empty(b * c, dtype=torch.float32, device=device)
            self.shift = torch.empty(b * c, dtype=torch.float32, device=device)
            self.mean = torch.empty(b * c, dtype=torch.float32, device=device)
            self.rstd = torch.empty(b * c, dtype=torch.float32, device=device)
        else:
            self.scale = self.shift = self.mean = self.rstd = None
        self.grad: Optional[torch.Tensor] = None
        self.needs_grad = True
        self._grad_written = False                          # plan-time bookkeeping
        self.producer = None                                # the ConvOp whose pre-norm output this is (normed tensors)
        self.last_writer = None                             # the op that touches .grad last in the backward pass
        self.sums_ready = False                             # set by a last writer that formed the InstanceNorm-backward sums

    @property
    def spatial(self):
        return self.shape[2] * self.shape[3] * self.shape[4]

    def alloc_grad(self):
        if self.grad is None and self.needs_grad:
            self.grad = torch.empty(self.shape, dtype=torch.float32, device=self.data.device)

    def claim_grad_write(self, op=None) -> int:
        """Plan-time: returns the accumulate flag for the next writer (in backward order)."""
        acc = 1 if self._grad_written else 0
        self._grad_written = True
        self.last_writer = op
        return acc


class SparsePlan:
    """One direction of a DSFF-masked conv on the load-balanced kernel (csrc/conv133_sparse.hip): the host plan
    (e2e_conv133_sparse_plan: which output planes a wave owns, which input planes form a chunk -- chosen from the kernel map
    so that the eight waves of a workgroup carry the same work in every chunk), its device copies and the packed weights."""

    def __init__(self, km_host, transpose, device):
        L = lib()
        r, cc = km_host.shape
        self.Q, self.P = (cc, r) if transpose else (r, cc)
        self.groups, self.nchunks = (self.Q + 31) // 32, (self.P + 7) // 8
        qslot = np.empty(self.groups * 32, dtype=np.int32)
        pslot = np.empty(self.groups * self.nchunks * 8, dtype=np.int32)
        quads = np.empty(self.groups * 8 * self.nchunks, dtype=np.uint32)
        woff = np.empty(self.groups * self.nchunks * 8, dtype=np.int32)
        flush, kmax = C.c_int(1), C.c_int(1)
        km_host = np.ascontiguousarray(km_host, dtype=np.uint8)
        L.conv133_sparse_plan(km_host.ctypes.data, r, cc, 1 if transpose else 0, qslot.ctypes.data, pslot.ctypes.data,
                              quads.ctypes.data, woff.ctypes.data, C.addressof(kmax), C.addressof(flush))
        self.flush_every, self.kmax = int(flush.value), int(kmax.value)
        self.woff = torch.from_numpy(woff).to(device)
        self.qslot_host, self.pslot_host = qslot, pslot
        self.qslot = torch.from_numpy(qslot).to(device)
        self.pslot = torch.from_numpy(pslot).to(device)
        self.quads = torch.from_numpy(quads.view(np.int32)).to(device)
        self.wpk = torch.zeros(int(L.conv133_sparse_wpk_floats(self.P, self.Q, self.kmax)), dtype=torch.float32, device=device)
        self.table = None           # forward: plane descriptors in plan order; data gradient: destinations in plan order

    def job(self, w, cin, reverse):
        wq, wp = (9, cin * 9) if reverse else (cin * 9, 9)
        return SparsePackJob(w.data_ptr(), self.wpk.data_ptr(), self.qslot.data_ptr(), self.pslot.data_ptr(), self.quads.data_ptr(),
                             self.woff.data_ptr(), self.groups, self.nchunks, wq, wp, 1 if reverse else 0, self.kmax)


def pack_sparse_weights(jobs, device):
    """ONE launch that (re)builds the packed weights of every planned conv (after an optimizer step or a parameter load)."""
    if not jobs:
        return None
    table = _upload_structs(jobs, device)
    mx = max(j.groups * j.nchunks for j in jobs) * 32 * 8 * 12
    return table, len(jobs), mx


class ConvOp:
    """depth shift + concat + Conv3d(1,3,3) + InstanceNorm statistics (reference ConvDropoutNormNonlin,
    unetpp_d.py:61-111; LeakyReLU/affine are applied by the consumers)."""

    def __init__(self, eng, prefix, sources: Sequence[Act], cout, stride):
        self.eng, self.prefix, self.sources, self.stride = eng, prefix, list(sources), tuple(stride)
        b, _, di, hi, wi = sources[0].shape
        for s in sources:
            assert s.shape[0] == b and s.shape[2:] == sources[0].shape[2:], "concat sources must agree spatially"
        self.cin = sum(s.shape[1] for s in sources)
        self.cout = cout
        self.in_dims = (di, hi, wi)
        sd, sh, sw = self.stride
        self.out_dims = ((di - 1) // sd + 1, (hi - 1) // sh + 1, (wi - 1) // sw + 1)
        self.out = Act(prefix, (b, cout) + self.out_dims, True, eng.device)
        self.out.producer = self
        self.dy_absmax = torch.zeros(1, dtype=torch.int32, device=eng.device)   # bit pattern of max |dy| (e2e_in_lrelu_bwd -> e2e_conv133_wgrad)
        # operand ranges of the fp16 two-piece kernels (conv133_mm.hip, conv133_wgrad_bf3.hip): a bound of |x| over this conv's input
        # planes, derived from the parameters by e2e_conv133_input_ranges (Engine._refresh_weight_caches), and max |w| recorded by the
        # packing launch; the packed weights of the two directions live as long as the plan (rebuilt when the weights change)
        self.x_absmax = torch.zeros(1, dtype=torch.int32, device=eng.device)
        self.w_absmax = torch.zeros(1, dtype=torch.int32, device=eng.device)
        self.wpk_fwd = self.wpk_bwd = None
        self.own_sums = None    # [B, Cout, own_np, 2] fp64 records of the InstanceNorm-backward sums, written by the last writer of out.grad
        self.own_np = 0
        self.np = lib().conv133_num_partials(*self.out_dims, sh, sw)
        self.part = torch.empty(b * cout * self.np * 3, dtype=torch.float64, device=eng.device)      # (count, mean, M2), fp64
        self.w_name = prefix + ".conv.weight"
        self.live = None        # quad words [ceil(Cout/4), ceil(Cin/8)] int32 (e2e_dsff_expand_quads)
        self.live_t = None      # quad words [ceil(Cin/4), ceil(Cout/8)] int32
        self.density = 1.0      # live (out, in) kernels / all, set with the kernel maps (dispatch between the two conv kernels)
        # per input channel plane table
        shifts = shift_amounts(self.cin, getattr(getattr(eng, 'cfg', None), 'shift_size', 5))
        structs = []
        c = 0
        for s in sources:
            cs = s.shape[1]
            plane = s.spatial
            for k in range(cs):
                structs.append(InChan(s.data.data_ptr() + 4 * k * plane,
                                      (s.scale.data_ptr() + 4 * k) if s.normed else None,
                                      (s.shift.data_ptr() + 4 * k) if s.normed else None,
                                      cs * plane, cs, shifts[c], LRELU_SLOPE if s.normed else 1.0, 0))
                c += 1
        self.chans = _upload_structs(structs, eng.device)
        self.in_structs = structs
        c0 = 0
        for s in sources:            # the transposed conv behind a source learns who consumes it and where (operand range of its dy)
            if getattr(s, "up_of", None) is not None:
                s.up_of.consumer = (self, c0, s.shape[1])
            c0 += s.shape[1]
        self.out_structs = None
        self.shifts = shifts
        self.outs = None
        self.do_dgrad = any(s.needs_grad for s in sources)
        # load-balanced sparse kernel (conv133_sparse.hip): plans are built with the kernel maps (Engine.set_kernel_masks)
        self.sp_fwd = self.sp_bwd = None
        self.sparse_ok = bool(lib().conv133_sparse_eligible(self.cin, cout, di, hi, wi, sd, sh, sw))
        self.range_known = False    # set with the range job table (Engine._refresh_weight_caches)

    def plan_backward(self):
        if not self.do_dgrad:
            return
        structs, srcmap = [], []
        c = 0
        for s in self.sources:
            cs = s.shape[1]
            plane = s.spatial
            if s.needs_grad:
                s.alloc_grad()
                acc = s.claim_grad_write(self)
            for k in range(cs):
                if s.needs_grad:
                    structs.append(OutChan(s.grad.data_ptr() + 4 * k * plane, cs * plane, self.shifts[c], acc))
                else:
                    structs.append(OutChan(None, 0, 0, 0))
                srcmap.append((s, k))
                c += 1
        self.outs = _upload_structs(structs, self.eng.device)
        self.out_structs = structs
        self.out_srcs = srcmap
        if self.sp_bwd is not None:
            self.sp_bwd.table = None

    def build_sparse_plans(self, km):
        """km: uint8 [Cout, Cin] kernel map (any device) or None.  Plans for the forward and (if it has one) the data gradient."""
        self.sp_fwd = self.sp_bwd = None
        if km is None or not self.sparse_ok or not SPARSE2:
            return
        kh = km.detach().to("cpu", torch.uint8).numpy()
        dev = self.eng.device
        self.sp_fwd = SparsePlan(kh, False, dev)
        empty = InChan(None, None, None, 0, 0, 0, 1.0, 0)
        self.sp_fwd.table = _upload_structs([self.in_structs[p] if p >= 0 else empty for p in self.sp_fwd.pslot_host], dev)
        if self.do_dgrad:
            self.sp_bwd = SparsePlan(kh, True, dev)

    def sparse_jobs(self):
        w = self.eng.params[self.w_name]
        jobs = []
        if self.sp_fwd is not None:
            jobs.append(self.sp_fwd.job(w, self.cin, False))
        if self.sp_bwd is not None:
            jobs.append(self.sp_bwd.job(w, self.cin, True))
        return jobs

    def _bwd_table(self):
        sp = self.sp_bwd
        if sp.table is None:
            empty = OutChan(None, 0, 0, 0)
            sp.table = _upload_structs([self.out_structs[q] if q >= 0 else empty for q in sp.qslot_host], self.eng.device)
            # fused InstanceNorm-backward sums: for the normalised sources whose gradient buffer THIS op writes last the kernel
            # also forms sum dz lrelu' and sum dz lrelu' xhat of their producer (its e2e_in_lrelu_bwd then skips the first pass)
            sp.fused_srcs = [s for s in self.sources if FUSE_IN_SUMS >= 2 and s.needs_grad and s.normed and s.last_writer is self
                             and s.producer is not None and s.producer.own_sums is not None]
            sp.insum = None
            if sp.fused_srcs:
                none = InSumChan(None, None, None, None, None, None, 0, 0, 0, 0.0)
                rows = []
                for q in sp.qslot_host:
                    s, k = self.out_srcs[q] if q >= 0 else (None, 0)
                    if s is None or not any(s is f for f in sp.fused_srcs):
                        rows.append(none)
                        continue
                    cs = s.shape[1]
                    npr = s.producer.own_np
                    rows.append(InSumChan(s.data.data_ptr() + 4 * k * s.spatial, s.scale.data_ptr() + 4 * k, s.shift.data_ptr() + 4 * k,
                                          s.mean.data_ptr() + 4 * k, s.rstd.data_ptr() + 4 * k, s.producer.own_sums.data_ptr() + 16 * k * npr,
                                          cs * s.spatial, 2 * cs * npr, cs, LRELU_SLOPE))
                sp.insum = _upload_structs(rows, self.eng.device)
        return sp.table

    def x_absmax_ptr(self):
        """the range word of this conv's input planes, or None where no bound exists (a raw network input of <= 4 channels: the kernels
        that serve it do not split operands)"""
        return self.x_absmax.data_ptr() if self.range_known else None

    def range_job(self):
        """e2e_range_job_t: where the bound of |x| over this conv's sources comes from (reference concat order, unetpp_d.py:453-478)"""
        e = self.eng
        srcs = []
        for s in self.sources:
            prod = s.producer if s.normed else getattr(s, "pool_of", None)
            if prod is not None:                                   # a conv block's output, or the max-pool of one
                srcs.append(RangeSrc(1, prod.cout, prod.out.spatial, e.params[prod.prefix + ".instnorm.weight"].data_ptr(),
                                     e.params[prod.prefix + ".instnorm.bias"].data_ptr(), None, 0, 0, None))
            elif getattr(s, "up_of", None) is not None:            # transposed conv of a conv block's output
                up = s.up_of
                prod = up.src.producer
                kd, kh, kw = up.kernel
                srcs.append(RangeSrc(2, prod.cout, prod.out.spatial, e.params[prod.prefix + ".instnorm.weight"].data_ptr(),
                                     e.params[prod.prefix + ".instnorm.bias"].data_ptr(), e.params[up.w_name].data_ptr(), up.cout,
                                     kd * kh * kw, None))
            elif s is e.input and e.input_absmax is not None:
                srcs.append(RangeSrc(3, 0, 0, None, None, None, 0, 0, e.input_absmax.data_ptr()))
            else:
                return None
        while len(srcs) < 3:
            srcs.append(RangeSrc(0, 0, 0, None, None, None, 0, 0, None))
        return RangeJob((RangeSrc * 3)(*srcs), self.x_absmax.data_ptr())

    def mm_jobs(self, directions):
        """e2e_mm_pack_job_t of this layer for the directions in `directions` ('f', 'b'); allocates the packed buffers"""
        if not self.use_mm():
            return []
        w = self.eng.params[self.w_name]
        dev = self.eng.device
        jobs = []
        if "f" in directions and self.use_mm("f"):
            if self.wpk_fwd is None:
                self.wpk_fwd = torch.empty(self.mm_ws_bytes, dtype=torch.uint8, device=dev)
            jobs.append(MmPackJob(w.data_ptr(), _ptr(self.live), self.wpk_fwd.data_ptr(), self.w_absmax.data_ptr(), self.cin, self.cout,
                                  self.cin * 9, 9, 0, 1))
        if "b" in directions and self.use_mm("b") and self.do_dgrad:
            if self.wpk_bwd is None:
                self.wpk_bwd = torch.empty(self.mm_ws_bytes, dtype=torch.uint8, device=dev)
            jobs.append(MmPackJob(w.data_ptr(), _ptr(self.live_t), self.wpk_bwd.data_ptr(), self.w_absmax.data_ptr(), self.cout, self.cin,
                                  9, self.cin * 9, 1, 0 if jobs else 1))
        return jobs

    def pack_mm_standalone(self, directions):
        """an op driven outside an Engine (operator tests, tools/kbench.py): pack this layer's weights now"""
        jobs = self.mm_jobs(directions)
        if jobs:
            table = _upload_structs(jobs, self.eng.device)
            mx = max(((j.Q + 31) // 32) * ((j.P + 15) // 16) * 9 * 512 for j in jobs)
            lib().conv133_mm_pack(table.data_ptr(), len(jobs), mx, _stream())

    def set_input_range(self, max_abs):
        """operator tests: hand the conv a measured max |x| of its input planes (after normalise-on-load) instead of the bound the
        engine derives from the parameters; None = no range word (the fixed 2^3 scale of round 5)"""
        self.range_known = max_abs is not None
        if max_abs is not None:
            self.x_absmax.copy_(torch.tensor([float(max_abs)], dtype=torch.float32).view(torch.int32))

    def use_mm(self, direction=None):
        """fp16 two-piece matrix-pipe kernel (conv133_mm.hip): stride-1 layers of the 16 x 32 tile class with 17..320 channels,
        DSFF-masked or not (the mask is packed into the weights)."""
        if not DENSE_ENABLED or self.mm_ws_bytes <= 0:
            return False
        if direction == "f" and not MM_FORWARD or direction == "b" and not MM_BACKWARD:
            return False
        return self.live is None or self.density >= MM_MIN_DENSITY or not (self.sparse_ok and SPARSE2)

    def use_dense(self):
        """Dense layers (no DSFF map, or a map too dense for the kernel-granular sparse walk to pay) run on the bf16 matrix
        pipe (conv133_dense.hip: fp32-exact three-piece operands) where the shape is served; E2E_DENSE_MIN_DENSITY moves the
        switch-over (measured on 64 -> 32 @128^3: at d = 0.5 the dense kernel 1.16 ms, the sparse walk 1.27 ms; at d = 0.2 the walk 0.69 ms)."""
        if not DENSE_ENABLED or self.dense_ws_bytes <= 0 or getattr(self.eng, "fwd_ws", None) is None:
            return False
        return self.live is None or self.density >= DENSE_MIN_DENSITY

    def forward(self):
        e = self.eng
        p = e.params
        b = self.out.shape[0]
        di, hi, wi = self.in_dims
        sd, sh, sw = self.stride
        L = lib()
        ws = getattr(e, "fwd_ws", None)
        if self.use_mm("f"):
            if not hasattr(e, "_refresh_weight_caches"):
                self.pack_mm_standalone("f")
            L.conv133_fwd_mm(self.chans.data_ptr(), self.cin, self.wpk_fwd.data_ptr(), self.w_absmax.data_ptr(),
                             p[self.prefix + ".conv.bias"].data_ptr(), self.x_absmax_ptr(), self.out.data.data_ptr(), self.part.data_ptr(),
                             b, self.cout, di, hi, wi, _stream())
        elif self.use_dense():
            L.conv133_fwd_dense(self.chans.data_ptr(), self.cin, p[self.w_name].data_ptr(), p[self.prefix + ".conv.bias"].data_ptr(),
                                _ptr(self.live), self.out.data.data_ptr(), self.part.data_ptr(), b, self.cout, di, hi, wi, ws.data_ptr(),
                                ws.numel() * 4, _stream())
        elif self.sp_fwd is not None:                       # DSFF-masked full-resolution layers: load-balanced plan
            sp = self.sp_fwd
            L.conv133_fwd_sparse(sp.table.data_ptr(), self.cin, sp.wpk.data_ptr(), p[self.prefix + ".conv.bias"].data_ptr(),
                                 sp.quads.data_ptr(), sp.woff.data_ptr(), sp.kmax, sp.qslot.data_ptr(), sp.flush_every, self.out.data.data_ptr(),
                                 self.part.data_ptr(), b, self.cout, di, hi, wi, _stream())
        elif ws is not None and self.fwd_ws_bytes > 0:      # deep levels: input-plane chunks split over several workgroups
            L.conv133_fwd_splitk(self.chans.data_ptr(), self.cin, p[self.w_name].data_ptr(),
                                 p[self.prefix + ".conv.bias"].data_ptr(), _ptr(self.live), self.out.data.data_ptr(),
                                 self.part.data_ptr(), b, self.cout, di, hi, wi, sd, sh, sw, ws.data_ptr(), ws.numel() * 4,
                                 _stream())
        else:
            L.conv133_fwd(self.chans.data_ptr(), self.cin, p[self.w_name].data_ptr(), p[self.prefix + ".conv.bias"].data_ptr(),
                          _ptr(self.live), self.out.data.data_ptr(), self.part.data_ptr(), b, self.cout, di, hi, wi,
                          sd, sh, sw, _stream())
        L.in_stats_finalize(self.part.data_ptr(), self.np, p[self.prefix + ".instnorm.weight"].data_ptr(),
                            p[self.prefix + ".instnorm.bias"].data_ptr(), IN_EPS, self.out.scale.data_ptr(),
                            self.out.shift.data_ptr(), self.out.mean.data_ptr(), self.out.rstd.data_ptr(), b,
                            self.cout, _stream())

    def backward(self):
        e = self.eng
        p, g = e.params, e.grads
        o = self.out
        b = o.shape[0]
        di, hi, wi = self.in_dims
        sd, sh, sw = self.stride
        L = lib()
        # dz (w.r.t. the post-activation output) -> dy (w.r.t. the pre-norm conv output), in place; the first pass (two sums per
        # instance) has already been done by the last writer of dz where that writer could (conv133_sparse.hip)
        ready = o.sums_ready
        o.sums_ready = False
        L.in_lrelu_bwd(o.grad.data_ptr(), o.data.data_ptr(), o.mean.data_ptr(), o.rstd.data_ptr(), o.scale.data_ptr(), o.shift.data_ptr(),
                       p[self.prefix + ".instnorm.weight"].data_ptr(),
                       LRELU_SLOPE, g[self.prefix + ".instnorm.weight"].data_ptr(),
                       g[self.prefix + ".instnorm.bias"].data_ptr(), g[self.prefix + ".conv.bias"].data_ptr(),
                       e.in_sums.data_ptr(), b, self.cout, o.spatial, self.own_sums.data_ptr() if ready else None, self.own_np,
                       self.dy_absmax.data_ptr(), _stream())
        late = WGRAD_LATE and getattr(e, "_wg_active", None) is not None and o.data.numel() > WGRAD_STREAM_MAX_ELEMS
        if not late:
            self._wgrad(e, L, g, o, b, di, hi, wi, sd, sh, sw, o.data.numel())
        if self.do_dgrad:
            ws = getattr(e, "fwd_ws", None)
            if self.use_mm("b"):
                if not hasattr(e, "_refresh_weight_caches"):
                    self.pack_mm_standalone("b")
                L.conv133_dgrad_mm(o.grad.data_ptr(), self.dy_absmax.data_ptr(), self.wpk_bwd.data_ptr(), self.w_absmax.data_ptr(),
                                   self.outs.data_ptr(), b, self.cin, self.cout, di, hi, wi, _stream())
            elif self.use_dense():
                L.conv133_dgrad_dense(o.grad.data_ptr(), p[self.w_name].data_ptr(), _ptr(self.live_t), self.outs.data_ptr(), b, self.cin, self.cout,
                                      di, hi, wi, ws.data_ptr(), ws.numel() * 4, _stream())
            elif self.sp_bwd is not None:
                sp = self.sp_bwd
                table = self._bwd_table()
                L.conv133_dgrad_sparse(o.grad.data_ptr(), sp.wpk.data_ptr(), sp.quads.data_ptr(), sp.woff.data_ptr(), sp.kmax, sp.pslot.data_ptr(),
                                       table.data_ptr(), _ptr(sp.insum), sp.flush_every, b, self.cin, self.cout, di, hi, wi, _stream())
                for s in sp.fused_srcs:
                    s.sums_ready = True
            elif ws is not None and self.dgrad_ws_bytes > 0:        # deep levels: split-K (the workspace is idle during backward)
                L.conv133_dgrad_splitk(o.grad.data_ptr(), p[self.w_name].data_ptr(), _ptr(self.live_t), self.outs.data_ptr(),
                                       b, self.cin, self.cout, di, hi, wi, sd, sh, sw, ws.data_ptr(), ws.numel() * 4, _stream())
            else:
                L.conv133_dgrad(o.grad.data_ptr(), p[self.w_name].data_ptr(), _ptr(self.live_t), self.outs.data_ptr(),
                                b, self.cin, self.cout, di, hi, wi, sd, sh, sw, _stream())
        if late:
            # full-resolution layers: the matrix-pipe weight gradient is issued BEHIND the layer's data gradient on the
            # weight-gradient stream, so that it runs beside the next layer's HBM-bound in_lrelu_bwd passes (no LDS, few
            # registers: they share CUs with it) instead of beside the LDS/VALU-bound data gradient
            self._wgrad(e, L, g, o, b, di, hi, wi, sd, sh, sw, 0)

    def _wgrad(self, e, L, g, o, b, di, hi, wi, sd, sh, sw, elems):
        with _wgrad_stream(e, elems) as ws:
            L.conv133_wgrad(self.chans.data_ptr(), o.grad.data_ptr(), g[self.w_name].data_ptr(), ws.data_ptr(),
                            b, self.cin, self.cout, di, hi, wi, sd, sh, sw, self.dy_absmax.data_ptr(), self.x_absmax_ptr(), _stream())

    def wgrad_ws_bytes(self):
        di, hi, wi = self.in_dims
        return lib().conv133_wgrad_ws_bytes(self.out.shape[0], self.cin, self.cout, di, hi, wi, *self.stride)

    @property
    def dense_ws_bytes(self):
        if not hasattr(self, "_dense_ws_bytes"):
            di, hi, wi = self.in_dims
            self._dense_ws_bytes = int(lib().conv133_dense_ws_bytes(self.out.shape[0], self.cin, self.cout, di, hi, wi, *self.stride))
        return self._dense_ws_bytes

    @property
    def mm_ws_bytes(self):
        if not hasattr(self, "_mm_ws_bytes"):
            di, hi, wi = self.in_dims
            self._mm_ws_bytes = int(lib().conv133_mm_ws_bytes(self.out.shape[0], self.cin, self.cout, di, hi, wi, *self.stride))
        return self._mm_ws_bytes

    @property
    def dgrad_ws_bytes(self):
        if not hasattr(self, "_dgrad_ws_bytes"):
            di, hi, wi = self.in_dims
            self._dgrad_ws_bytes = int(lib().conv133_dgrad_ws_bytes(self.out.shape[0], self.cin, self.cout, di, hi, wi, *self.stride))
        return self._dgrad_ws_bytes

    @property
    def fwd_ws_bytes(self):
        if not hasattr(self, "_fwd_ws_bytes"):
            di, hi, wi = self.in_dims
            self._fwd_ws_bytes = int(lib().conv133_fwd_ws_bytes(self.out.shape[0], self.cin, self.cout, di, hi, wi, *self.stride))
        return self._fwd_ws_bytes


class UpOp:
    """nn.ConvTranspose3d(Cin, Cout, k, k, bias=False) (unetpp_d.py:521-522)."""

    def __init__(self, eng, w_name, src: Act, cout, kernel):
        self.eng, self.w_name, self.src, self.cout, self.kernel = eng, w_name, src, cout, tuple(kernel)
        b, cin, d, h, w = src.shape
        self.cin = cin
        kd, kh, kw = self.kernel
        self.out = Act(w_name, (b, cout, d * kd, h * kh, w * kw), False, eng.device)
        self.out.up_of = self if (src.normed and src.producer is not None) else None     # (ConvOp.range_job)
        self.live = None      # [Cout, ceil(Cin/32)]
        self.live_t = None    # [Cin, ceil(Cout/32)]
        self.acc = 0
        # operand ranges of the fp16 two-piece GEMMs (convt.hip, round 6): words [x bound, max |w|, dy factor] filled by
        # e2e_conv133_input_ranges (Engine._refresh_weight_caches); the bound of dy is words[2] x the consuming conv's max |dy| word
        self.words = torch.zeros(3, dtype=torch.int32, device=eng.device)
        self.consumer = None          # (ConvOp, first channel, channels) of the concat this output feeds
        self.ranges_known = False

    def range_jobs(self):
        """e2e_range_job_t x 3 (activations, weights, dy factor) or None when a bound cannot be derived (raw source, no consumer)"""
        e, s = self.eng, self.src
        if not (s.normed and s.producer is not None and self.consumer is not None):
            return None
        none = RangeSrc(0, 0, 0, None, None, None, 0, 0, None)
        prod = s.producer
        w = e.params[self.w_name]
        conv, c0, cs = self.consumer
        wc = e.params[conv.w_name]
        base = self.words.data_ptr()
        jx = RangeJob((RangeSrc * 3)(RangeSrc(1, prod.cout, prod.out.spatial, e.params[prod.prefix + ".instnorm.weight"].data_ptr(),
                                              e.params[prod.prefix + ".instnorm.bias"].data_ptr(), None, 0, 0, None), none, none), base)
        jw = RangeJob((RangeSrc * 3)(RangeSrc(4, 0, w.numel(), None, None, w.data_ptr(), 0, 0, None), none, none), base + 4)
        jd = RangeJob((RangeSrc * 3)(RangeSrc(5, c0, conv.cin, None, None, wc.data_ptr(), conv.cout, cs, None), none, none), base + 8)
        return [jx, jw, jd]

    def set_ranges(self, x_max, w_max, dy_factor):
        """operator tests: measured maxima instead of the derived bounds (None: no words -> bf16 three-piece operands)"""
        self.ranges_known = x_max is not None
        if x_max is not None:
            self.words.copy_(torch.tensor([float(x_max), float(w_max), float(dy_factor)], dtype=torch.float32).view(torch.int32))

    def _w(self, i):
        return self.words.data_ptr() + 4 * i if self.ranges_known else None

    def plan_backward(self):
        self.src.alloc_grad()
        self.acc = self.src.claim_grad_write(self)

    def forward(self):
        s = self.src
        b, cin, d, h, w = s.shape
        lib().convT_fwd(s.data.data_ptr(), _ptr(s.scale), _ptr(s.shift), LRELU_SLOPE, self.eng.params[self.w_name].data_ptr(),
                        _ptr(self.live), self.out.data.data_ptr(), b, cin, self.cout, d, h, w, *self.kernel, self._w(0), self._w(1),
                        _stream())

    def backward(self):
        s, e = self.src, self.eng
        b, cin, d, h, w = s.shape
        L = lib()
        # the bound of dy = (the consuming conv's max |dy| word, written by its e2e_in_lrelu_bwd earlier in this pass) x words[2]
        dya = self.dy_word if getattr(self, "dy_word", None) is not None else \
            (self.consumer[0].dy_absmax.data_ptr() if (self.ranges_known and self.consumer is not None) else None)
        with _wgrad_stream(e, self.out.data.numel()) as ws:
            L.convT_wgrad(s.data.data_ptr(), _ptr(s.scale), _ptr(s.shift), LRELU_SLOPE, self.out.grad.data_ptr(),
                          e.grads[self.w_name].data_ptr(), ws.data_ptr(), b, cin, self.cout, d, h, w, *self.kernel,
                          self._w(0), dya, self._w(2) if dya is not None else None, _stream())
        L.convT_dgrad(self.out.grad.data_ptr(), e.params[self.w_name].data_ptr(), _ptr(self.live_t), s.grad.data_ptr(),
                      self.acc, b, cin, self.cout, d, h, w, *self.kernel, self._w(1), dya, self._w(2) if dya is not None else None,
                      _stream())

    def wgrad_ws_bytes(self):
        b, cin, d, h, w = self.src.shape
        return lib().convT_wgrad_ws_bytes(b, cin, self.cout, d, h, w, *self.kernel)


class PoolOp:
    """nn.MaxPool3d(k) of the down-fusion branch (unetpp_d.py:523-524)."""

    def __init__(self, eng, name, src: Act, kernel):
        self.eng, self.src, self.kernel = eng, src, tuple(kernel)
        b, c, d, h, w = src.shape
        kd, kh, kw = self.kernel
        self.out = Act(name, (b, c, d // kd, h // kh, w // kw), False, eng.device)
        self.out.pool_of = src.producer if src.normed else None                          # (ConvOp.range_job)
        self.acc = 0

    def plan_backward(self):
        self.src.alloc_grad()
        self.acc = self.src.claim_grad_write(self)

    def forward(self):
        s = self.src
        b, c, d, h, w = s.shape
        lib().maxpool_fwd(s.data.data_ptr(), _ptr(s.scale), _ptr(s.shift), LRELU_SLOPE, self.out.data.data_ptr(),
                          b, c, d, h, w, *self.kernel, _stream())

    def backward(self):
        s = self.src
        b, c, d, h, w = s.shape
        fuse = FUSE_IN_SUMS >= 1 and s.normed and s.last_writer is self and s.producer is not None and s.producer.own_sums is not None
        lib().maxpool_bwd(s.data.data_ptr(), _ptr(s.scale), _ptr(s.shift), LRELU_SLOPE, self.out.grad.data_ptr(),
                          s.grad.data_ptr(), self.acc, b, c, d, h, w, *self.kernel,
                          s.mean.data_ptr() if fuse else None, s.rstd.data_ptr() if fuse else None,
                          s.producer.own_sums.data_ptr() if fuse else None, _stream())
        if fuse:
            s.sums_ready = True


class HeadOp:
    """nn.Conv3d(C, K, 1, bias=False) segmentation head (unetpp_d.py:394-401, :480-483)."""

    def __init__(self, eng, w_name, src: Act, num_classes):
        self.eng, self.w_name, self.src, self.k = eng, w_name, src, num_classes
        b, c = src.shape[:2]
        self.out = Act(w_name, (b, num_classes) + src.shape[2:], False, eng.device)
        self.acc = 0
        self.active = True           # deep supervision off: only head 0 runs

    def plan_backward(self):
        self.src.alloc_grad()
        self.acc = self.src.claim_grad_write(self)

    def forward(self):
        s = self.src
        b, c = s.shape[:2]
        lib().head1x1_fwd(s.data.data_ptr(), _ptr(s.scale), _ptr(s.shift), LRELU_SLOPE,
                          self.eng.params[self.w_name].data_ptr(), self.out.data.data_ptr(), b, c, self.k, s.spatial,
                          _stream())

    def backward(self):
        s, e = self.src, self.eng
        b, c = s.shape[:2]
        L = lib()
        with _wgrad_stream(e, s.data.numel()) as ws:
            L.head1x1_wgrad(s.data.data_ptr(), _ptr(s.scale), _ptr(s.shift), LRELU_SLOPE, self.out.grad.data_ptr(),
                            e.grads[self.w_name].data_ptr(), ws.data_ptr(), b, c, self.k, s.spatial, _stream())
        L.head1x1_dgrad(self.out.grad.data_ptr(), e.params[self.w_name].data_ptr(), s.grad.data_ptr(), self.acc, b, c,
                        self.k, s.spatial, _stream())


class NetConfig:
    """Static description of a shiftConvPP network (what reference Generic_UNetPlusPlus.__init__ derives,
    unetpp_d.py:227-445)."""

    def __init__(self, in_channels, base_features, num_classes, pool_kernels, convs_per_stage=2, max_features=320,
                 shift_size=5, graph="unetpp"):
        if graph not in ("unetpp", "unet"):
            raise ValueError("graph must be 'unetpp' (unetpp_d.py) or 'unet' (unetpp_d_nodff.py)")
        self.graph = graph
        if graph == "unetpp" and len(pool_kernels) != 5:
            # reference forward() indexes six levels literally (unetpp_d.py:451-483)
            raise ValueError("shiftConvPP needs exactly 5 pooling stages, got %d" % len(pool_kernels))
        self.in_channels, self.base_features, self.num_classes = in_channels, base_features, num_classes
        self.pool_kernels = [tuple(int(v) for v in k) for k in pool_kernels]
        self.convs_per_stage, self.max_features = convs_per_stage, max_features
        if shift_size < 1 or shift_size % 2 == 0:
            raise ValueError("shift_size must be odd and >= 1 (reference unetpp_d.py:89 uses 5; 1 disables the shift)")
        self.shift_size = shift_size      # groups of the restricted depth shift (3/7/11, 1 = 'noshift')
        feats, f = [], base_features
        for _ in range(len(pool_kernels) + 1):
            feats.append(min(f, max_features))
            f = min(int(round(f * 2)), max_features)
        self.feats = feats

    @property
    def num_pool(self):
        return len(self.pool_kernels)

    def nest_nodes(self, z):
        k = self.num_pool - z
        return [(m, k - 1 - m) for m in range(k)]

    def loc_prefixes(self, z, m):
        n = self.convs_per_stage
        if z != 0:
            return ["loc%d.%d.0.blocks.%d" % (z, m, b) for b in range(n - 1)]
        return ["loc0.%d.0.blocks.%d" % (m, b) for b in range(n - 1)] + ["loc0.%d.1.blocks.0" % m]

    def unet_loc_prefixes(self, u):
        """unetpp_d_nodff.py:303-311: conv_blocks_localization[u] = Sequential(Stacked(2 skip -> skip, n - 1), Stacked(skip -> skip, 1))"""
        n = self.convs_per_stage
        return (["conv_blocks_localization.%d.0.blocks.%d" % (u, b) for b in range(n - 1)] +
                ["conv_blocks_localization.%d.1.blocks.0" % u])

    def encoder_prefixes(self, stage):
        n = self.convs_per_stage
        if stage < self.num_pool:
            return ["conv_blocks_context.%d.blocks.%d" % (stage, b) for b in range(n)]
        return (["conv_blocks_context.%d.0.blocks.%d" % (stage, b) for b in range(n - 1)] +
                ["conv_blocks_context.%d.1.blocks.0" % stage])


class Engine:
    """Plan + executor for one (batch, patch) shape."""

    def __init__(self, cfg: NetConfig, params: Dict[str, torch.Tensor], batch: int, patch: Tuple[int, int, int],
                 device, training: bool = True):
        lib()                                    # fail loudly if the HIP library is missing
        self.cfg, self.params, self.device, self.training = cfg, params, device, training
        self.batch, self.patch = batch, tuple(patch)
        P = cfg.num_pool
        self.ops: List = []
        self.conv_ops: Dict[str, ConvOp] = {}
        self.up_ops: Dict[str, UpOp] = {}
        self.input = Act("input", (batch, cfg.in_channels) + self.patch, False, device)
        self.input.needs_grad = False
        # a network input of more than four channels reaches split-operand kernels (weight gradient, K1m from 17 channels): its
        # max |x| is measured at every forward (e2e_absmax_word); narrower inputs are served by kernels that do not split
        self.input_absmax = torch.zeros(1, dtype=torch.int32, device=device) if cfg.in_channels > 4 else None
        self.heads: List[HeadOp] = []
        if cfg.graph == "unet":
            self._build_unet(cfg)
        else:
            self._build_unetpp(cfg)
        self.grads: Dict[str, torch.Tensor] = {}
        # callable(lo, hi): flat gradient slice [lo, hi) is final (data-parallel overlap).  Called with a private communication
        # stream current (it waits for everything issued so far; nothing waits for it until backward() returns, where the
        # caller's stream joins it); the tail slice is handed over on the caller's stream after that join
        self.grad_bucket_hook = None
        self._comm_stream = None           # the stream the bucket hook is called on (backward)
        self.batch_dice_hook = None        # callable(fp64 tensor [K*3]): all-reduce of the folded tp/fp/fn (data-parallel batch dice)
        self._backward_ready = False
        self.loss_ws = None
        self.loss_val = None
        self.generation = 0                # bumped by every forward(): activations are reused in place
        fws = max([max(op.fwd_ws_bytes, op.dgrad_ws_bytes if op.do_dgrad else 0, op.dense_ws_bytes) for op in self.conv_ops.values()] + [0])
        # split-K partial sums (deep levels) / packed weights of the matrix-core conv; one per lane (see _exec)
        self.lane_divs = _lane_divs(batch * self.patch[0] * self.patch[1] * self.patch[2])
        nl = len(self.lane_divs) + 1
        self._fwd_ws = [torch.empty(fws // 4, dtype=torch.float32, device=self.device) if fws > 0 else None for _ in range(nl)]
        self._in_sums, self._wgrad_ws = [None] * nl, [None] * nl
        self._lane = 0
        self._bwd_order = self._backward_order()
        self._plan_lanes()
        self.pre_forward_hook = None       # callable(): set by the owning network, brings masks / parameters up to date
        self._eval_counts = None
        # HIP-graph replay of the op lists for small plans (the host issues ~300-600 launches per pass: a Hippocampus-sized
        # patch is launch bound, 128^3 is not).  E2E_GRAPHS=0 / 1 / auto (default: plans of at most E2E_GRAPH_MAX_VOXELS voxels)
        self._wg_side, self._wg_active = None, None
        self._graphs = {}                  # key -> torch.cuda.CUDAGraph
        self._graph_seen = set()           # keys that ran eagerly once (lazy allocations done)
        self.maps_generation = 0           # bumped when the liveness tables are replaced (their pointers are baked into a graph)
        self._sparse_jobs = None           # pack-job table of the load-balanced convs (rebuilt with the kernel maps)
        self._tgt_static = None
        self._mm_tables = {}               # directions -> (dispatch key, device job table, njobs, max elems)
        self._mm_packed = {}               # direction -> weights key its packed buffers were built from
        self._range_table = None           # (device job table, njobs) of e2e_conv133_input_ranges
        self._range_key = None
        self.weights_epoch = 0             # bumped by weights_changed(): in-place parameter writes nothing else can see

    def _build_unetpp(self, cfg):
        """reference Generic_UNetPlusPlus.forward (unetpp_d.py:447-488) and create_nest (:491-550)"""
        P = cfg.num_pool
        nodes = {}
        cur = self.input
        for st in range(P + 1):
            stride = (1, 1, 1) if st == 0 else cfg.pool_kernels[st - 1]
            for bi, prefix in enumerate(cfg.encoder_prefixes(st)):
                op = self._add_conv(prefix, [cur], cfg.feats[st], stride if bi == 0 else (1, 1, 1))
                cur = op.out
            nodes[(st, 0)] = cur
            if st == 0:
                continue
            z = P - st
            for m, lvl in cfg.nest_nodes(z):
                j = st - lvl
                up = UpOp(self, "up%d.%d.weight" % (z, m), nodes[(lvl + 1, j - 1)], cfg.feats[lvl], cfg.pool_kernels[lvl])
                self.ops.append(up)
                self.up_ops[up.w_name] = up
                srcs = [nodes[(lvl, j - 1)], up.out]
                if lvl > 0:
                    pool = PoolOp(self, "down%d.%d" % (z, m), nodes[(lvl - 1, j - 1)], cfg.pool_kernels[lvl - 1])
                    self.ops.append(pool)
                    srcs.append(pool.out)
                t = None
                for prefix in cfg.loc_prefixes(z, m):
                    op = self._add_conv(prefix, srcs if t is None else [t], cfg.feats[lvl], (1, 1, 1))
                    t = op.out
                nodes[(lvl, j)] = t
        self.nodes = nodes
        for h in range(4):
            head = HeadOp(self, "seg_outputs.%d.weight" % h, nodes[(h, P - h)], cfg.num_classes)
            self.heads.append(head)
            self.ops.append(head)

    def _build_unet(self, cfg):
        """The 'shiftConvPP_nodff' ablation (reference unetpp_d_nodff.py:356-378): encoder with strided first convs, then per
        level ConvTranspose3d, cat((up, skip)), two conv blocks, a 1x1x1 head; heads ordered [full res, 1/2, ..., lowest]."""
        P = cfg.num_pool
        skips, cur = [], self.input
        for st in range(P + 1):
            stride = (1, 1, 1) if st == 0 else cfg.pool_kernels[st - 1]
            for bi, prefix in enumerate(cfg.encoder_prefixes(st)):
                op = self._add_conv(prefix, [cur], cfg.feats[st], stride if bi == 0 else (1, 1, 1))
                cur = op.out
            if st < P:
                skips.append(cur)
        self.nodes = {(st, 0): a for st, a in enumerate(skips)}
        by_u = []
        for u in range(P):
            lvl = P - 1 - u
            up = UpOp(self, "tu.%d.weight" % u, cur, cfg.feats[lvl], cfg.pool_kernels[lvl])
            self.ops.append(up)
            self.up_ops[up.w_name] = up
            t = None
            for prefix in cfg.unet_loc_prefixes(u):
                op = self._add_conv(prefix, [up.out, skips[lvl]] if t is None else [t], cfg.feats[lvl], (1, 1, 1))
                t = op.out
            cur = t
            self.nodes[(lvl, 1)] = t
            head = HeadOp(self, "seg_outputs.%d.weight" % u, t, cfg.num_classes)
            self.ops.append(head)
            by_u.append(head)
        self.heads = [by_u[-1]] + by_u[:-1][::-1]

    def _add_conv(self, prefix, sources, cout, stride):
        op = ConvOp(self, prefix, sources, cout, stride)
        self.ops.append(op)
        self.conv_ops[op.w_name] = op
        return op

    # ------------------------------------------------------------------------------------------ sparsity
    def set_kernel_masks(self, kmasks: Dict[str, torch.Tensor]):
        """kmasks: weight name -> uint8 [dim0, dim1] kernel map (1 = alive).  Builds the liveness bit tables the
        kernels walk.  Names absent from the dict are treated as dense."""
        L = lib()
        self.maps_generation += 1
        self._graphs.clear()
        self._graph_seen.clear()
        self._sparse_jobs = None
        for name, op in list(self.conv_ops.items()) + list(self.up_ops.items()):
            km = kmasks.get(name)
            if isinstance(op, ConvOp):
                op.build_sparse_plans(km)
            if km is None:
                op.live = op.live_t = None
                op.density = 1.0
                continue
            r, cc = km.shape
            op.density = float(km.float().mean().item())
            km = km.to(device=self.device, dtype=torch.uint8).contiguous()
            if isinstance(op, ConvOp):         # weight [Cout, Cin, ...]: quad words (4 output x 8 input planes)
                rows = torch.empty(((r + 3) // 4) * ((cc + 7) // 8), dtype=torch.int32, device=self.device)
                cols = torch.empty(((cc + 3) // 4) * ((r + 7) // 8), dtype=torch.int32, device=self.device)
                L.dsff_expand_quads(km.data_ptr(), rows.data_ptr(), cols.data_ptr(), r, cc, _stream())
                op.live, op.live_t = rows, cols
            else:                              # weight [Cin, Cout, ...]: 32-plane words, rows index in channels
                rows = torch.empty(r * ((cc + 31) // 32), dtype=torch.int32, device=self.device)
                cols = torch.empty(cc * ((r + 31) // 32), dtype=torch.int32, device=self.device)
                L.dsff_expand(km.data_ptr(), None, rows.data_ptr(), cols.data_ptr(), r, cc, 1, _stream())
                op.live, op.live_t = cols, rows

    def kernel_masks_from_weights(self, names=None):
        """Inference: a kernel is alive iff any tap is non-zero (pruned kernels are exact zeros in checkpoints)."""
        L = lib()
        out = {}
        for name, op in list(self.conv_ops.items()) + list(self.up_ops.items()):
            if names is not None and name not in names:
                continue
            w = self.params[name]
            r, cc = w.shape[0], w.shape[1]
            ks = w[0, 0].numel()
            km = torch.empty((r, cc), dtype=torch.uint8, device=self.device)
            L.dsff_kmask_from_weights(w.data_ptr(), km.data_ptr(), r, cc, ks, _stream())
            out[name] = km
        return out

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, deep_supervision: bool = True):
        assert x.is_cuda and x.dtype == torch.float32, "engine input must be a float32 GPU tensor"
        assert tuple(x.shape) == self.input.shape, "engine built for %s, got %s" % (self.input.shape, tuple(x.shape))
        if self.pre_forward_hook is not None:
            # a plan that is held across iterations (benchmark loop, tests) must see the kernel maps of a prune/grow that
            # happened since it was handed out: with stale maps the regrown kernels stay skipped although their weights
            # have started to move (found by the nine-iteration trajectory test)
            self.pre_forward_hook()
        self.input.data.copy_(x)
        self.generation += 1
        for h in self.heads:               # (set here, not inside the op list: a graph replay does not run Python)
            h.active = bool(deep_supervision) or h is self.heads[0]
        self._run(("fwd", bool(deep_supervision)), self._forward_ops)
        outs = [h.out.data for h in self.heads]
        return outs if deep_supervision else outs[0]

    def _pack_sparse(self):
        """packed weights of the planned convs (one launch): the weights may have moved since the last pass"""
        # the job table is cached under the dispatch decisions it was built for (advisor, round 4: a table built while an op went to a
        # matrix-pipe kernel left that op's packed weights zero when tests / knobs sent it to the planned walk afterwards)
        def matrix_only(op):
            return (op.use_mm("f") and op.use_mm("b")) or op.use_dense()
        key = tuple(matrix_only(op) for op in self.conv_ops.values())
        if self._sparse_jobs is None or self._sparse_jobs[0] != key:
            jobs = [j for op in self.conv_ops.values() if not matrix_only(op) for j in op.sparse_jobs()]
            self._sparse_jobs = (key, pack_sparse_weights(jobs, self.device) or ())
        if self._sparse_jobs[1]:
            table, n, mx = self._sparse_jobs[1]
            lib().conv133_sparse_pack(table.data_ptr(), n, mx, _stream())

    # ------------------------------------------------------------------------------------------ caches derived from the weights
    def weights_changed(self):
        """Tell the plan that parameters were modified in a way it cannot see (in place through ``parameter.data``, foreign native
        code): the packed matrix-pipe weights and the operand ranges are rebuilt at the next pass."""
        self.weights_epoch += 1

    def _weights_key(self):
        ptrs = tuple(p.data_ptr() for p in self.params.values())
        return ptrs, (PARAM_EPOCH, self.weights_epoch, self.maps_generation, tuple(p._version for p in self.params.values()))

    def _refresh_weight_caches(self, directions):
        """What the split-operand kernels cache from the parameters, rebuilt only when the parameters (storage, torch version,
        native-write epoch) or the kernel maps have changed since it was built -- in a training loop once per optimizer step, in
        sliding-window inference once per checkpoint (round 5: a packing launch in front of every conv launch):
          * the operand-range words of every conv (e2e_conv133_input_ranges: ONE launch),
          * the packed fp16 two-piece weights of the K1m layers in the directions asked for (e2e_conv133_mm_pack: two launches).
        Inside a HIP-graph capture (small plans, E2E_GRAPHS) the launches are issued unconditionally: a replay runs no Python."""
        L = lib()
        ptrs, state = self._weights_key()
        always = self._graph_ok() or torch.cuda.is_current_stream_capturing()
        if self.input_absmax is not None and "f" in directions:      # (depends on the data: every forward)
            L.absmax_word(self.input.data.data_ptr(), self.input.data.numel(), self.input_absmax.data_ptr(), _stream())
        if always or self._range_key != (ptrs, state):
            if self._range_table is None or self._range_table[2] != ptrs:
                jobs = []
                for op in self.conv_ops.values():
                    jb = op.range_job()
                    op.range_known = jb is not None
                    if jb is not None:
                        jobs.append(jb)
                for up in self.up_ops.values():
                    jbs = up.range_jobs()
                    up.ranges_known = jbs is not None
                    if jbs is not None:
                        jobs += jbs
                ws = torch.empty(max(4, int(L.conv133_input_ranges_ws_bytes(len(jobs))) // 4), dtype=torch.float32, device=self.device)
                self._range_table = (_upload_structs(jobs, self.device) if jobs else None, len(jobs), ptrs, ws)
            if self._range_table[1]:
                L.conv133_input_ranges(self._range_table[0].data_ptr(), self._range_table[1], self._range_table[3].data_ptr(), _stream())
            self._range_key = (ptrs, state)
        # which layers run on K1m in which direction is part of what a packed set is valid for (tests and knobs move it)
        dkey = (tuple((op.use_mm("f"), op.use_mm("b")) for op in self.conv_ops.values()), self.maps_generation, ptrs)
        todo = "".join(d for d in directions if always or self._mm_packed.get(d) != (dkey, state))
        if not todo:
            return
        tab = self._mm_tables.get(todo)
        if tab is None or tab[0] != dkey:
            jobs = [j for op in self.conv_ops.values() for j in op.mm_jobs(todo)]
            mx = max([((j.Q + 31) // 32) * ((j.P + 15) // 16) * 9 * 512 for j in jobs] + [0])
            tab = (dkey, _upload_structs(jobs, self.device) if jobs else None, len(jobs), mx)
            self._mm_tables[todo] = tab
        if tab[2]:
            L.conv133_mm_pack(tab[1].data_ptr(), tab[2], tab[3], _stream())
        for d in todo:
            self._mm_packed[d] = (dkey, state)

    def _forward_ops(self):
        self._refresh_weight_caches("fb" if self._backward_ready else "f")
        self._pack_sparse()

        def act(op):
            if not (isinstance(op, HeadOp) and not op.active):
                op.forward()
        self._exec(range(len(self.ops)), self._deps_fwd, self._ev_fwd, act)

    # ------------------------------------------------------------------------------------------ two lanes
    # The UNet++ nest is a wavefront: the deep end of diagonal d+1 only needs the encoder, not the full-resolution end of
    # diagonal d.  Ops of the deep levels (tiny grids, latency bound) are issued on a second HIP stream ('light' lane) so
    # they run beside the full-resolution kernels instead of between them.  The op ORDER is unchanged (every wait points
    # at an op issued earlier), gradient buffers are written in the planned order (overwrite first, then accumulate), so
    # results are bit-identical to the single-stream pass.  Scratch buffers exist once per lane.
    fwd_ws = property(lambda self: self._fwd_ws[self._lane])
    in_sums = property(lambda self: self._in_sums[self._lane])
    wgrad_ws = property(lambda self: self._wgrad_ws[self._lane])

    @staticmethod
    def _reads(op):
        return list(op.sources) if isinstance(op, ConvOp) else [op.src]

    def _plan_lanes(self):
        vox = self.patch[0] * self.patch[1] * self.patch[2]
        self._lane_of = [sum(1 for dv in self.lane_divs if op.out.spatial * dv <= vox) for op in self.ops]
        writer = {}
        self._deps_fwd = []
        for i, op in enumerate(self.ops):
            self._deps_fwd.append(sorted({writer[id(a)] for a in self._reads(op)
                                          if id(a) in writer and self._lane_of[writer[id(a)]] != self._lane_of[i]}))
            writer[id(op.out)] = i
        self._ev_fwd = self._events_for(self._deps_fwd)
        self._deps_bwd, self._ev_bwd = None, None
        self._lane_streams = None

    def _backward_order(self):
        """The reverse op list, except that a pooling backward is issued LAST among the writers of its source's gradient buffer
        (right in front of the source's producer): maxpool_bwd is an HBM-bound pass that reads the source's pre-norm values and
        the final dz of every cell anyway, so as last writer it forms the first pass of the producer's InstanceNorm backward at
        no cost (FUSE_IN_SUMS).  For a buffer with two writers the sum is the same to the bit (a + b = b + a); with three (levels
        >= 1: transposed conv, conv, pool) the fp32 association changes."""
        order = list(reversed(range(len(self.ops))))
        if FUSE_IN_SUMS < 1:
            return order
        index = {id(op): i for i, op in enumerate(self.ops)}
        for i, op in enumerate(self.ops):
            prod = getattr(op.src, "producer", None) if isinstance(op, PoolOp) else None
            if prod is None or id(prod) not in index:
                continue
            order.remove(i)
            order.insert(order.index(index[id(prod)]), i)
        return order

    def _plan_lanes_backward(self):
        last = {}                              # gradient buffer -> op that touched it last (in backward order)
        self._deps_bwd = [None] * len(self.ops)
        for i in self._bwd_order:
            op = self.ops[i]
            touched = [op.out] + [a for a in self._reads(op) if a.grad is not None]
            self._deps_bwd[i] = sorted({last[id(a)] for a in touched
                                        if id(a) in last and self._lane_of[last[id(a)]] != self._lane_of[i]})
            for a in touched:
                last[id(a)] = i
        self._ev_bwd = self._events_for(self._deps_bwd)

    @staticmethod
    def _events_for(deps):
        return {j: None for d in deps for j in d}

    def _lanes_on(self):
        return LANES and any(self._lane_of) and not self._graph_ok() and not torch.cuda.is_current_stream_capturing()

    def _exec(self, order, deps, events, action, checkpoint=None):
        """Issue action(op) for the ops in `order`; `checkpoint(op, wait_all)` (data-parallel bucket hook) runs after op was issued:
        wait_all(stream) makes `stream` wait for everything issued so far on the caller's stream and on the lanes, without
        holding any of THEM up."""
        main = torch.cuda.current_stream()
        if not self._lanes_on():
            for i in order:
                action(self.ops[i])
                if checkpoint is not None:
                    checkpoint(self.ops[i], lambda st: st.wait_stream(main))
            return
        if self._lane_streams is None:
            self._lane_streams = [torch.cuda.Stream(device=self.device) for _ in self.lane_divs]
        side = self._lane_streams
        streams = [main] + side

        def join():
            for st in side:
                main.wait_stream(st)

        def wait_all(st):
            for s_ in streams:
                st.wait_stream(s_)
        for st in side:
            st.wait_stream(main)
        try:
            for i in order:
                ln = self._lane_of[i]
                for j in deps[i]:
                    streams[ln].wait_event(events[j])
                self._lane = ln
                if ln:
                    with torch.cuda.stream(streams[ln]):
                        action(self.ops[i])
                else:
                    action(self.ops[i])
                self._lane = 0
                if i in events:
                    if events[i] is None:
                        events[i] = torch.cuda.Event()
                    events[i].record(streams[ln])
                if checkpoint is not None:
                    checkpoint(self.ops[i], wait_all)
        finally:
            self._lane = 0
            join()

    # ------------------------------------------------------------------------------------------ graph replay
    def _graph_ok(self):
        mode = os.environ.get("E2E_GRAPHS", "auto")
        if mode == "0" or self.grad_bucket_hook is not None or self.batch_dice_hook is not None:
            return False                     # (collectives inside the pass: issued eagerly)
        if torch.cuda.is_current_stream_capturing():
            return False                     # a caller is capturing the whole pass itself
        if mode == "1":
            return True
        if LANES and self.lane_divs:
            # measured (tools/scratch/small_bench.py): issued eagerly on two lanes + the weight-gradient stream a small plan
            # is shorter on the GPU than its single-stream graph replay (64^3 x 2 fwd+loss+bwd 11.4 vs 14.5 ms, Hippocampus
            # patch 9.1 vs 10.8) and the host still issues a step in less than half of that; graphs are for E2E_LANES=0
            return False
        vox = self.batch * self.patch[0] * self.patch[1] * self.patch[2]
        return vox <= int(os.environ.get("E2E_GRAPH_MAX_VOXELS", str(1 << 20)))

    def _run(self, key, fn):
        """fn() issues a fixed list of kernel launches over this plan's buffers.  Small plans replay it as a HIP graph: first
        call eager (lazy allocations), second call captured, later calls replayed; a new set of liveness tables drops the
        graphs (set_kernel_masks).  Weights, masks and inputs are read through stable pointers, so replays see their
        current contents."""
        if not self._graph_ok():
            return fn()
        key = key + (self.maps_generation,)
        g = self._graphs.get(key)
        if g is None:
            if key not in self._graph_seen:
                self._graph_seen.add(key)
                return fn()
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                fn()
            self._graphs[key] = g
        g.replay()

    # ------------------------------------------------------------------------------------------ backward
    def prepare_backward(self):
        if self._backward_ready:
            return
        for op in (self.ops[i] for i in self._bwd_order):      # backward order: first writer of a gradient buffer overwrites
            op.out.alloc_grad()
            op.plan_backward()
        # all parameter gradients live in ONE flat buffer (each tensor at a 256-byte aligned offset), laid out in the
        # order the backward pass completes them: the data-parallel all-reduce runs on it in place, and a prefix of the
        # buffer can be reduced while the rest of the backward pass is still running (grad_bucket_hook)
        order, seen = [], set()
        for op in (self.ops[i] for i in self._bwd_order):
            if isinstance(op, ConvOp):
                produced = [op.w_name, op.prefix + ".conv.bias", op.prefix + ".instnorm.weight", op.prefix + ".instnorm.bias"]
            elif isinstance(op, (UpOp, HeadOp)):
                produced = [op.w_name]
            else:
                produced = []
            for name in produced:
                if name in self.params and name not in seen:
                    order.append((name, op))
                    seen.add(name)
        order += [(name, None) for name in self.params if name not in seen]
        offs, total = {}, 0
        for name, _ in order:
            offs[name] = total
            total += (self.params[name].numel() + 63) // 64 * 64
        self.grad_flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        for name, p in self.params.items():
            self.grads[name] = self.grad_flat[offs[name]:offs[name] + p.numel()].view(p.shape)
        # bucket boundaries: after the op that completes roughly each quarter of the buffer
        self._bucket_after_op, lo, nb = {}, 0, 4
        for i, (name, op) in enumerate(order):
            end = offs[name] + (self.params[name].numel() + 63) // 64 * 64
            last_of_op = i + 1 == len(order) or order[i + 1][1] is not op
            if op is not None and last_of_op and (end - lo >= total // nb):
                self._bucket_after_op[id(op)] = (lo, end)
                lo = end
        self._bucket_tail = (lo, total)
        ws = max([op.wgrad_ws_bytes() for op in self.ops if hasattr(op, "wgrad_ws_bytes")] +
                 [lib().head1x1_wgrad_ws_bytes(self.batch, h.src.shape[1], h.k, h.src.spatial) for h in self.heads])
        self._wgrad_ws = [torch.empty((ws + 3) // 4, dtype=torch.float32, device=self.device) for _ in range(len(self.lane_divs) + 1)]
        cmax = max(op.cout for op in self.conv_ops.values())
        # (zeroed once: e2e_in_lrelu_bwd leaves its workspace ready for the next call)
        self._in_sums = [torch.zeros(int(lib().in_lrelu_bwd_ws_doubles(self.batch, cmax)), dtype=torch.float64, device=self.device)
                         for _ in range(len(self.lane_divs) + 1)]
        for op in self.conv_ops.values():       # records of the fused InstanceNorm-backward sums, laid out for the buffer's last writer
            d_, h_, w_ = op.out_dims
            lw = op.out.last_writer
            op.own_sums, op.own_np = None, 0
            if isinstance(lw, PoolOp) and FUSE_IN_SUMS >= 1:
                op.own_np = int(lib().maxpool_bwd_num_records(d_, h_, w_, *lw.kernel))
            elif isinstance(lw, ConvOp) and FUSE_IN_SUMS >= 2 and lw.sparse_ok:
                op.own_np = d_ * ((h_ + 15) // 16) * ((w_ + 31) // 32)          # the 16 x 32 tiles of conv133_sparse_kernel
            if op.own_np > 0:
                op.own_sums = torch.zeros(self.batch * op.cout * op.own_np * 2, dtype=torch.float64, device=self.device)
            if op.sp_bwd is not None:
                op.sp_bwd.table = None          # (who writes a buffer last is known only now)
        self._plan_lanes_backward()
        self._loss_buffers()
        self._backward_ready = True

    def backward(self, dlogits: Optional[Sequence[Optional[torch.Tensor]]] = None):
        """dlogits: gradients w.r.t. the 4 logits (None entries = zero).  When omitted the heads' grad buffers are
        expected to be filled already (loss_backward)."""
        self.prepare_backward()
        if dlogits is not None:
            for h, g in zip(self.heads, dlogits):
                if g is None:
                    h.out.grad.zero_()
                else:
                    h.out.grad.copy_(g)
        for op in self.conv_ops.values():       # (a pass that was abandoned between a writer and its producer)
            op.out.sums_ready = False
        self._refresh_weight_caches("b")        # (a no-op when the forward of this step packed both directions)
        hook = self.grad_bucket_hook
        main = torch.cuda.current_stream()
        if WGRAD_STREAM and not self._graph_ok() and not torch.cuda.is_current_stream_capturing():
            # weight gradients have no consumer inside the backward pass: they run on a second stream beside the data
            # gradient of the same layer (with their own workspace, in issue order among themselves).  Plans replayed as
            # HIP graphs stay single-stream: replaying a two-stream capture faults on ROCm 7.2.
            if self._wg_side is None:
                self._wg_side = torch.cuda.Stream(device=self.device)
                self.wgrad_ws_side = torch.empty_like(self._wgrad_ws[0])
            self._wg_active = self._wg_side
        side = self._wg_active

        def act(op):
            if isinstance(op, HeadOp) and not op.active:
                # inactive head (no deep supervision): its source still needs a defined gradient
                if op.acc == 0:
                    op.src.grad.zero_()
                self.grads[op.w_name].zero_()
            else:
                op.backward()

        def checkpoint(op, wait_all):
            if hook is not None and id(op) in self._bucket_after_op:
                # gradients in flat[lo:hi] are final once everything issued so far has run.  The collective is issued from a stream
                # of its own that waits for the caller's stream, the lanes and the weight-gradient stream -- none of THEM waits: until
                # round 5 the caller's stream joined all the others here, four times per pass, which cost 1.2 ms of lost overlap per
                # step even with one rank (bench rccl.allreduce_exposed_ms)
                if self._comm_stream is None:
                    self._comm_stream = torch.cuda.Stream(device=self.device)
                comm = self._comm_stream
                wait_all(comm)
                if side is not None:
                    comm.wait_stream(side)
                with torch.cuda.stream(comm):
                    hook(*self._bucket_after_op[id(op)])
        try:
            self._exec(self._bwd_order, self._deps_bwd, self._ev_bwd, act, checkpoint)
        finally:
            if side is not None:
                main.wait_stream(side)
                self._wg_active = None
            if self._comm_stream is not None:
                # whatever a bucket hook ENQUEUED on the communication stream (a synchronous collective, a scale, a copy) is
                # ordered in front of the caller's next kernel -- the optimizer step.  Free for the asynchronous RCCL path:
                # there the stream holds event waits only and finish() joins the work handles themselves.
                main.wait_stream(self._comm_stream)
        if hook is not None and self._bucket_tail[1] > self._bucket_tail[0]:
            hook(*self._bucket_tail)
        return self.grads

    def _loss_buffers(self):
        if self.loss_ws is None:
            self.loss_ws = torch.empty(lib().loss_ws_bytes(self.batch, self.cfg.num_classes) // 8, dtype=torch.float64,
                                       device=self.device)
            self.loss_val = torch.zeros(1, dtype=torch.float32, device=self.device)

    def _loss(self, targets, weights, batch_dice, smooth, with_grad):
        L = lib()
        self._loss_buffers()
        self.loss_val.zero_()
        k = self.cfg.num_classes
        for i, h in enumerate(self.heads):
            wgt = float(weights[i]) if i < len(weights) else 0.0
            if wgt == 0.0 or not h.active:
                if with_grad:
                    h.out.grad.zero_()
                continue
            t = targets[i]
            assert t.is_cuda and t.dtype == torch.float32 and t.numel() == self.batch * h.src.spatial
            L.dc_ce_reduce(h.out.data.data_ptr(), t.data_ptr(), self.loss_ws.data_ptr(), self.batch, k, h.src.spatial,
                           _stream())
            if batch_dice and self.batch_dice_hook is not None:
                # data-parallel batch dice (reference nnUNetTrainerV2_DDP.py:263-268): global tp/fp/fn over all ranks
                L.dc_ce_fold_batch(self.loss_ws.data_ptr(), self.batch, k, _stream())
                self.batch_dice_hook(self.loss_ws[:3 * k])
            L.dc_ce_grad(h.out.data.data_ptr(), t.data_ptr(), self.loss_ws.data_ptr(), wgt, 1 if batch_dice else 0,
                         smooth, h.out.grad.data_ptr() if with_grad else None, self.loss_val.data_ptr(), self.batch, k,
                         h.src.spatial, _stream())
        return self.loss_val

    def loss_backward(self, targets: Sequence[torch.Tensor], weights: Sequence[float], batch_dice=False, smooth=1e-5):
        """Deep-supervision Dice+CE loss (reference MultipleOutputLoss2(DC_and_CE_loss), deep_supervision.py:31-43)
        and the full backward pass.  targets[i]: [B,1,...] float labels at scale i.  Returns the device loss scalar."""
        self.prepare_backward()
        if not self._graph_ok():
            self._loss(targets, weights, batch_dice, smooth, True)
            self.backward(None)
            return self.loss_val
        # graph replay: the targets are staged into buffers of this plan (their pointers are part of the graph)
        if self._tgt_static is None:
            self._tgt_static = [torch.empty((self.batch, 1) + tuple(h.out.shape[2:]), dtype=torch.float32, device=self.device)
                                for h in self.heads]
        n = min(len(targets), len(self.heads))
        for i in range(n):
            self._tgt_static[i].copy_(targets[i].reshape(self._tgt_static[i].shape))
        wkey = tuple(float(weights[i]) if i < len(weights) else 0.0 for i in range(len(self.heads)))
        active = tuple(h.active for h in self.heads)

        def body():
            self._loss(self._tgt_static, weights, batch_dice, smooth, True)
            self.backward(None)
        self._run(("lossbwd", wkey, bool(batch_dice), float(smooth), active), body)
        return self.loss_val

    def loss_value(self, targets: Sequence[torch.Tensor], weights: Sequence[float], batch_dice=False, smooth=1e-5):
        """The same loss without gradients (validation batches, reference nnUNetTrainer_simple.py:980-988)."""
        return self._loss(targets, weights, batch_dice, smooth, False)

    def online_eval_counts(self, target: torch.Tensor) -> torch.Tensor:
        """Hard tp/fp/fn voxel counts [K, 3] (int64, device) of the full-resolution prediction against `target`
        (reference run_online_evaluation, nnUNetTrainer_simple.py:373-405)."""
        h = self.heads[0]
        k = self.cfg.num_classes
        if self._eval_counts is None:
            self._eval_counts = torch.zeros((k, 3), dtype=torch.int64, device=self.device)
        assert target.is_cuda and target.dtype == torch.float32 and target.numel() == self.batch * h.src.spatial
        lib().online_eval_counts(h.out.data.data_ptr(), target.data_ptr(), self._eval_counts.data_ptr(), self.batch, k,
                                 h.src.spatial, _stream())
        return self._eval_counts

    # ------------------------------------------------------------------------------------------ accounting
    def activation_bytes(self):
        seen, total = set(), 0
        for op in self.ops:
            for a in [op.out]:
                if id(a) not in seen:
                    seen.add(id(a))
                    total += a.data.numel() * 4 + (a.grad.numel() * 4 if a.grad is not None else 0)
        return total
