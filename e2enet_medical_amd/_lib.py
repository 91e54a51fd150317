"""ctypes binding of libe2e_hip.so (the C-ABI declared in include/e2e_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C e2enet_medical_amd/csrc``.
There is NO fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# E2E_LIB_PATH: A/B runs against a diagnostic build of the same sources (e.g. `make DEFS=-D... LIB=...`); still the HIP
# library, still no fallback
LIB_PATH = os.environ.get("E2E_LIB_PATH") or os.path.join(_HERE, "csrc", "libe2e_hip.so")

# every E2E_* environment variable the package, bench.py and the library read (INTEGRATION.md section 7 says what each does).
# Anything else spelled E2E_* in the environment is probably a typo or a knob that no longer exists -- but "E2E_" is also a
# common prefix of end-to-end test harness settings (E2E_BASE_URL ...), so the default is a warning that names the variable;
# E2E_STRICT_ENV=1 (benchmark and A/B scripts) turns it into a refusal to load the library.
KNOWN_ENV = frozenset({
    # library (csrc/*.hip)
    "E2E_CONV_MM", "E2E_MM_GRID", "E2E_MM_GEOM", "E2E_MM_PAIRQ", "E2E_CONV_DENSE", "E2E_CONV_SPARSE2", "E2E_CONV_PERSIST", "E2E_CONV_WGS",
    "E2E_CONV_KSPLIT", "E2E_WG_H2", "E2E_WG_BF3", "E2E_CT_BF3", "E2E_CT_H2",
    # diagnostic builds of the library only (-DE2E_CONV_DEBUG / -DMM_STAMPS); ignored by the shipped build
    "E2E_CONV_DBG", "E2E_MM_STAMPS",
    # host side
    "E2E_LIB_PATH", "E2E_STRICT_ENV", "E2E_DENSE_MIN_DENSITY", "E2E_MM_MIN_DENSITY", "E2E_WGRAD_STREAM", "E2E_LANES", "E2E_GRAPHS",
    "E2E_GRAPH_MAX_VOXELS", "E2E_PLAN_CACHE_GB", "E2E_SW_BLOCKING", "E2E_FORCE_DIST",
    # bench.py
    "E2E_CPU_THREADS", "E2E_BENCH_DRY", "E2E_BENCH_DRY_FAIL_RANK", "E2E_BENCH_ALL_LAUNCHES",
})


def check_env(environ=None, strict=None):
    """Report an E2E_* variable nobody reads (called when the library is loaded): a warning, or a RuntimeError under
    E2E_STRICT_ENV=1 / strict=True."""
    environ = os.environ if environ is None else environ
    unknown = sorted(k for k in environ if k.startswith("E2E_") and k not in KNOWN_ENV)
    if not unknown:
        return
    msg = ("unknown E2E_* environment variable(s) %s: not read by this build (known: %s)"
           % (", ".join(unknown), ", ".join(sorted(KNOWN_ENV))))
    if strict is None:
        strict = environ.get("E2E_STRICT_ENV") == "1"
    if strict:
        raise RuntimeError(msg)
    import warnings
    warnings.warn(msg + "; set E2E_STRICT_ENV=1 to refuse instead", RuntimeWarning, stacklevel=2)


class InChan(C.Structure):
    """e2e_in_chan_t"""
    _fields_ = [("ptr", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("nstride", C.c_longlong), ("ab_nstride", C.c_int), ("dshift", C.c_int),
                ("slope", C.c_float), ("reserved", C.c_int)]


class OutChan(C.Structure):
    """e2e_out_chan_t"""
    _fields_ = [("ptr", C.c_void_p), ("nstride", C.c_longlong), ("dshift", C.c_int), ("accumulate", C.c_int)]


class InSumChan(C.Structure):
    """e2e_in_sum_chan_t"""
    _fields_ = [("y", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("part", C.c_void_p), ("nstride", C.c_longlong), ("part_nstride", C.c_longlong), ("ab_nstride", C.c_int),
                ("slope", C.c_float)]


class ParamEntry(C.Structure):
    """e2e_param_t"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("momentum", C.c_void_p), ("mask", C.c_void_p),
                ("numel", C.c_longlong)]


class SparsePackJob(C.Structure):
    """e2e_sparse_pack_job_t"""
    _fields_ = [("w", C.c_void_p), ("wpk", C.c_void_p), ("qslot", C.c_void_p), ("pslot", C.c_void_p), ("quads", C.c_void_p),
                ("woff", C.c_void_p), ("groups", C.c_int), ("nchunks", C.c_int), ("wq_stride", C.c_int), ("wp_stride", C.c_int),
                ("reverse", C.c_int), ("kmax", C.c_int)]


class MmPackJob(C.Structure):
    """e2e_mm_pack_job_t"""
    _fields_ = [("w", C.c_void_p), ("quads", C.c_void_p), ("wpk", C.c_void_p), ("w_absmax", C.c_void_p), ("P", C.c_int), ("Q", C.c_int),
                ("wq_stride", C.c_int), ("wp_stride", C.c_int), ("reverse", C.c_int), ("owns_absmax", C.c_int)]


class RangeSrc(C.Structure):
    """e2e_range_src_t"""
    _fields_ = [("kind", C.c_int), ("C", C.c_int), ("N", C.c_longlong), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("w", C.c_void_p),
                ("wCout", C.c_int), ("wks", C.c_int), ("word", C.c_void_p)]


class RangeJob(C.Structure):
    """e2e_range_job_t"""
    _fields_ = [("src", RangeSrc * 3), ("out", C.c_void_p)]


assert C.sizeof(MmPackJob) == 56 and C.sizeof(RangeSrc) == 56 and C.sizeof(RangeJob) == 176
assert C.sizeof(InChan) == 48 and C.sizeof(OutChan) == 24 and C.sizeof(ParamEntry) == 40 and C.sizeof(SparsePackJob) == 72 and C.sizeof(InSumChan) == 72

P, I, F, LL = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> (restype, argtypes); every symbol declared in include/e2e_hip.h
SIGNATURES = {
    "e2e_last_error": (C.c_char_p, []),
    "e2e_abi_version": (I, []),
    "e2e_last_kernel": (C.c_char_p, []),
    "e2e_conv133_num_partials": (I, [I, I, I, I, I]),
    "e2e_conv133_fwd": (I, [P, I, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "e2e_conv133_fwd_ws_bytes": (LL, [I, I, I, I, I, I, I, I, I]),
    "e2e_conv133_fwd_splitk": (I, [P, I, P, P, P, P, P, I, I, I, I, I, I, I, I, P, LL, P]),
    "e2e_conv133_dgrad": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "e2e_conv133_dgrad_ws_bytes": (LL, [I, I, I, I, I, I, I, I, I]),
    "e2e_conv133_dgrad_splitk": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, P, LL, P]),
    "e2e_conv133_dense_ws_bytes": (LL, [I, I, I, I, I, I, I, I, I]),
    "e2e_conv133_sparse_eligible": (I, [I, I, I, I, I, I, I, I]),
    "e2e_conv133_sparse_wpk_floats": (LL, [I, I, I]),
    "e2e_conv133_sparse_plan": (I, [P, I, I, I, P, P, P, P, P, P]),
    "e2e_conv133_sparse_pack": (I, [P, I, LL, P]),
    "e2e_conv133_fwd_sparse": (I, [P, I, P, P, P, P, I, P, I, P, P, I, I, I, I, I, P]),
    "e2e_conv133_dgrad_sparse": (I, [P, P, P, P, I, P, P, P, I, I, I, I, I, I, I, P]),
    "e2e_conv133_fwd_dense": (I, [P, I, P, P, P, P, P, I, I, I, I, I, P, LL, P]),
    "e2e_conv133_mm_ws_bytes": (LL, [I, I, I, I, I, I, I, I, I]),
    "e2e_conv133_mm_pack": (I, [P, I, LL, P]),
    "e2e_conv133_input_ranges_ws_bytes": (LL, [I]),
    "e2e_conv133_input_ranges": (I, [P, I, P, P]),
    "e2e_absmax_word": (I, [P, LL, P, P]),
    "e2e_conv133_fwd_mm": (I, [P, I, P, P, P, P, P, P, I, I, I, I, I, P]),
    "e2e_conv133_dgrad_mm": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "e2e_conv133_dgrad_dense": (I, [P, P, P, P, I, I, I, I, I, I, P, LL, P]),
    "e2e_conv133_wgrad_ws_bytes": (LL, [I, I, I, I, I, I, I, I, I]),
    "e2e_conv133_wgrad": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, P, P, P]),
    "e2e_diag_split_gemm": (I, [P, P, P, I, I, P, P]),
    "e2e_diag_kernel_clock": (I, [I, P, P, I]),
    "e2e_in_stats_finalize": (I, [P, I, P, P, F, P, P, P, P, I, I, P]),
    "e2e_in_lrelu_bwd_ws_doubles": (LL, [I, I]),
    "e2e_in_lrelu_bwd": (I, [P, P, P, P, P, P, P, F, P, P, P, P, I, I, LL, P, I, P, P]),
    "e2e_convT_fwd": (I, [P, P, P, F, P, P, P, I, I, I, I, I, I, I, I, I, P, P, P]),
    "e2e_convT_dgrad": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, P, P, P, P]),
    "e2e_convT_wgrad_ws_bytes": (LL, [I, I, I, I, I, I, I, I, I]),
    "e2e_convT_wgrad": (I, [P, P, P, F, P, P, P, I, I, I, I, I, I, I, I, I, P, P, P, P]),
    "e2e_maxpool_fwd": (I, [P, P, P, F, P, I, I, I, I, I, I, I, I, P]),
    "e2e_maxpool_bwd_num_records": (I, [I, I, I, I, I, I]),
    "e2e_maxpool_bwd": (I, [P, P, P, F, P, P, I, I, I, I, I, I, I, I, I, P, P, P, P]),
    "e2e_head1x1_fwd": (I, [P, P, P, F, P, P, I, I, I, LL, P]),
    "e2e_head1x1_dgrad": (I, [P, P, P, I, I, I, I, LL, P]),
    "e2e_head1x1_wgrad_ws_bytes": (LL, [I, I, I, LL]),
    "e2e_head1x1_wgrad": (I, [P, P, P, F, P, P, P, I, I, I, LL, P]),
    "e2e_loss_ws_bytes": (LL, [I, I]),
    "e2e_dc_ce_reduce": (I, [P, P, P, I, I, LL, P]),
    "e2e_dc_ce_grad": (I, [P, P, P, F, I, F, P, P, I, I, LL, P]),
    "e2e_dc_ce_fold_batch": (I, [P, I, I, P]),
    "e2e_online_eval_counts": (I, [P, P, P, I, I, LL, P]),
    "e2e_ds_target_gather": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "e2e_grad_sqnorm": (I, [P, I, P, P]),
    "e2e_sgd_clip_mask_step": (I, [P, I, P, F, F, F, F, I, I, P]),
    "e2e_apply_mask": (I, [P, I, P]),
    "e2e_dsff_kernel_l1": (I, [P, P, I, I, I, I, I, P]),
    "e2e_dsff_kth_value": (I, [P, I, I, P, P, P]),
    "e2e_dsff_death": (I, [P, P, P, I, P]),
    "e2e_dsff_grad_score": (I, [P, P, F, P, P, I, I, I, I, I, P]),
    "e2e_dsff_grow_above": (I, [P, P, P, I, I, I, P]),
    "e2e_dsff_expand": (I, [P, P, P, P, I, I, I, P]),
    "e2e_dsff_expand_quads": (I, [P, P, P, I, I, P]),
    "e2e_dsff_kmask_from_weights": (I, [P, P, I, I, I, P]),
    "e2e_flip3d": (I, [P, P, I, I, I, I, I, P]),
    "e2e_softmax_flip_acc": (I, [P, P, F, I, I, I, I, I, I, P]),
    "e2e_nonlin_flip_acc": (I, [P, P, F, I, I, I, I, I, I, I, P]),
    "e2e_sw_accumulate": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, P]),
    "e2e_sw_finalize_argmax": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, P]),
    "e2e_ensemble_accumulate": (I, [P, P, LL, I, I, P]),
    "e2e_export_argmax_u8": (I, [P, P, I, LL, I, I, I, LL, LL, LL, I, I, I, I, I, I, P, I, P]),
    "e2e_resample_linear": (I, [P, P, I, LL, I, I, I, LL, LL, LL, I, I, I, I, P]),
    "e2e_aug_spatial": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, F, P]),
    "e2e_aug_bspline_prefilter_axis": (I, [P, P, I, I, I, I, I, P]),
    "e2e_aug_stats_ws_bytes": (LL, [I]),
    "e2e_aug_stats": (I, [P, P, P, I, LL, P]),
    "e2e_aug_pointwise": (I, [P, P, I, I, LL, C.c_ulonglong, P]),
    "e2e_aug_blur_axis": (I, [P, P, P, I, I, I, I, I, P]),
    "e2e_aug_lowres": (I, [P, P, P, I, I, I, I, P]),
    "e2e_aug_lowres_down": (I, [P, P, I, I, I, I, I, I, I, P]),
    "e2e_aug_lowres_up3": (I, [P, P, P, I, I, I, I, I, I, I, P]),
    "e2e_aug_finish": (I, [P, P, P, I, I, I, LL, P]),
}

_NO_STATUS = {"e2e_last_error", "e2e_abi_version", "e2e_last_kernel", "e2e_conv133_num_partials", "e2e_conv133_wgrad_ws_bytes", "e2e_conv133_dense_ws_bytes", "e2e_conv133_mm_ws_bytes", "e2e_conv133_sparse_eligible", "e2e_conv133_sparse_wpk_floats", "e2e_maxpool_bwd_num_records", "e2e_conv133_fwd_ws_bytes", "e2e_conv133_dgrad_ws_bytes",
              "e2e_convT_wgrad_ws_bytes", "e2e_conv133_input_ranges_ws_bytes", "e2e_in_lrelu_bwd_ws_doubles", "e2e_head1x1_wgrad_ws_bytes", "e2e_loss_ws_bytes", "e2e_aug_stats_ws_bytes"}


class E2EError(RuntimeError):
    pass


class _Lib:
    def __init__(self, path):
        if not os.path.exists(path):
            raise ImportError(
                "libe2e_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C e2enet_medical_amd/csrc`. There is no CPU fallback." % path)
        self._dll = C.CDLL(path)
        self.path = path
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self._dll, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
            if name in _NO_STATUS:
                setattr(self, name[4:], fn)
            else:
                setattr(self, name[4:], self._checked(name, fn))

    def _checked(self, name, fn):
        last_error = self._dll.e2e_last_error

        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise E2EError("%s failed (%d): %s" % (name, rc, (last_error() or b"").decode()))
        call.__name__ = name
        return call


_lib = None


ABI_VERSION = 19          # e2e_abi_version() of the library this binding was written against


def lib() -> _Lib:
    """The loaded library (loads on first use; raises ImportError when it has not been built)."""
    global _lib
    if _lib is None:
        # torch first: libe2e_hip.so must bind to the HIP runtime PyTorch-ROCm has loaded (one runtime per process,
        # so that torch's streams and device pointers are valid in our launches)
        import torch  # noqa: F401
        check_env()
        handle = _Lib(LIB_PATH)
        got = handle.abi_version()
        if got != ABI_VERSION:           # a stale in-tree .so from another revision: fail loudly, never guess
            raise ImportError("%s reports ABI version %d, this package needs %d: rebuild with `make -C %s`"
                              % (LIB_PATH, got, ABI_VERSION, os.path.dirname(LIB_PATH)))
        _lib = handle
    return _lib
