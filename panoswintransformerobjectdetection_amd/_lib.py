"""ctypes binding of libpswin_hip.so (include/pswin.h).  There is no fallback: if the library is missing
or a call fails, this raises.  PyTorch is used only for device memory and the current HIP stream."""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSWIN_LIB") or os.path.join(_PKG, "libpswin_hip.so")   # PSWIN_LIB: another build (A/B runs)

F32, BF16 = 0, 1
MODE_PLANAR, MODE_PANO = 0, 1
WS, WTOK, WPAD, HEAD_DIM = 7, 49, 64, 32
ABI_VERSION = 2

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_ip = ctypes.POINTER(ctypes.c_int)

# name -> argtypes; every function returns int.  Mirrors include/pswin.h one to one.
_PROTOTYPES = {
    "pswin_version": [],
    "pswin_window_grid": [_i, _i, _i, _ip, _ip, _ip],
    "pswin_window_map": [_i, _i, _i, _i, _vp, _vp, _vp],
    "pswin_planar_mask": [_i, _i, _i, _vp, _vp],
    "pswin_uv_grid": [_i, _i, _vp, _vp],
    "pswin_abs_pos_features": [_vp, _i, _vp, _vp],
    "pswin_gather_uv": [_vp, _vp, _i, _vp, _vp],
    "pswin_haversine_windows": [_vp, _vp, _i, _vp, _vp],
    "pswin_window_gather": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "pswin_window_scatter_add": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "pswin_ln_gather_fwd": [_vp, _i, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "pswin_ln_gather_fwd_add": [_vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "pswin_ln_gather_bwd": [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "pswin_ln_gather_bwd_ex": [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i,
                               _vp],
    "pswin_scatter_add_ln_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "pswin_scatter_add_ln_fwd_map": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp],
    "pswin_ln_nchw_supported": [_i, _i],
    "pswin_ln_nchw_fwd": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    "pswin_ln_nchw_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "pswin_scatter_add_ln_nchw_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    "pswin_ln_nchw_bwd_ex": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "pswin_ln_workspace": [ctypes.c_longlong, _i],
    "pswin_ln_patch_merge_fwd": [_vp, _i, _vp, _vp, _f, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "pswin_ln_patch_merge_bwd": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "pswin_patch_merge_gather": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "pswin_patch_merge_scatter": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "pswin_bn_workspace": [_i],
    "pswin_bn_relu_fwd": [_vp, _i, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _vp],
    "pswin_bn_relu_bwd": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _vp],
    "pswin_stem_workspace": [_i, _i, _i],
    "pswin_stem_pack_input": [_vp, _i, _i, _i, _vp, _vp],
    "pswin_stem_conv1_stats": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "pswin_stem_conv2_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pswin_stem_conv3_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "pswin_stem_conv3_bwd_stats": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "pswin_stem_conv3_bwd_data": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "pswin_stem_conv3_wgrad": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pswin_stem_conv2_wgrad": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pswin_stem_conv2_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "pswin_stem_pack_weights": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pswin_stem_bn_fold": [_vp, _vp, ctypes.c_double, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _i, _vp, _vp, _vp],
    "pswin_stem_bn2_coefs": [_vp, _vp, ctypes.c_double, _i, _vp, _vp],
    "pswin_stem_conv1_wgrad": [_vp, _vp, _vp, _vp, ctypes.c_double, _i, _vp, _vp, _vp],
    "pswin_gemm_tn_ring_supported": [ctypes.c_longlong, _i, _i],
    "pswin_gemm_tn_ring_splits": [ctypes.c_longlong, _i, _i, _i],
    "pswin_gemm_tn_ring": [_vp, _vp, _vp, _i, ctypes.c_longlong, _i, _i, _i, _vp],
    "pswin_gemm_tn_ring_bias": [_vp, _vp, _vp, _i, _vp, _i, _i, ctypes.c_longlong, _i, _i, _i, _vp],
    "pswin_gemm_tn_ring_jobs": [_vp, _i, _vp],
    "pswin_transpose_jobs": [_vp, _i, _vp],
    "pswin_adamw_flat": [_vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp, _vp],
    "pswin_adamw_flat_groups": [_vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _vp, _i, _vp, _vp, ctypes.c_double, ctypes.c_double,
                                ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp, _vp],
    "pswin_roi_align_supported": [_i, _i],
    "pswin_roi_align_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "pswin_roi_align_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "pswin_nms_workspace": [_i, _i],
    "pswin_nms_groups": [_vp, _vp, _i, _i, ctypes.c_float, _vp, _vp, _vp],
    "pswin_gemm_nt_supported": [ctypes.c_longlong, _i, _i],
    "pswin_gemm_nt": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _vp],
    "pswin_gemm_nt_f32": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _vp],
    "pswin_gemm_nt_gelu_fwd": [_vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _vp],
    "pswin_gemm_nt_partial_rows": [ctypes.c_longlong, _i],
    "pswin_gemm_nt_gelu_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _vp],
    "pswin_gemm_skinny_supported": [_i, _i],
    "pswin_gemm_skinny": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _vp],
    "pswin_fc1_gelu_supported": [_i, _i],
    "pswin_fc1_gelu_fwd": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp],
    "pswin_fc1_gelu_workspace": [_i],
    "pswin_mlp0_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp],
    "pswin_mlp0_bwd_supported": [_i, _i],
    "pswin_mlp0_bwd_partial_rows": [ctypes.c_longlong],
    "pswin_mlp0_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp],
    "pswin_fc1_gelu_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp],
    "pswin_bias_gelu_fwd": [_vp, _i, _vp, _vp, ctypes.c_longlong, _i, _vp],
    "pswin_bias_gelu_workspace": [ctypes.c_longlong, _i],
    "pswin_bias_gelu_tune": [_i, _i],
    "pswin_bias_gelu_bwd": [_vp, _vp, _i, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _vp],
    "pswin_colsum_workspace": [ctypes.c_longlong, _i, _i],
    "pswin_colsum": [_vp, _i, ctypes.c_longlong, _i, _vp, _vp, _vp],
    "pswin_colsum_skip": [_vp, _i, ctypes.c_longlong, _i, _i, _i, _vp, _vp],
    "pswin_reduce_jobs": [_vp, _i, _vp],
    "pswin_attn_table_grads_batch": [_vp, _i, _i, _vp],
    "pswin_attn_table_grads_partial_rows": [_i, _i],
    "pswin_ln_partial_rows": [ctypes.c_longlong, _i],
    "pswin_bias_gelu_partial_rows": [ctypes.c_longlong, _i, _i],
    "pswin_fc1_gelu_partial_rows": [ctypes.c_longlong],
    "pswin_interp_rows": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "pswin_interp_rows_adjoint": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "pswin_attn_pad_tiles": [_vp, _i, _i, _vp, _vp],
    "pswin_attn_fwd": [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "pswin_win_attn_fused_supported": [_i, _i, _i],
    "pswin_win_attn_fused_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i,
                                 _f, _i, _vp],
    "pswin_qkv_attn_fused_supported": [_i, _i, _i],
    "pswin_qkv_attn_fused_fwd": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _f, _i, _vp],
    "pswin_attn_suggest_chunks": [_i, _i, _i, _i],
    "pswin_attn_bwd": [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp,
                       _i, _i, _i, _i, _f, _i, _vp],
    "pswin_attn_bwd_ex": [_vp, _vp, _vp, _i, ctypes.c_longlong, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp,
                          _i, _i, _i, _i, _f, _i, _vp],
    "pswin_attn_table_grads_workspace": [_i],
    "pswin_attn_table_grads": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp],
}

_lib = None


class PswinError(RuntimeError):
    pass


class ReduceJob(ctypes.Structure):
    """pswin_reduce_job of include/pswin.h"""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("dtype", ctypes.c_int), ("rows", ctypes.c_int),
                ("cols", ctypes.c_int), ("ld", ctypes.c_int)]


class TnJob(ctypes.Structure):
    """pswin_tn_job of include/pswin.h"""
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("partial", ctypes.c_void_p), ("dbias_partial", ctypes.c_void_p),
                ("M", ctypes.c_longlong), ("N", ctypes.c_int), ("K", ctypes.c_int), ("splits", ctypes.c_int), ("partial_dtype", ctypes.c_int),
                ("zero_lo", ctypes.c_int), ("zero_hi", ctypes.c_int)]


class TransposeJob(ctypes.Structure):
    """pswin_transpose_job of include/pswin.h"""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int)]


class RoiLevels(ctypes.Structure):
    """pswin_roi_levels of include/pswin.h"""
    _fields_ = [("feat", ctypes.c_void_p * 4), ("dfeat", ctypes.c_void_p * 4), ("H", ctypes.c_int * 4), ("W", ctypes.c_int * 4),
                ("spatial_scale", ctypes.c_float * 4), ("n_levels", ctypes.c_int)]


class TableGradJob(ctypes.Structure):
    """pswin_table_grad_job of include/pswin.h"""
    _fields_ = [("dscore_sum", ctypes.c_void_p), ("dist_tiles_t", ctypes.c_void_p), ("dalpha", ctypes.c_void_p),
                ("dbeta", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("n_tiles", ctypes.c_int),
                ("n_bias_windows", ctypes.c_int), ("n_dist", ctypes.c_int), ("heads", ctypes.c_int)]


def exported_symbols():
    return sorted(_PROTOTYPES)


def load():
    """Load the library once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise PswinError(
            f"{LIB_PATH} is missing: build it with `python -m panoswintransformerobjectdetection_amd.build` "
            "(hipcc, gfx950).  This package has no CPU or PyTorch fallback for its kernels.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    v = lib.pswin_version()
    if v != ABI_VERSION:
        raise PswinError(f"libpswin_hip.so ABI {v} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported configuration"}.get(rc, f"hipError_t {rc}")
        raise PswinError(f"{what} failed: {kind}")


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise PswinError(f"unsupported dtype {t.dtype}: the kernels take float32 or bfloat16")


def ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_of(t):
    if not t.is_cuda:
        raise PswinError("the PanoSwin kernels run on an MI355X (HIP) device only; got a CPU tensor")
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


# Optional per-launch timing (bench.py): name -> list of (start_event, end_event, algorithmic_bytes, algorithmic_flops, partial_bytes);
# partial_bytes = bytes of implementation artefacts a launch writes beside its algorithmic output (split partial slabs): reported, never
# counted as algorithmic traffic.
# The events are recorded on the stream the kernel is launched on (torch's current stream), around that launch only.
_TIMED = None


def enable_timing(names):
    """Start collecting HIP-event pairs around every launch of the named entry points (and `timed` regions)."""
    global _TIMED
    _TIMED = {n: [] for n in names}


def disable_timing():
    """Stop collecting; returns {name: [(milliseconds, algorithmic_bytes, algorithmic_flops, partial_bytes), ...]} (synchronises)."""
    global _TIMED
    rec, _TIMED = _TIMED, None
    if rec is None:
        return {}
    torch.cuda.synchronize()
    return {n: [(s.elapsed_time(e), b, f, pb) for s, e, b, f, pb in lst] for n, lst in rec.items()}


class timed:
    """with timed("lib_gemm_fwd", bytes, flops): ...   -- the same event bracket for launches that do not go through
    `call` (the library GEMMs issued through PyTorch).  Free when timing is off."""

    def __init__(self, name, algo_bytes=0, algo_flops=0):
        self.rec = _TIMED.get(name) if _TIMED is not None else None
        self.b, self.f = algo_bytes, algo_flops

    def __enter__(self):
        if self.rec is not None:
            self.s, self.e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            self.e.record()
            self.rec.append((self.s, self.e, self.b, self.f, 0))
        return False


def call(name, ref_tensor, *args, algo_bytes=0, algo_flops=0, timed_as=None, partial_bytes=0):
    """Invoke an entry point on the current stream of ref_tensor's device and raise on a non-zero status.
    timed_as: the row of the per-kernel timing table the launch is booked under (default: its own name)."""
    lib = load()
    if not ref_tensor.is_cuda:
        raise PswinError(f"{name}: the PanoSwin kernels run on an MI355X (HIP) device only; got a CPU tensor")
    row = timed_as or name
    with torch.cuda.device(ref_tensor.device):
        if _TIMED is not None and row in _TIMED:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = getattr(lib, name)(*args, stream_of(ref_tensor))
            e.record()
            _TIMED[row].append((s, e, algo_bytes, algo_flops, partial_bytes))
        else:
            rc = getattr(lib, name)(*args, stream_of(ref_tensor))
    check(rc, name)


def window_grid(mode, H, W):
    hp, wp, nw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    check(load().pswin_window_grid(mode, H, W, ctypes.byref(hp), ctypes.byref(wp), ctypes.byref(nw)), "pswin_window_grid")
    return hp.value, wp.value, nw.value
