"""ctypes binding of csrc/libembrace_hip.so (the C ABI declared in include/embrace_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, this raises.
The library is loaded lazily so that model objects stay picklable and importable on a machine
without a GPU (the reference's harness pickles whole models: training_models_multimodal.py:413).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libembrace_hip.so")

EMB_F32, EMB_BF16, EMB_F64 = 0, 1, 2
CODE_IDX, CODE_ACTIVE = 1, 2
STATUS_INVALID_DISTRIBUTION = 1

DTYPE_CODE = {torch.float32: EMB_F32, torch.bfloat16: EMB_BF16, torch.float64: EMB_F64}
PARAM_DTYPE = {torch.float32: torch.float32, torch.bfloat16: torch.float32, torch.float64: torch.float64}

_vp, _i, _i64, _u64, _f, _d = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64,
                                ctypes.c_float, ctypes.c_double)

# name -> argtypes; must list every symbol include/embrace_hip.h declares (tests check this)
SIGNATURES = {
    "emb_abi_version": [],
    "emb_last_error": [],
    "emb_select_prep": [_vp, _i, _vp, _i, _u64, _u64, _vp, _i64, _vp, _vp, _i, _vp],
    "emb_embrace_fwd": [_vp] * 8 + [_u64, _u64, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "emb_embrace_fwd_select": [_vp] * 7 + [_i, _vp, _i, _vp, _vp, _u64, _u64, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "emb_embrace_bwd": [_vp] * 12 + [_vp, _i64, _i, _i, _i, _i, _i, _vp],
    "emb_embrace_bypass_fwd": [_vp] * 4 + [_i, _vp, _i, _vp, _vp, _u64, _u64, _vp, _i64, _vp, _vp, _i, _i, _i, _vp],
    "emb_embrace_bypass_bwd": [_vp] * 4 + [_i, _i, _i, _vp],
    "emb_select_prep_m": [_vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "emb_embrace_select_fwd": [_vp, _i, _vp, _vp, _u64, _u64, _vp, _i64, _vp, _vp, _i, _i, _i, _vp],
    "emb_embrace_select_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "emb_linear_fwd": [_vp] * 5 + [_i, _f, _i, _u64, _u64, _vp, _i64, _i, _i, _i, _i, _vp],
    "emb_linear_bwd": [_vp] * 7 + [_i, _f, _vp, _i64, _i, _i, _i, _i, _vp],
    "emb_weighted_ce": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "emb_count_labels": [_vp, _vp, _i, _vp],
    "emb_adam_step": [_vp] * 5 + [_i64, _d, _d, _d, _d, _d, _u64, _vp, _i, _vp],
    "emb_rmsprop_step": [_vp] * 4 + [_i64, _d, _d, _d, _d, _i, _vp],
    "emb_nadam_step": [_vp] * 6 + [_i64, _d, _d, _d, _d, _d, _d, _u64, _vp, _i, _vp],
    "emb_mlp_supported": [_i, _vp, _i, _i],
    "emb_mlp_workspace_bytes": [_i, _vp, _i, _i, _i],
    "emb_mlp_fwd": [_vp] * 9 + [_i, _i, _i, _u64, _u64, _vp, _i64, _i, _vp],
    "emb_rider_defer": [_vp, _i],
    "emb_rider_flush": [_vp],
    "emb_mlp_bwd": [_vp] * 11 + [_i, _i, _i, _vp, _i64, _i, _vp],
    "emb_adam_step_multi": [_vp] * 6 + [_i, _d, _d, _d, _d, _d, _u64, _vp, _i, _vp],
    "emb_rmsprop_step_multi": [_vp] * 5 + [_i, _d, _d, _d, _d, _i, _vp],
    "emb_nadam_step_multi": [_vp] * 7 + [_i, _d, _d, _d, _d, _d, _d, _u64, _vp, _i, _vp],
    "emb_convblock_workspace_bytes": [_i, _i, _i, _i, _i, _i],
    "emb_ncl_to_nlc": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "emb_conv_pack_weight": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "emb_convblock_fwd": [_vp] * 7 + [_i, _d, _d, _f, _u64, _u64, _vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _i64, _vp,
                                     _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "emb_convblock_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i] + [_vp] * 7 + [_i64, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "emb_conv_pack_register": [_vp, _vp, _vp, _i, _i, _i, _i, _i],
    "emb_conv_pack_unregister": [_vp],
    "emb_head_ce_supported": [_i, _i, _i],
    "emb_head_ce_workspace_bytes": [_i, _i],
    "emb_head_ce": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _i, _vp],
    "emb_head_ce_masked": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _i, _vp],
    "emb_embrace_premask": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "emb_embrace_bwd_masked_supported": [_i, _i, _i, _i, _i],
    "emb_embrace_bwd_masked": [_vp] * 13 + [_i64, _i, _i, _i, _i, _i, _vp],
    "emb_head_ce_finish": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "emb_gather_rows": [_vp, _vp, _vp, _i, _vp, _i64, _i64, _vp],
    "emb_mt19937_shuffle": [_vp, _vp, _vp, _i64],
    "emb_reduce_defer": [_vp, _i],
    "emb_reduce_flush": [_vp],
    "emb_copy_park": [_vp, _vp, _vp, _i],
    "emb_copy_flush": [_vp],
    "emb_parked_count": [_vp, _i],
    "emb_reset": [],
    "emb_reset_stream": [_vp],
    "emb_convblock_needs_y": [_i, _i, _i, _i, _i, _i],
    "emb_convblock_stats_elems": [_i, _i, _i, _i, _i, _i],
    "emb_convblock_first_linear": [_i],
    "emb_cast": [_vp, _i, _vp, _i, _i64, _vp],
    "emb_counter_add": [_vp, _u64, _vp],
}

_lib = None


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into csrc/libembrace_hip.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libembrace_hip.so failed")
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is the only implementation of this path "
                "(no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = {"emb_last_error": ctypes.c_char_p,
                          "emb_convblock_workspace_bytes": ctypes.c_int64,
                          "emb_convblock_stats_elems": ctypes.c_int64,
                          "emb_mlp_workspace_bytes": ctypes.c_int64,
                          "emb_head_ce_workspace_bytes": ctypes.c_int64}.get(name, ctypes.c_int)
        if L.emb_abi_version() != 2:
            raise RuntimeError("libembrace_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().emb_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device (what every launch of this package goes to).  The raw
    accessor skips ~20 us of Python per call (torch.cuda.current_stream() re-checks availability and builds an object)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("embracenet_amd: tensors must live on a ROCm device (no CPU fallback); "
                               f"got device {t.device}")
