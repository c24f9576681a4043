"""ctypes binding of libsemabs_hip.so (the C ABI in include/semabs.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.  Nothing
on the product path routes through `oracle/` or a torch/CPU re-implementation.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsemabs_hip.so")
if os.environ.get("SEMABS_LIB_PATH"):                # debugging: an explicit build of the same sources
    LIB_PATH = os.environ["SEMABS_LIB_PATH"]
elif os.environ.get("SEMABS_TUNE_LIB") == "1":        # tools/ only: the -DSEMABS_TUNING build of the same sources (build.py --tuning)
    LIB_PATH = os.path.join(_HERE, "lib", "libsemabs_hip_tune.so")

_lib = None

P = C.c_void_p
I = C.c_int
L = C.c_long
F = C.c_float
D = C.c_double

# name -> argtypes (all return int).  Kept in the order of include/semabs.h.
SIGNATURES = {
    "semabs_abi_version": [],
    "semabs_stream_create_cumask": [P, I, P],
    "semabs_stream_destroy": [P],
    "semabs_stream_cu_count": [P, P],
    "semabs_fill_u32": [P, C.c_longlong, C.c_uint, P],
    "semabs_replicate": [P, P, C.c_longlong, I, P],
    "semabs_copy2d": [P, C.c_longlong, P, C.c_longlong, C.c_longlong, C.c_longlong, P],
    "semabs_poison_empty": [P, P, C.c_longlong, P, C.c_longlong, P],
    "semabs_device_info": [C.c_char_p, I, C.POINTER(I), C.POINTER(C.c_longlong)],
    # geometry.hip
    "semabs_pointcloud": [P, I, I, P, I, P, P, P],
    "semabs_pointcloud_f64": [P, I, I, P, I, P, P],
    "semabs_voxel_index": [P, L, C.POINTER(F), C.POINTER(F), C.POINTER(I), P, P, P],
    "semabs_tsdf_integrate": [P, P, I, I, P, C.POINTER(F), C.POINTER(F), C.POINTER(I), P, P, P, P, P],
    "semabs_frustum_mask": [P, L, P, I, I, P, P],
    "semabs_compact_subsample": [P, L, C.c_ulonglong, L, P, P, P, P],
    "semabs_gather_point_features": [P, P, P, I, L, L, F, I, P, P, P],
    "semabs_ovssc_labels": [P, P, P, I, L, F, P, P],
    # tiles.hip
    "semabs_resize_coeffs": [I, I, P, P, I, P],
    "semabs_tile_patches": [P, I, I, I, P, I, P, P, P, P, P, I, I, I, P],
    "semabs_patchify": [P, P, I, I, I, P],
    "semabs_aggregate": [P, P, I, I, I, I, I, P, I, I, I, P, P],
    "semabs_unflip_average": [P, P, P, L, I, P],
    "semabs_unflip_average_rows": [P, P, P, I, L, L, I, P],
    "semabs_color_jitter": [P, I, I, C.POINTER(I), C.POINTER(F), P, P],
    "semabs_color_jitter_op": [P, I, I, I, F, P, P],
    # gemm.hip
    "semabs_gemm_f16": [P, P, P, P, P, L, I, I, L, I, L, I, C.POINTER(I), P],
    "semabs_gemm_f16_ex": [P, P, P, P, P, L, I, I, L, I, L, I, C.POINTER(I), I, P, P, P],
    "semabs_gemm_f16_ln": [P, P, P, P, L, I, I, L, I, L, I, P, P, P, P, P, P, P, I, L, I, P, P, P],
    "semabs_ln_rowstats": [P, L, I, I, F, P, P, P, P],
    # vit.hip
    "semabs_layernorm": [P, P, P, P, L, I, F, I, L, P, P],
    "semabs_layernorm2": [P, P, P, P, P, P, P, L, I, F, I, P, P],
    "semabs_add_layernorm": [P, P, P, P, P, L, I, F, P],
    "semabs_embed_finish": [P, P, P, I, I, I, P],
    "semabs_attention": [P, P, P, I, I, I, I, I, I, P],
    "semabs_attention_split": [P, P, P, P, I, I, I, I, I, I, I, P],
    "semabs_attention_cls": [P, P, P, P, P, I, I, I, I, I, P],
    "semabs_cls_scores": [P, P, P, P, L, I, P, I, I, I, P],
    "semabs_rows_gather": [P, P, L, I, L, L, P],
    "semabs_eot_rows_gather": [P, P, P, I, I, I, P],
    "semabs_quickgelu": [P, P, L, I, P],
    "semabs_quickgelu_grad": [P, P, L, P],
    "semabs_logit_grad": [P, P, I, I, I, P, P, P, I, P],
    "semabs_ln_bwd": [P, P, P, P, P, P, L, I, I, L, F, I, P],
    "semabs_gelu_bwd": [P, P, P, L, I, I, I, P],
    "semabs_rollout": [P, P, P, P, P, I, I, I, I, I, L, L, P],
    "semabs_gather_text": [P, P, P, P, I, I, I, P],
    # vitl.hip (multi-layer rollout: attention backward)
    "semabs_attention_bwd": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "semabs_seq_rescale": [P, P, P, L, L, P],
    "semabs_rollout_step": [P, P, L, P],
    "semabs_text_finish": [P, P, I, I, I, P],
    # unet.hip
    "semabs_point_mlp": [P, P, P, P, P, P, P, P, P, I, L, I, I, P],
    "semabs_point_mlp_fma": [P, P, P, P, P, P, P, P, P, I, L, I, I, P],
    "semabs_scatter_mean": [P, P, P, P, P, I, L, I, L, I, P],
    "semabs_scatter_mean_stats": [P, P, P, P, P, I, L, I, L, I, P, P],
    "semabs_scatter_mean_sparse": [P, P, P, P, P, P, I, L, I, L, I, P, P],
    "semabs_gn_stats": [P, P, I, L, I, I, I, P],
    "semabs_gn_finalize": [P, P, P, P, P, I, I, I, L, F, P],
    "semabs_conv3d": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "semabs_conv3d_stats": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P, I, P],
    "semabs_conv3d_sparse_stats": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P, I, P],
    "semabs_convtranspose3d": [P, P, P, C.POINTER(C.c_long), P, P, P, I, I, I, I, I, I, I, P],
    "semabs_convtranspose3d_stats": [P, P, P, C.POINTER(C.c_long), P, P, P, I, I, I, I, I, I, I, P, I, P],
    "semabs_maxpool3d": [P, P, I, I, I, I, I, I, P],
    "semabs_vool_head": [P, P, P, P, P, C.POINTER(F), C.POINTER(F), C.POINTER(I), F, I, L, I, P, P],
    "semabs_lamb_step": [P, I, P, I, D, D, D, D, D, I, P, P, P],
    "semabs_decoder": [P, P, C.POINTER(F), C.POINTER(F), C.POINTER(I), P, P, P, P, I, I, L, L, I, P, C.POINTER(I), P, P, P],
    # training step (train.hip, unet.hip)
    "semabs_conv3d_gather": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, C.c_char_p, I, P],
    "semabs_wgrad": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, C.c_char_p, I, P],
    "semabs_wgrad_mfma": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, C.c_char_p, I, P, L, P],
    "semabs_wgrad_conv3": [P, P, P, P, P, P, I, I, I, I, I, I, I, P, L, P],
    "semabs_wgrad_conv3_supported": [I, I, I, I, I, L, C.POINTER(I)],
    "semabs_conv3d_gnbwd_supported": [I, I, I, I, I, I, I, I, C.POINTER(I)],
    "semabs_conv3d_gnbwd": [P, P, P, P, P, P, P, P, P, P, I, P, I, P, I, I, I, I, I, I, I, P],
    "semabs_wgrad_conv3_gn_supported": [I, I, I, I, I, I, L, C.POINTER(I)],
    "semabs_wgrad_conv3_gn": [P, P, P, P, I, P, P, P, P, P, P, I, I, I, I, I, I, P, L, P],
    "semabs_chan_reduce": [P, P, P, P, P, I, L, I, I, P],
    "semabs_gn_meanrstd": [P, P, P, I, I, L, F, P],
    "semabs_gn_bwd_coef": [P, P, P, P, P, P, P, I, I, I, L, P],
    "semabs_gn_bwd_apply": [P, P, P, P, P, P, P, P, P, P, I, L, I, I, P],
    "semabs_ew": [P, P, P, L, I, F, P, P],
    "semabs_ew_scaled": [P, P, P, P, L, I, F, P, P],
    "semabs_grad_scale": [P, L, P, P, I, P, P, I, P],
    "semabs_maxpool3d_bwd": [P, P, P, I, I, I, I, I, P],
    "semabs_maxpool3d_bwd_add": [P, P, P, P, P, I, I, I, I, I, I, P],
    "semabs_linear_f32": [P, P, P, P, L, I, I, I, F, P],
    "semabs_linear_rows": [P, L, P, L, L, P, P, L, I, I, I, F, P, P, P, P, P, P, P],
    "semabs_scatter_mean_bwd": [P, P, P, P, I, L, I, L, P],
    "semabs_vool_sample": [P, P, P, C.POINTER(F), C.POINTER(F), C.POINTER(I), I, L, P, P],
    "semabs_vool_sample_bwd": [P, P, C.POINTER(F), C.POINTER(F), C.POINTER(I), I, L, P, P, P, P, P, P],
    "semabs_cos_bce": [P, P, P, P, I, L, F, L, P, P, P, P, P, P],
    "semabs_cos_head": [P, P, P, I, L, F, P, P, P, P],
    "semabs_gather_split16": [P, P, L, P, P, P],
    "semabs_gather_split16_batched": [P, I, L, P],
    "semabs_clip_grad_norm": [P, I, P, I, F, F, P, P],
    # relio.hip
    "semabs_relevancy_pack": [P, P, I, I, I, I, I, P],
    "semabs_text_pack": [P, P, I, I, P],
    "semabs_relevancy_unpack": [P, P, P, P, I, I, I, I, I, F, P],
    # evalm.hip
    "semabs_voxelize_eval": [P, P, P, P, P, P, P, P, L, L, L, P],
    "semabs_prediction_counts": [P, P, P, P, L, L, P],
    # timing helpers
    "semabs_event_create": [C.POINTER(P)],
    "semabs_event_destroy": [P],
    "semabs_event_elapsed_ms": [P, P, C.POINTER(F)],
}


TUNING_SIGNATURES = {"semabs_gemm_tune": [I, C.c_longlong], "semabs_conv_tune": [I, C.c_longlong]}


def lib():
    """Load (once) and return the ctypes handle; raise loudly when the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension was not built. Run "
                "`python semantic-abstraction_amd/build.py` (or __graft_entry__.build()). There is no CPU fallback."
            )
        h = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            if not hasattr(h, name):
                continue  # reported by tests/test_abi.py; calling it raises below
            fn = getattr(h, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        for name, argtypes in TUNING_SIGNATURES.items():        # present in the tuning build only
            if hasattr(h, name):
                fn = getattr(h, name)
                fn.argtypes = argtypes
                fn.restype = C.c_int
        h.semabs_last_error.restype = C.c_char_p
        _lib = h
    return _lib


CALL_HOOK = None      # measurement only (bench.py's per-kernel-class table): object with before(name, args) -> token and after(token)


def call(name: str, *args) -> None:
    h = lib()
    fn = getattr(h, name, None)
    if fn is None:
        raise RuntimeError(f"libsemabs_hip.so does not export {name}")
    hook = CALL_HOOK
    tok = hook.before(name, args) if hook is not None else None
    rc = fn(*args)
    if hook is not None:
        hook.after(tok)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {h.semabs_last_error().decode()}")


def ptr(t) -> int | None:
    """Device (or host) address of a tensor; None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "semabs C ABI takes contiguous buffers"
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("semabs_amd needs a HIP device (MI355X); there is no CPU fallback on the product path")
    return torch.device("cuda", torch.cuda.current_device())


def filled(shape, dtype: torch.dtype, value=0, device=None) -> torch.Tensor:
    """torch.full(shape, value, dtype) without an ATen kernel: torch.empty + semabs_fill_u32 on the current stream (4-byte dtypes, or zero for
    any dtype whose byte count is a multiple of 4)."""
    t = torch.empty(shape, dtype=dtype, device=device if device is not None else require_gpu())
    fill_(t, value)
    return t


def fill_(t: torch.Tensor, value=0) -> torch.Tensor:
    import struct
    assert t.is_contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes == 0:
        return t
    if value == 0:
        pat = 0
    elif t.dtype == torch.float32:
        pat = struct.unpack("<I", struct.pack("<f", float(value)))[0]
    elif t.dtype == torch.int32:
        pat = int(value) & 0xFFFFFFFF
    else:
        raise TypeError(f"fill_: non-zero fill of {t.dtype}")
    assert nbytes % 4 == 0, "fill_: byte count must be a multiple of 4"
    call("semabs_fill_u32", t.data_ptr(), nbytes, pat, stream())
    return t


def farr(vals):
    return (F * len(vals))(*[float(v) for v in vals])


def iarr(vals):
    return (I * len(vals))(*[int(v) for v in vals])
