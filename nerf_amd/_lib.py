"""ctypes binding of libnerf_hip.so (the C ABI declared in include/nerf_hip.h).

The library is the only compute path of this package: there is no PyTorch or
CPU fallback, and a missing library raises at first use.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NERF_HIP_LIB", os.path.join(_HERE, "libnerf_hip.so"))  # override: A/B-testing a build

_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float

# name -> (restype, argtypes); mirrors include/nerf_hip.h
SIGNATURES = {
    "nerf_abi_version": (_I, []),
    "nerf_last_error": (C.c_char_p, []),
    "nerf_raygen": (_I, [_P, _F, _F, _F, _F, _I, _I, _P, _L, _P, _P, _P]),
    "nerf_raygen_multi": (_I, [_P, _I, _I, _I, _P, _P, _L, _P, _P, _P, _P, _P]),
    "nerf_stratified_sample": (_I, [_P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "nerf_raygen_sample": (_I, [_P, _F, _F, _F, _F, _I, _I, _P, _L, _P, _P, _I, _P, _P, _P, _P, _P]),
    "nerf_pdf_sample": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P]),
    "nerf_ndc_rays": (_I, [_P, _P, _L, _I, _I, _F, _F, _P, _P, _P]),
    "nerf_normalize_dirs": (_I, [_P, _L, _P, _P]),
    "nerf_encode_pe": (_I, [_P, _L, _I, _I, _P, _P]),
    "nerf_hashgrid_fwd": (_I, [_P, _L, _P, _P, _P, _I, _P, _P, _P, _P]),
    "nerf_hashgrid_bwd_workspace_bytes": (C.c_size_t, [_L, _P, _I]),
    "nerf_hashgrid_bwd": (_I, [_P, _L, _P, _P, _P, _I, _P, _P, C.c_size_t, _P]),
    "nerf_composite_fwd": (_I, [_P, _P, _P, _L, _I, _I, _F, _P, _P, _P, _P]),
    "nerf_composite_bwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _P, _P, _P]),
    "nerf_occupancy_mask": (_I, [_P, _P, _L, _P, _L, _I, _P, _P]),
    "nerf_occupancy_compact_workspace_ints": (C.c_size_t, [_L]),
    "nerf_occupancy_compact": (_I, [_P, _L, _P, _L, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "nerf_expand_kept": (_I, [_P, _P, _P, _L, _P, _P, _P]),
    "nerf_gather_kept": (_I, [_P, _P, _P, _L, _P, _P, _P]),
    "nerf_gather_kept_by_slot": (_I, [_P, _P, _P, _L, _P, _P, _P]),
    "nerf_mlp_vanilla_pack_floats": (C.c_size_t, []),
    "nerf_mlp_vanilla_pack": (_I, [_P, _I, _P, _P]),
    "nerf_mlp_vanilla_ws_floats_per_point": (C.c_size_t, []),
    "nerf_mlp_vanilla_ws2_floats_per_point": (C.c_size_t, []),
    "nerf_mlp_vanilla_fwd": (_I, [_P, _P, _I, _L, _P, _P, _P, _P, _P]),
    "nerf_mlp_vanilla_bwd": (_I, [_P, _P, _L, _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    "nerf_mlp_vanilla_relu_masks": (_I, [_P, _L, _P, _P]),
    "nerf_mlp_ngp_pack_floats": (C.c_size_t, []),
    "nerf_mlp_ngp_ws_floats_per_point": (C.c_size_t, []),
    "nerf_mlp_ngp_ws2_floats_per_point": (C.c_size_t, []),
    "nerf_mlp_ngp_pack": (_I, [_P, _P, _P]),
    "nerf_mlp_ngp_fwd": (_I, [_P, _P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nerf_mlp_ngp_bwd": (_I, [_P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _I, _P]),
    "nerf_mlp_ngp_fwd_counted": (_I, [_P, _P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nerf_mlp_ngp_bwd_counted": (_I, [_P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _I, _P]),
    "nerf_render_loss_workspace_floats": (C.c_size_t, [_L]),
    "nerf_render_loss_fwd": (_I, [_P, _P, _P, _P, _L, _I, _F, _P, _P, _P]),
    "nerf_render_loss_bwd": (_I, [_P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P, _P]),
    "nerf_adam_step": (_I, [_I, _P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P]),
    "nerf_adam_step_capturable": (_I, [_I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P]),
    "nerf_adam_set_lr": (_I, [_P, _F, _P]),
    "nerf_clip_grad_norm_workspace_floats": (C.c_size_t, [_I, _P]),
    "nerf_clip_grad_norm": (_I, [_I, _P, _P, _F, _P, _P, _P]),
    "nerf_clip_grad_norm_static": (_I, [_I, _P, _P, _F, _P, _P]),
    "nerf_rccl_version": (_I, []),
    "nerf_comm_unique_id": (_I, [_P]),
    "nerf_comm_init_rank": (_I, [_P, _I, _I, _P]),
    "nerf_comm_destroy": (_I, [_P]),
    "nerf_allreduce_sum": (_I, [_P, C.c_size_t, _P, _P]),
}

_lib = None


class NerfHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the library once.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NerfHipError(
                f"{LIB_PATH} is missing: build it with `make -C nerf_amd/csrc` "
                "(or __graft_entry__.build()); nerf_amd has no fallback path")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().nerf_last_error()
        raise NerfHipError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_of(t: torch.Tensor):
    """The current HIP stream of the tensor's device, as a void*."""
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require(cond: bool, message: str = "bad argument") -> None:
    """Argument checks that guard a kernel's indexing: they must hold under ``python -O`` too, so no ``assert``."""
    if not cond:
        raise ValueError(message)


def need_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NerfHipError("nerf_amd ops run on the GPU only: got a tensor on " + str(t.device))


def f32c(t: torch.Tensor) -> torch.Tensor:
    """Dense float32 view/copy (expanded views are materialised)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def ptr_array(tensors):
    """Host array of device pointers."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def int_array(vals):
    arr = (C.c_int * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = int(v)
    return arr
