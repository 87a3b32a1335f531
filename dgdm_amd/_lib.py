"""ctypes binding of libdgdm_hip.so (the C-ABI declared in include/dgdm_hip.h).

The library is the product path: if it cannot be loaded, or a call fails, this module raises -
there is no CPU or eager-PyTorch fallback anywhere in ``dgdm_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Sequence

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdgdm_hip.so")

OK, EINVAL, EKEY, EHIP, EOBJECTIVE, EMODE, ENODEVICE = 0, -1, -2, -3, -4, -5, -6


class DgdmError(RuntimeError):
    pass


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64), ("dtype", C.c_int32)]


class Objective(C.Structure):
    _fields_ = [("lin", C.c_float * 3), ("quad", C.c_float * 3), ("use_rowcoef", C.c_int32), ("object", C.c_int32)]


class TrainGroups(C.Structure):
    _fields_ = [("t_index_dev", C.c_void_p), ("t_values_dev", C.c_void_p), ("n_t", C.c_int32), ("rows_per_object", C.c_int32)]


class GuidanceConfig(C.Structure):
    _fields_ = [("batch", C.c_int32), ("grid_size", C.c_int32), ("num_pos", C.c_int32), ("ori_lo", C.c_float),
                ("ori_hi", C.c_float), ("max_chains", C.c_int32), ("num_train_timesteps", C.c_int32),
                ("sub_batch_size", C.c_int32), ("num_object_points", C.c_int32), ("max_objects", C.c_int32)]


_P = C.c_void_p
# name -> (restype, argtypes); mirrors include/dgdm_hip.h line by line (tests/test_host_logic.py::test_abi_exports_every_declared_symbol checks the symbol list)
PROTOTYPES = {
    "dgdm_version": (C.c_int, []),
    "dgdm_last_error": (C.c_char_p, []),
    "dgdm_device_init": (C.c_int, [C.c_int]),
    "dgdm_objective_from_name": (C.c_int, [C.c_char_p, C.POINTER(Objective)]),
    "dgdm_unet1d_create": (C.c_int, [C.POINTER(_P), C.POINTER(Tensor), C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int]),
    "dgdm_unet1d_destroy": (None, [_P]),
    "dgdm_unet1d_forward": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "dgdm_ddim_guided_step": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "dgdm_ddim_add_noise": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, C.c_float, _P]),
    "dgdm_dynamics_create": (C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(Tensor), C.c_int, C.c_int, C.c_int]),
    "dgdm_dynamics_destroy": (None, [_P]),
    "dgdm_dyn2d_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "dgdm_pointnet2_forward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "dgdm_dyn3d_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "dgdm_farthest_point_sample": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "dgdm_query_ball_point": (C.c_int, [C.c_float, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "dgdm_square_distance": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "dgdm_index_points": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "dgdm_linear_act": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "dgdm_group_max": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _P, _P]),
    "dgdm_guidance_create": (C.c_int, [C.POINTER(_P), _P, C.POINTER(GuidanceConfig)]),
    "dgdm_guidance_destroy": (None, [_P]),
    "dgdm_debug_chain_layer": (C.c_int, [_P, _P, _P, _P, _P]),
    "dgdm_unet1d_set_contraction_dtype": (C.c_int, [_P, C.c_int]),
    "dgdm_unet1d_effective_form": (C.c_int, [_P, C.c_int, C.c_int]),
    "dgdm_finger_decode_2d": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P, _P]),
    "dgdm_finger_decode_3d": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P, _P]),
    "dgdm_guidance_set_contraction_dtype": (C.c_int, [_P, C.c_int]),
    "dgdm_guidance_set_objects": (C.c_int, [_P, _P, C.c_int, _P]),
    "dgdm_guidance_rows": (C.c_int64, [_P]),
    "dgdm_guidance_starts_per_call": (C.c_int64, [_P]),
    "dgdm_dyn2d_guidance_grad": (C.c_int, [_P, _P, C.c_int, C.POINTER(Objective), _P, C.c_int, _P, _P]),
    "dgdm_dyn3d_guidance_grad": (C.c_int, [_P, _P, C.c_int, C.POINTER(Objective), _P, _P, C.c_int, _P, _P]),
    "dgdm_guided_chains_run": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.POINTER(Objective), _P, _P, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                         C.POINTER(C.c_float), C.c_int, _P, _P]),
    "dgdm_guidance_orientation_sweep": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P]),
    "dgdm_convergence_rowcoef": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _P]),
    "dgdm_torch_rng_seed": (C.c_int, [_P, C.c_int64, C.c_uint64]),
    "dgdm_torch_rng_randint": (C.c_int, [_P, C.c_int64, C.c_uint32, C.c_int64, _P]),
    "dgdm_torch_rng_fps_starts": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int64]),
    "dgdm_prof_enable": (C.c_int, [C.c_int]),
    "dgdm_guidance_debug_fps_path": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32)]),
    "dgdm_debug_pointnet_indices": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dgdm_guidance_debug_partials": (C.c_int, [_P, C.c_int, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P]),
    "dgdm_prof_read_stage": (C.c_int, [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dgdm_prof_read": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dgdm_trainer2d_create": (C.c_int, [C.POINTER(_P), C.POINTER(Tensor), C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "dgdm_trainer2d_destroy": (None, [_P]),
    "dgdm_trainer2d_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_int, _P, C.POINTER(C.c_float), _P]),
    "dgdm_trainer2d_forward_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, _P, C.POINTER(C.c_float), _P]),
    "dgdm_trainer2d_set_groups": (C.c_int, [_P, C.POINTER(TrainGroups)]),
    "dgdm_trainer2d_gradient_count": (C.c_int64, [_P]),
    "dgdm_trainer2d_gradients": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "dgdm_trainer2d_apply": (C.c_int, [_P, C.c_float, _P]),
    "dgdm_trainer2d_running_stats": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "dgdm_trainer2d_export": (C.c_int, [_P, C.c_int, C.POINTER(Tensor), C.c_int]),
    "dgdm_trainer2d_steps": (C.c_int64, [_P]),
    "dgdm_unet_trainer_create": (C.c_int, [C.POINTER(_P), C.POINTER(Tensor), C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_float, C.c_float, C.c_float, C.c_float]),
    "dgdm_unet_trainer_destroy": (None, [_P]),
    "dgdm_unet_trainer_step": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_float, _P, C.POINTER(C.c_float), _P]),
    "dgdm_unet_trainer_forward_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int64, C.c_int, _P, C.POINTER(C.c_float), _P]),
    "dgdm_unet_trainer_gradient_count": (C.c_int64, [_P]),
    "dgdm_unet_trainer_gradients": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_float, _P]),
    "dgdm_unet_trainer_apply": (C.c_int, [_P, C.c_float, _P]),
    "dgdm_unet_trainer_ema_step": (C.c_int, [_P, C.c_float, C.c_float, _P]),
    "dgdm_unet_trainer_export": (C.c_int, [_P, C.c_int, C.POINTER(Tensor), C.c_int]),
    "dgdm_unet_trainer_import": (C.c_int, [_P, C.c_int, C.POINTER(Tensor), C.c_int, C.c_int64]),
    "dgdm_unet_trainer_steps": (C.c_int64, [_P]),
    "dgdm_trainer3d_create": (C.c_int, [C.POINTER(_P), C.POINTER(Tensor), C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "dgdm_trainer3d_destroy": (None, [_P]),
    "dgdm_trainer3d_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_int, _P, C.POINTER(C.c_float), _P]),
    "dgdm_trainer3d_export": (C.c_int, [_P, C.c_int, C.POINTER(Tensor), C.c_int]),
    "dgdm_trainer3d_steps": (C.c_int64, [_P]),
    "dgdm_trainer3d_forward_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, _P, C.POINTER(C.c_float), _P]),
    "dgdm_trainer3d_gradient_count": (C.c_int64, [_P]),
    "dgdm_trainer3d_gradients": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "dgdm_trainer3d_apply": (C.c_int, [_P, C.c_float, _P]),
    "dgdm_trainer3d_running_stats_count": (C.c_int64, [_P]),
    "dgdm_trainer3d_running_stats": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "dgdm_trainer3d_debug_read": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P]),
}

_lib = None


def lib() -> C.CDLL:
    """The loaded library; raises if it has not been built (``python -m dgdm_amd.build``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DgdmError(f"{LIB_PATH} is missing: the HIP library has not been built (run `python -m dgdm_amd.build`). "
                            "dgdm_amd has no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def check(rc: int) -> None:
    if rc == OK:
        return
    msg = lib().dgdm_last_error().decode("utf-8", "replace")
    if rc == EOBJECTIVE:
        raise ValueError('opt obj not supported')            # generator/diffusion.py:470
    if rc == EMODE:
        raise ValueError('model type not supported')         # generator/diffusion.py:502
    raise DgdmError(f"libdgdm_hip error {rc}: {msg}")


def device_init(ordinal: int = 0) -> None:
    check(lib().dgdm_device_init(ordinal))


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device tensors handed to libdgdm_hip must be contiguous CUDA tensors"
    return t.data_ptr()


class PackedStateDict:
    """Keeps the host copies alive while a DgdmTensor array points at them."""

    def __init__(self, sd: Dict[str, torch.Tensor], strip_prefix: Sequence[str] = ("module.",)):
        self.keep: List[np.ndarray] = []
        self.names: List[bytes] = []
        items = []
        for k, v in sd.items():
            if not isinstance(v, torch.Tensor):
                continue
            for p in strip_prefix:
                if k.startswith(p):
                    k = k[len(p):]
            v = v.detach().cpu()
            if v.dtype in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
                arr, dt = np.ascontiguousarray(v.to(torch.float32).numpy()), 0
            elif v.dtype == torch.int64:
                arr, dt = np.ascontiguousarray(v.numpy()), 1
            else:
                continue
            self.keep.append(arr)
            self.names.append(k.encode())
            items.append((self.names[-1], arr.ctypes.data, arr.size, dt))
        self.n = len(items)
        self.array = (Tensor * self.n)()
        for i, (n, p, sz, dt) in enumerate(items):
            self.array[i].name, self.array[i].data, self.array[i].numel, self.array[i].dtype = n, p, sz, dt
