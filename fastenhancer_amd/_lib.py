"""ctypes binding of include/fastenhancer_hip.h.  Loading fails loudly: there is no fallback."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# FASTENHANCER_HIP_LIB: a side build of the same library (ab/lib_<tag>.so from FE_BUILD_TAG=<tag> python -m fastenhancer_amd.build)
LIB_PATH = os.environ.get("FASTENHANCER_HIP_LIB") or os.path.join(HERE, "libfastenhancer_hip.so")
FE_MAX_KERNELS = 8

FE_OK = 0
FE_ARCH_FASTENHANCER = 0
FE_ARCH_BSRNN = 1
FE_ARCH_FSPEN = 2
FE_ARCH_LISENNET = 3
FE_OFFLINE_AUTO, FE_OFFLINE_FRAME_WALK, FE_OFFLINE_TIME_BATCHED = 0, 1, 2
FE_STEP_KERNEL_WAVES4, FE_STEP_KERNEL_WG8, FE_STEP_KERNEL_WG8_PERSIST = 0, 1, 2


class fe_config(ctypes.Structure):
    _fields_ = [
        ("arch", c_int), ("n_fft", c_int), ("hop_size", c_int), ("win_size", c_int),
        ("channels", c_int), ("n_kernels", c_int), ("kernel_size", c_int * FE_MAX_KERNELS),
        ("stride", c_int), ("rf_channels", c_int), ("rf_freq", c_int), ("rf_blocks", c_int),
        ("rf_heads", c_int), ("input_compression", c_float), ("kernel_size_time", c_int), ("channels_frnn", c_int), ("lookbehind", c_int), ("ln", c_int), ("rf_eps", c_float),
        ("bidirectional", c_int),
    ]


# every symbol include/fastenhancer_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "fe_create": (c_int, [POINTER(fe_config), POINTER(c_void_p)]),
    "fe_destroy": (None, [c_void_p]),
    "fe_weight_floats": (c_size_t, [c_void_p]),
    "fe_weight_sections": (c_int, [c_void_p]),
    "fe_weight_section": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_size_t), POINTER(c_size_t)]),
    "fe_load_weights": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fe_state_floats": (c_size_t, [c_void_p, c_int]),
    "fe_state_init": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "fe_step": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "fe_step_host": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fe_spec_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fe_set_time_pipeline": (c_int, [c_void_p, c_int]),
    "fe_set_offline_engine": (c_int, [c_void_p, c_int]),
    "fe_set_step_kernel": (c_int, [c_void_p, c_int]),
    "fe_set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "fe_get_option": (c_int, [c_void_p, c_char_p, POINTER(c_int)]),
    "fe_options": (c_int, []),
    "fe_option_name": (c_char_p, [c_int]),
    "fe_last_step_kernel": (c_char_p, [c_void_p]),
    "fe_offline_work_floats": (c_size_t, [c_void_p, c_int, c_int]),
    "fe_offline": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fe_offline_ragged_work_floats": (c_size_t, [c_void_p, c_int, c_int]),
    "fe_offline_ragged": (c_int, [c_void_p, c_void_p, c_size_t, POINTER(c_int), c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "fe_stft_step": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fe_istft_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "fe_stft_offline": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fe_istft_offline": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fe_flops_per_frame": (c_double, [c_void_p]),
    "fe_debug_stages": (c_int, [c_void_p]),
    "fe_debug_stage": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]),
    "fe_debug_floats": (c_size_t, [c_void_p]),
    "fe_debug_step": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p]),
    "fe_profile_step": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]),
    "fe_debug_poison_lds": (c_int, [c_void_p]),
    "fe_last_error": (c_char_p, []),
    "fe_version": (c_char_p, []),
    "fe_build_key": (c_char_p, []),
}

_lib = None


class FEError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libfastenhancer_hip.so (built in-tree by fastenhancer_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FEError(
            f"{LIB_PATH} is missing: build it with `python -m fastenhancer_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    # the in-tree library must be built from the tree's sources (a side build loaded through FASTENHANCER_HIP_LIB is its author's business)
    if not os.environ.get("FASTENHANCER_HIP_LIB"):
        from .build import source_key
        have, want = lib.fe_build_key().decode(), source_key()
        if have != want:
            raise FEError(f"{LIB_PATH} was built from other sources (fe_build_key {have}, tree {want}): rebuild it with `python -m fastenhancer_amd.build`")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != FE_OK:
        msg = load().fe_last_error().decode("utf-8", "replace")
        raise FEError(f"{what} failed ({rc}): {msg}")
