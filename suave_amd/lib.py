"""ctypes binding of ``libsuave_hip.so`` (C-ABI declared in ``include/suave_hip.h``).

The product path has no CPU fallback: if the HIP library is missing or an
entry point is absent this module raises, loudly.
"""

from __future__ import annotations

import ctypes as C
import os

from .plan import BatchT, ColT, ElboOutT, HeadT, ModelT, NoiseT

# SUAVE_HIP_LIB points at another build of the same library (tuning builds under build/); default: the in-tree one
_LIB_PATH = os.environ.get("SUAVE_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsuave_hip.so")
_lib = None

EXPORTS = (
    "suave_last_error",
    "suave_plan_create",
    "suave_plan_destroy",
    "suave_workspace_bytes",
    "suave_elbo_forward",
    "suave_elbo_backward",
    "suave_elbo_backward_overlap",
    "suave_adam_step",
    "suave_head_workspace_bytes",
    "suave_head_step",
    "suave_elbo_joint_backward",
    "suave_encode",
    "suave_prior_latents",
    "suave_conditional_latents",
    "suave_decode_sample",
    "suave_decode_sample_static",
    "suave_posterior_draws",
    "suave_posterior_draws_by_row",
    "suave_encode_hard",
    "suave_rng_draws",
    "suave_decode_static_max_rows",
    "suave_head_params_width",
    "suave_decode_params",
    "suave_decode_log_px",
    "suave_pack_columns",
    "suave_unpack_columns",
    "suave_gemm_arithmetic_set",
    "suave_gemm_arithmetic_get",
    "suave_small_batch_rows_set",
    "suave_small_batch_rows_get",
    "suave_profile_enable",
    "suave_profile_read",
)


class SuaveHipError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load the shared library (once) and declare the entry-point signatures."""

    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise SuaveHipError(
            f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or suave_amd/csrc/build.sh). There is no CPU fallback."
        )
    # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so).  The device pointers and streams this library
    # is handed come from torch, so both must bind to the SAME runtime: import torch first, then the soname of
    # libsuave_hip.so's dependency resolves to the copy already in the process.  Loaded the other way round the
    # process holds two runtimes and every launch here fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401

    lib = C.CDLL(_LIB_PATH)
    missing = [name for name in EXPORTS if not hasattr(lib, name)]
    if missing:
        raise SuaveHipError(f"libsuave_hip.so lacks entry points: {missing}")
    vp, i32, i64, f32, u64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_size_t
    lib.suave_last_error.restype = C.c_char_p
    lib.suave_last_error.argtypes = []
    lib.suave_plan_create.argtypes = [C.POINTER(ModelT), C.POINTER(ColT), C.POINTER(vp)]
    lib.suave_plan_destroy.argtypes = [vp]
    lib.suave_workspace_bytes.argtypes = [vp, i64, C.POINTER(sz)]
    lib.suave_elbo_forward.argtypes = [vp, vp, C.POINTER(BatchT), C.POINTER(NoiseT), f32, f32, C.c_int, vp, sz, vp,
                                       C.POINTER(ElboOutT)]
    lib.suave_elbo_backward.argtypes = [vp, vp, C.POINTER(BatchT), C.POINTER(NoiseT), f32, f32, i64, vp, sz, vp,
                                        C.POINTER(ElboOutT), vp, vp]
    lib.suave_elbo_backward_overlap.argtypes = [vp, vp, C.POINTER(BatchT), C.POINTER(NoiseT), f32, f32, i64, vp, sz, vp,
                                                C.POINTER(ElboOutT), vp, vp, vp]
    lib.suave_head_workspace_bytes.argtypes = [i32, i32, C.POINTER(sz)]
    lib.suave_head_step.argtypes = [C.POINTER(HeadT), i32, vp, i32, vp, i64, i64, vp, vp]
    lib.suave_elbo_joint_backward.argtypes = [vp, vp, C.POINTER(BatchT), C.POINTER(NoiseT), f32, f32, i64, vp, sz, vp,
                                              C.POINTER(ElboOutT), vp, C.POINTER(HeadT), vp]
    lib.suave_adam_step.argtypes = [vp, vp, vp, vp, C.POINTER(i32), C.POINTER(f32), i32, i64, f32, f32, f32, vp]
    lib.suave_encode.argtypes = [vp, vp, C.POINTER(BatchT), f32, vp, sz, vp, vp, vp, vp, vp]
    lib.suave_encode_hard.argtypes = [vp, vp, C.POINTER(BatchT), f32, vp, sz, vp, vp, vp]
    lib.suave_rng_draws.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.suave_decode_static_max_rows.argtypes = [vp, i32, C.POINTER(C.c_int64)]
    lib.suave_prior_latents.argtypes = [vp, vp, vp, vp, vp, u64, i64, i64, vp, vp, vp]
    lib.suave_conditional_latents.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, i64, i64, vp, vp, vp]
    lib.suave_decode_sample.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp, vp, u64, i64, vp, sz, vp, vp, i32, vp, i32, vp]
    lib.suave_decode_sample_static.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp, vp, u64, i64, vp, vp, i32, vp, i32, vp]
    lib.suave_posterior_draws.argtypes = [vp, vp, i64, i32, vp, vp, u64, i64, vp, vp, vp]
    lib.suave_posterior_draws_by_row.argtypes = [vp, vp, i64, i32, vp, vp, u64, i64, vp, vp, vp]
    lib.suave_head_params_width.argtypes = [vp, i32, C.POINTER(i32)]
    lib.suave_decode_params.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp, sz, vp, vp]
    lib.suave_decode_log_px.argtypes = [vp, vp, vp, vp, C.POINTER(BatchT), vp, sz, vp, vp, vp]
    lib.suave_pack_columns.argtypes = [vp, vp, i64, vp, i32, vp, i32, vp, i32, vp, vp]
    lib.suave_unpack_columns.argtypes = [vp, vp, i32, i64, vp, i32, vp, i32, vp, vp, vp, vp, i64, vp]
    lib.suave_gemm_arithmetic_set.argtypes = [C.c_int]
    lib.suave_gemm_arithmetic_get.argtypes = [C.POINTER(C.c_int)]
    lib.suave_small_batch_rows_set.argtypes = [C.c_int]
    lib.suave_small_batch_rows_get.argtypes = [C.POINTER(C.c_int)]
    lib.suave_profile_enable.argtypes = [C.c_int]
    lib.suave_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(i64), i32]
    for name in EXPORTS:
        if name != "suave_last_error":
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().suave_last_error()
        raise SuaveHipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
