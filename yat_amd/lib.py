"""ctypes binding of libyat_hip.so -- the only door between the Python host code and the HIP kernels.

The signatures below mirror include/yat_hip.h one to one (tests/test_abi.py checks that every
symbol the header declares is exported and bound).  There is NO fallback: if the library is
missing or a symbol cannot be resolved, importing the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YAT_HIP_LIB") or os.path.join(_HERE, "libyat_hip.so")   # override: A/B of two builds

P, I, I64, U64, F, D = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double


class GemmEpilogue(C.Structure):
    """yat_gemm_epilogue (include/yat_hip.h).  ``struct_size`` is filled in here: positional / keyword arguments start at
    ``bias``."""
    _fields_ = [("struct_size", C.c_uint32), ("bias", P), ("aux_out", P), ("activation", I), ("gate", P), ("residual", P),
                ("ld_aux", I), ("ld_gate", I), ("ld_residual", I), ("rows_per_batch", I),
                ("glu_u", P), ("ld_glu_u", I), ("pre_add", P), ("ld_pre_add", I), ("dact_z", P), ("ld_dact_z", I),
                ("a_rowsum_out", P), ("a_rowsum_accumulate", I), ("a2", P), ("b2", P), ("k2", I), ("a2_group_n", I)]


    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(GemmEpilogue), *args, **kw)


class PlanArg(C.Union):
    _fields_ = [("i", C.c_int64), ("d", C.c_double), ("p", C.c_void_p)]


class PlanEntry(C.Structure):
    """yat_plan_entry (include/yat_hip.h)."""
    _fields_ = [("op", C.c_int32), ("nargs", C.c_int32), ("a", PlanArg * 30)]


class GemmProblem(C.Structure):
    _fields_ = [("M", I), ("N", I), ("K", I), ("A", P), ("lda", I), ("B", P), ("ldb", I), ("C", P), ("ldc", I),
                ("epilogue", C.POINTER(GemmEpilogue))]


# name -> (restype, argtypes)
SIGNATURES = {
    "yat_version": (I, []),
    "yat_gemm_epilogue_size": (U64, []),
    "yat_gemm_bf16": (I, [I, I, I, I, I, P, I, P, I, P, I, C.POINTER(GemmEpilogue), P]),
    "yat_gemm_bf16_ex": (I, [I, I, I, I, I, P, I, P, I, P, I, C.POINTER(GemmEpilogue), I, P, U64, P]),
    "yat_gemm_grouped_bf16": (I, [I, I, I, C.POINTER(GemmProblem), P]),
    "yat_lokr_delta": (I, [I, I, I, I, I, P, P, P, F, P, I, P]),
    "yat_lokr_project_workspace_bytes": (U64, [I, I, I]),
    "yat_lokr_project": (I, [I, I, I, I, I, P, P, P, F, P, I, P, P, P, P, P]),
    "yat_colsum_workspace_bytes": (U64, [I, I]),
    "yat_colsum_bf16": (I, [I, I, P, I, P, I, P, P]),
    "yat_modulation_fwd": (I, [I, I, I, P, P, I, I, P, P]),
    "yat_modulation_bwd": (I, [I, I, I, P, P, I, P, I, I, P]),
    "yat_ln_bwd_workspace_bytes": (U64, [I, I, I]),
    "yat_ln_modulate_fwd": (I, [I, I, I, F, P, P, P, I, P, P, P, P]),
    "yat_ln_modulate_bwd": (I, [I, I, I, P, P, P, P, I, P, P, P, P, P, I, P, I, P]),
    "yat_rmsnorm_bwd_workspace_bytes": (U64, [I, I]),
    "yat_rmsnorm_fwd": (I, [I, I, F, P, P, P, P, P]),
    "yat_rmsnorm_bwd": (I, [I, I, P, P, P, P, P, P, I, P, P]),
    "yat_linear_attn_workspace_bytes": (U64, [I, I, I]),
    "yat_linear_attn_fwd": (I, [I, I, I, P, I, I, I, P, I, P, P]),
    "yat_linear_attn_bwd": (I, [I, I, I, P, I, I, I, P, I, P, I, P, P, P]),
    "yat_sdpa_fwd": (I, [I, I, I, I, I, F, P, I, P, P, I, P, P, P, I, P, P]),
    "yat_sdpa_bwd": (I, [I, I, I, I, I, F, P, I, P, P, I, P, P, P, I, P, I, P, P, P, I, P, P, I, P, I, I, P]),
    "yat_sdpa_fwd_packed": (I, [I, I, I, I, I, F, P, I, P, P, I, P, I, P, P, P, I, P, P]),
    "yat_sdpa_bwd_packed": (I, [I, I, I, I, I, F, P, I, P, P, I, P, I, P, P, P, I, P, I, P, P, P, I, P, P, I, P, I, I, P]),
    "yat_dwconv_glu_bwd_workspace_bytes": (U64, [I, I, I, I]),
    "yat_dwconv_glu_fwd": (I, [I, I, I, I, P, P, P, P, P, P]),
    "yat_dwconv_glu_bwd": (I, [I, I, I, I, P, P, P, P, P, P, P, P, P, I, P, P, P]),
    "yat_gate_bwd_workspace_bytes": (U64, [I, I, I]),
    "yat_gate_bwd": (I, [I, I, I, P, P, P, I, P, P, I, P, I, P, P]),
    "yat_act_fwd": (I, [I64, I, P, P, P]),
    "yat_act_bwd": (I, [I64, I, P, P, P, P]),
    "yat_add_bf16": (I, [I64, P, P, P, P]),
    "yat_f32_to_bf16": (I, [I64, P, P, P]),
    "yat_memset_zero": (I, [P, U64, P]),
    "yat_transpose_bf16": (I, [I, I, I, P, P, P]),
    "yat_timestep_embed_fwd": (I, [I, I, P, P, P]),
    "yat_pad_mask": (I, [I, I, I, P, P, P, P, P, P, P]),
    "yat_pack_mask": (I, [I, I, I, I, P, P, P, P, P, P, P]),
    "yat_flow_mix": (I, [I, I64, P, P, P, P, P, P]),
    "yat_mse_fwd_bwd": (I, [I64, P, P, F, P, P, P, P]),
    "yat_lokr_rows": (I, [I64, I, I, I, P, P, P, P]),
    "yat_lokr_rows_fwd_flat": (I, [I64, I, I, I, P, P, P, I, P]),
    "yat_dropout": (I, [I64, F, U64, I, P, P, P]),
    "yat_lora_scatter_b": (I, [I, I, I, F, P, P, P, P]),
    "yat_rank_expand": (I, [I64, I, I, P, P, P, I, F, I, P]),
    "yat_lokr_small_wgrad_workspace_bytes": (U64, [I64, I, I]),
    "yat_lokr_small_wgrad": (I, [I64, I, I, I, P, P, I, P, I, F, I, P, P]),
    "yat_patch_rearrange": (I, [I, I, I, I, I, I, I, P, P, P]),
    "yat_add_pos_embed": (I, [I64, I, I, P, P, P, P]),
    "yat_ddpm_add_noise": (I, [I, I64, P, P, P, P, P, P]),
    "yat_mse_bf16_chunk": (I, [I, I64, I64, P, P, F, P, P, P, P]),
    "yat_hadamard_scale": (I, [I, I, P, I, P, I, F, P, I, P]),
    "yat_hadamard_bwd": (I, [I, I, P, I, P, I, P, I, F, P, I, P, I, P]),
    "yat_dora_delta": (I, [I, I, P, I, P, I, P, F, P, I, P, P, P]),
    "yat_dora_bwd": (I, [I, I, P, I, P, I, P, I, F, P, P, P, I, P, P]),
    "yat_qknorm_concat_fwd": (I, [I, I, I, I, I, F, P, I, P, I, P, P, P, P, P, I, P, P]),
    "yat_qknorm_concat_bwd_workspace_bytes": (U64, [I, I, I, I]),
    "yat_qknorm_concat_bwd": (I, [I, I, I, I, I, P, I, P, I, P, P, P, P, P, P, I, P, I, P, I, P, P, P, P, I, P, P]),
    "yat_joint_rows": (I, [I, I, I, I, P, I, P, I, P, I, I, P]),
    "yat_gradnorm_workspace_bytes": (U64, [I64, I]),
    "yat_gradnorm_clip": (I, [I64, P, I, P, F, P, P, P, P]),
    "yat_gradnorm_pieces_partial": (I, [P, I, P, P, I, P, P, P]),
    "yat_gradnorm_pieces_finish": (I, [I, P, P, P, F, P, P, P]),
    "yat_adamw_step": (I, [I64, P, P, P, P, P, D, D, D, D, D, I, I, P, D, I, P]),
    "yat_plan_op_id": (I, [C.c_char_p]),
    "yat_plan_replay": (I, [C.POINTER(PlanEntry), I, C.POINTER(I)]),
    "yat_comm_available": (I, []),
    "yat_comm_unique_id": (I, [P]),
    "yat_comm_init": (I, [I, I, P]),
    "yat_comm_world": (I, []),
    "yat_comm_rank": (I, []),
    "yat_comm_broadcast": (I, [P, U64, I, P]),
    "yat_bucket_allreduce_async": (I, [P, U64, I, P, P]),
    "yat_comm_allreduce": (I, [P, U64, I, I, P]),
    "yat_bucket_reduce_scatter_async": (I, [P, U64, I, P, P]),
    "yat_comm_allgather": (I, [P, U64, P]),
    "yat_comm_wait": (I, [I, P]),
    "yat_comm_destroy": (I, []),
    "yat_comm_last_error": (C.c_char_p, []),
}


class YatLibraryError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """Load libyat_hip.so and bind every symbol.  Raises YatLibraryError -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YatLibraryError(
            f"{LIB_PATH} not found: build it with `python -m yat_amd.build` (or __graft_entry__.build()). "
            "The HIP path has no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise YatLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise YatLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        if rc >= 1000 or rc == -2:
            kind = f"communication error {rc}: {(load().yat_comm_last_error() or b'').decode()}"
        else:
            kind = "invalid argument" if rc < 0 else f"hipError_t {rc}"
        raise YatLibraryError(f"{what} failed: {kind}")
