"""ctypes loader for libvilfusion.so.  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvilfusion.so")


class VilFusionError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libvilfusion error {code}: {msg}")
        self.code = code


def lib_path() -> str:
    return _SO


class EngineOptsC(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("windows", C.c_int), ("capacity", C.c_int), ("bandwidth", C.c_int),
                ("device", C.c_int), ("gravity", C.c_double * 3),
                ("lambda0", C.c_double), ("lambda_up", C.c_double), ("lambda_down", C.c_double),
                ("lambda_min", C.c_double), ("lambda_max", C.c_double), ("chunks", C.c_int),
                ("cold_start", C.c_int), ("accept_rel", C.c_double),
                ("refine_iterations", C.c_int), ("refine_min_keyframes", C.c_int), ("refine_rel_stop", C.c_double),
                ("lm_excursion", C.c_int), ("gauge_floor", C.c_double),
                ("incremental", C.c_int), ("wildfire", C.c_double), ("min_model_fidelity", C.c_double),
                ("max_far_factors", C.c_int)]


class EngineTuningC(C.Structure):
    """solver-form switches (include/vilfusion.h vf_engine_tuning): what tests and tools set to reach one form on purpose"""
    _fields_ = [("struct_size", C.c_uint32), ("sweep_two_sided_max", C.c_int), ("hybrid_threshold", C.c_int), ("use_hip_graph", C.c_int),
                ("solve_split_min", C.c_int), ("solve_assemble_min", C.c_int), ("solve_assemble_waves", C.c_int),
                ("hybrid_active_list", C.c_int), ("far_batch_columns", C.c_int), ("far_big_forms", C.c_int)]


class ImuParamsC(C.Structure):
    _fields_ = [("acc_cov", C.c_double), ("gyro_cov", C.c_double), ("integration_cov", C.c_double),
                ("bias_acc_cov", C.c_double), ("bias_omega_cov", C.c_double),
                ("bias_acc_omega_int", C.c_double)]


class ShardInfoC(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("windows", C.c_int), ("chunks", C.c_int),
                ("sep", C.c_void_p), ("sep_per_chunk", C.c_long),
                ("delta", C.c_void_p), ("delta_count", C.c_long), ("refine_delta", C.c_void_p)]


class GraphOptsC(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("capacity", C.c_int), ("lag", C.c_int), ("iterations", C.c_int), ("device", C.c_int),
                ("prior_sigma", C.c_double * 15), ("rel_tol", C.c_double), ("abs_tol", C.c_double),
                ("cold_start", C.c_int), ("fixed_capacity", C.c_int), ("reference_compat", C.c_int),
                ("relin_threshold", C.c_double), ("incremental", C.c_int), ("wildfire", C.c_double), ("min_model_fidelity", C.c_double),
                ("synchronous_staging", C.c_int), ("max_far_factors", C.c_int)]


CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                       C.POINTER(C.c_double), C.POINTER(C.c_double))

_lib = None

# every symbol include/vilfusion.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "vf_last_error", "vf_version", "vf_device_count",
    "vf_engine_default_opts", "vf_engine_default_opts_sized", "vf_engine_default_tuning", "vf_engine_default_tuning_sized",
    "vf_engine_create", "vf_engine_create_tuned", "vf_engine_destroy",
    "vf_engine_set_range", "vf_engine_set_states", "vf_engine_get_states", "vf_engine_set_imu",
    "vf_engine_set_between", "vf_engine_clear_between", "vf_engine_set_extra_between", "vf_engine_get_extra_between", "vf_engine_get_linear_far", "vf_engine_set_prior",
    "vf_engine_linearize", "vf_engine_assemble", "vf_engine_solve", "vf_engine_retract",
    "vf_engine_decide", "vf_engine_iterate", "vf_engine_slide", "vf_engine_predict",
    "vf_engine_sync", "vf_engine_graph_info", "vf_engine_solve_form",
    "vf_engine_read_imu_lin", "vf_engine_read_between_lin", "vf_engine_read_normal",
    "vf_engine_read_delta", "vf_engine_read_panels", "vf_engine_read_lm",
    "vf_engine_time_stage", "vf_engine_time_iterate", "vf_engine_counts",
    "vf_engine_preintegrate", "vf_engine_get_imu", "vf_engine_ingest_tail", "vf_engine_ingest_status",
    "vf_engine_marginalize", "vf_engine_drop_oldest", "vf_engine_read_marginal", "vf_engine_compact", "vf_engine_grow",
    "vf_engine_set_stream", "vf_engine_set_shard", "vf_engine_shard_info", "vf_engine_solve_local",
    "vf_engine_solve_global", "vf_engine_reset_lambda",
    "vf_engine_refine_count", "vf_engine_refine_begin", "vf_engine_refine_step", "vf_engine_refine_end", "vf_engine_read_refine",
    "vf_engine_gn_begin", "vf_shard_iterate", "vf_shard_gn_step", "vf_shard_exchange_plan", "vf_engine_read_excursions", "vf_engine_close_excursions",
    "vf_chunk_geometry", "vf_shard_range", "vf_engine_set_convergence",
    "vf_engine_isam_step", "vf_engine_predict_from_estimate", "vf_engine_get_estimate", "vf_engine_incremental_info", "vf_engine_set_async", "vf_engine_read_result", "vf_engine_marginalize_ahead",
    "vf_graph_default_opts", "vf_graph_default_opts_sized", "vf_create", "vf_destroy", "vf_add_imu", "vf_reserve_node",
    "vf_add_between", "vf_solve", "vf_get_state", "vf_get_bias", "vf_most_recent_pose_time",
    "vf_set_callback", "vf_graph_staged", "vf_get_trajectory", "vf_get_imu_factor",
    "vf_add_imu_factor", "vf_get_most_recent_estimate", "vf_graph_lm_stats", "vf_graph_solver_info", "vf_set_initial_state", "vf_graph_incremental_info", "vf_graph_get_staged",
    "vf_degeneracy_batch", "vf_degeneracy_spectrum_batch", "vf_dopt_filter_f32",
]


def lib():
    """Load libvilfusion.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise VilFusionError(-7, f"{_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        l = C.CDLL(_SO)
        l.vf_last_error.restype = C.c_char_p
        l.vf_version.restype = C.c_char_p
        l.vf_engine_destroy.restype = None
        l.vf_engine_default_opts.restype = None
        l.vf_engine_default_tuning.restype = None
        l.vf_engine_default_opts_sized.restype = None
        l.vf_engine_default_tuning_sized.restype = None
        l.vf_graph_default_opts_sized.restype = None
        l.vf_graph_default_opts.restype = None
        l.vf_destroy.restype = None
        _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        raise VilFusionError(rc, lib().vf_last_error().decode())
