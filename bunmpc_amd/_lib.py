"""ctypes binding of libbunmpc_hip.so (include/bunmpc.h).  There is no CPU fallback:
if the library is missing and cannot be built, or a call needs a GPU that is not
there, this raises."""
import ctypes as C
import os

from . import build as _build

OK, BAD_ARG, DIVERGED, DEVICE_ERROR = 0, 1, 2, 3
L0_X, L0_F = 2.25e6, 506.25
NSTATS = 6


class BmpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bunmpc status %d: %s" % (code, msg))
        self.code = code


class Batch(C.Structure):
    """bmpc_batch_t"""
    _fields_ = ([("B", C.c_int), ("n_col", C.c_int), ("n_eff", C.c_int), ("raw", C.c_int),
                 ("num_iters", C.c_int), ("maxit", C.c_int), ("cold_start", C.c_int),
                 ("precision", C.c_int)] +
                [(n, C.c_double) for n in ("m", "rho", "mu", "beta", "tol", "exit_tol")] +
                [(n, C.c_void_p) for n in ("cnt_plan", "dt", "x_init", "W_X", "W_X_ter", "W_F",
                                           "bounds", "X_nom", "X_ter")] +
                [(n, C.c_long) for n in ("sW_X", "sW_X_ter", "sW_F", "sbounds")] +
                [(n, C.c_void_p) for n in ("Qx", "qx", "lbx", "ubx", "Qf", "qf", "X", "F", "P",
                                           "L_x", "L_f", "dyn_viol", "hist", "stats", "trace")])


class GaitParams(C.Structure):
    """bmpc_gait_params_t"""
    _fields_ = [("gait_period", C.c_double), ("gait_dt", C.c_double), ("gait_horizon", C.c_double), ("nom_ht", C.c_double),
                ("stance_percent", C.c_double * 4), ("phase_offset", C.c_double * 4), ("ori_correction", C.c_double * 3),
                ("offsets_xy", (C.c_double * 2) * 4)]


class PlanBatch(C.Structure):
    """bmpc_plan_batch_t"""
    _fields_ = ([("B", C.c_int), ("n_col", C.c_int), ("n_gaits", C.c_int), ("reserved_", C.c_int)] +
                [(n, C.c_void_p) for n in ("gaits", "gait_id", "t0", "com", "feet0", "v_des", "w_des", "x_init", "amom",
                                           "hip_off", "cnt_plan", "swing_time", "dt", "X_nom", "X_ter")])


class WbPlanBatch(C.Structure):
    """bmpc_wb_plan_batch_t"""
    _fields_ = ([("B", C.c_int), ("n_col", C.c_int), ("ik_col", C.c_int), ("reserved_", C.c_int), ("model", C.c_void_p),
                 ("gait", C.c_void_p), ("foot_frame", C.c_int * 4), ("step_ht", C.c_double), ("swing_wt", C.c_double * 2),
                 ("cent_wt", C.c_double * 2), ("reg_wt", C.c_double * 2)] +
                [(n, C.c_void_p) for n in ("x", "t0", "v_des_body", "com", "feet0", "v_des", "w_des", "hip_off", "amom", "x_init",
                                           "cnt_plan", "swing_time", "dt", "X_nom", "X_ter", "ik_tasks")])


class InterpBatch(C.Structure):
    """bmpc_interp_batch_t"""
    _fields_ = [("B", C.c_int), ("n_knots", C.c_int), ("width", C.c_int), ("size", C.c_int), ("max_rows", C.c_int),
                ("dt_stride", C.c_int), ("step", C.c_double), ("knots", C.c_void_p), ("dt", C.c_void_p), ("out", C.c_void_p),
                ("rows", C.c_void_p)]


class IdBatch(C.Structure):
    """bmpc_id_batch_t"""
    _fields_ = ([("n", C.c_long), ("model", C.c_void_p), ("foot_frame", C.c_int * 4), ("kp", C.c_double * 12),
                 ("kd", C.c_double * 12)] +
                [(n, C.c_void_p) for n in ("q", "v", "q_des", "v_des", "a_des", "f")] +
                [(n, C.c_long) for n in ("s_q", "s_v", "s_q_des", "s_v_des", "s_a_des", "s_f")] +
                [(n, C.c_void_p) for n in ("tau_ff", "tau_fb", "action", "state")])


class PerturbBatch(C.Structure):
    """bmpc_perturb_batch_t"""
    _fields_ = ([("B", C.c_int), ("K", C.c_int), ("model", C.c_void_p), ("foot_frame", C.c_int * 4), ("mu", C.c_double * 4),
                 ("sigma", C.c_double * 4), ("q", C.c_void_p), ("v", C.c_void_p), ("contact", C.c_void_p),
                 ("s_contact_b", C.c_long), ("s_contact_e", C.c_long)] +
                [(n, C.c_void_p) for n in ("z", "q_out", "v_out", "chosen")])


_lib = None

_D = C.c_double
_I = C.c_int
_P = C.c_void_p

_SIGS = {
    "bmpc_abi_version": (_I, []),
    "bmpc_set_three_per_wave": (_I, [_I]),
    "bmpc_set_work_stealing": (_I, [_I]),
    "bmpc_set_two_waves_per_simd": (_I, [_I]),
    "bmpc_set_steal_grid": (_I, [_I]),
    "bmpc_biconvex_last_waves_per_simd": (_I, []),
    "bmpc_biconvex_last_lanes_per_problem": (_I, []),
    "bmpc_batch_struct_size": (_I, []),
    "bmpc_set_latency_mapping_max_batch": (_I, [_I]),
    "bmpc_set_exact_step_decisions": (_I, [_I]),
    "bmpc_biconvex_fp32_scratch_bytes": (_I, []),
    "bmpc_last_error": (C.c_char_p, []),
    "bmpc_device_count": (_I, [_P]),
    "bmpc_set_device": (_I, [_I]),
    "bmpc_selftest_lanes": (_I, []),
    "bmpc_gait_create": (_P, [_D, _P, _P, _I, _D]),
    "bmpc_gait_destroy": (None, [_P]),
    "bmpc_gait_n_eff": (_I, [_P]),
    "bmpc_gait_get_phase": (_I, [_P, _D, _I, _P]),
    "bmpc_gait_get_phase_all": (_I, [_P, _D, _P]),
    "bmpc_gait_get_phi": (_I, [_P, _D, _I, _P]),
    "bmpc_gait_get_phi_all": (_I, [_P, _D, _P]),
    "bmpc_gait_get_percent_in_phase": (_I, [_P, _D, _I, _P]),
    "bmpc_gait_get_percent_in_phase_all": (_I, [_P, _D, _P]),
    "bmpc_gait_get_contact_phase_plan": (_I, [_P, _I, _D, _D, _P]),
    "bmpc_gait_set_step_height": (_I, [_P, _D]),
    "bmpc_gait_set_stance_percent": (_I, [_P, _D, _D, _D, _D]),
    "bmpc_biconvex_create": (_P, [_D, _I, _I]),
    "bmpc_biconvex_destroy": (None, [_P]),
    "bmpc_biconvex_n_col": (_I, [_P]),
    "bmpc_biconvex_n_eff": (_I, [_P]),
    "bmpc_biconvex_set_contact_plan": (_I, [_P, _P, _D]),
    "bmpc_biconvex_set_rotation_matrix_f": (_I, [_P, _P]),
    "bmpc_biconvex_return_A_x": (_I, [_P, _P, _P]),
    "bmpc_biconvex_return_b_x": (_I, [_P, _P, _P]),
    "bmpc_biconvex_return_A_f": (_I, [_P, _P, _P, _P]),
    "bmpc_biconvex_return_b_f": (_I, [_P, _P, _P, _P]),
    "bmpc_biconvex_set_cost_x": (_I, [_P, _P, _P]),
    "bmpc_biconvex_set_cost_f": (_I, [_P, _P, _P]),
    "bmpc_biconvex_create_cost_X": (_I, [_P, _P, _P, _P, _P]),
    "bmpc_biconvex_create_cost_F": (_I, [_P, _P]),
    "bmpc_biconvex_set_bounds_x": (_I, [_P, _P, _P]),
    "bmpc_biconvex_set_bounds_f": (_I, [_P, _P, _P]),
    "bmpc_biconvex_create_bound_constraints": (_I, [_P, _P, _I, _I, _D, _D, _D]),
    "bmpc_biconvex_set_rho": (_I, [_P, _D]),
    "bmpc_biconvex_return_opt_x": (_I, [_P, _P]),
    "bmpc_biconvex_return_opt_f": (_I, [_P, _P]),
    "bmpc_biconvex_return_opt_p": (_I, [_P, _P]),
    "bmpc_biconvex_return_opt_com": (_I, [_P, _P]),
    "bmpc_biconvex_return_opt_mom": (_I, [_P, _P]),
    "bmpc_biconvex_set_warm_start_vars": (_I, [_P, _P, _P, _P]),
    "bmpc_biconvex_optimize": (_I, [_P, _P, _I]),
    "bmpc_biconvex_dyn_viol_hist_size": (_I, [_P]),
    "bmpc_biconvex_return_dyn_viol_hist": (_I, [_P, _P]),
    "bmpc_biconvex_collect_statistics": (_I, [_P]),
    "bmpc_biconvex_get_step_constants": (_I, [_P, _P, _P]),
    "bmpc_biconvex_set_step_constants": (_I, [_P, _D, _D]),
    "bmpc_biconvex_last_stats": (_I, [_P, _P]),
    "bmpc_biconvex_set_friction_coefficient": (_I, [_P, _D]),
    "bmpc_biconvex_set_robot_mass": (_I, [_P, _D]),
    "bmpc_batch_defaults": (None, [_P]),
    "bmpc_biconvex_solve_batch_device": (_I, [_P, _P]),
    "bmpc_biconvex_solve_batch_host": (_I, [_P]),
    "bmpc_biconvex_kernel_name": (C.c_char_p, [_I, _I]),
    "bmpc_biconvex_last_kernel_name": (C.c_char_p, []),
    "bmpc_plan_batch_device": (_I, [_P, _P]),
    "bmpc_wb_plan_batch_device": (_I, [_P, _P]),
    "bmpc_interp_batch_device": (_I, [_P, _P]),
    "bmpc_id_batch_device": (_I, [_P, _P]),
    "bmpc_perturb_batch_device": (_I, [_P, _P]),
    "bmpc_ik_set_speculative_below": (_I, [_I]),
    "bmpc_ik_set_spec_one_wave_above": (_I, [_I]),
    "bmpc_model_create": (_P, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "bmpc_model_destroy": (None, [_P]),
    "bmpc_model_total_mass": (_D, [_P]),
    "bmpc_ik_create": (_P, [_P, _I]),
    "bmpc_ik_destroy": (None, [_P]),
    "bmpc_ik_n_col": (_I, [_P]),
    "bmpc_ik_setup_costs": (_I, [_P, _P, _I]),
    "bmpc_ik_optimize": (_I, [_P, _P]),
    "bmpc_ik_get_xs": (_I, [_P, _P]),
    "bmpc_ik_get_us": (_I, [_P, _P]),
    "bmpc_ik_return_opt_com": (_I, [_P, _P]),
    "bmpc_ik_return_opt_mom": (_I, [_P, _P]),
    "bmpc_ik_add_position_tracking_task": (_I, [_P, _I, _I, _I, _P, _D, C.c_char_p]),
    "bmpc_ik_add_position_tracking_task_single": (_I, [_P, _I, _P, _D, C.c_char_p, _I]),
    "bmpc_ik_add_terminal_position_tracking_task": (_I, [_P, _I, _P, _D, C.c_char_p]),
    "bmpc_ik_add_velocity_tracking_task": (_I, [_P]),
    "bmpc_ik_add_com_position_tracking_task": (_I, [_P, _I, _I, _P, _I, _D, C.c_char_p, _I]),
    "bmpc_ik_add_centroidal_momentum_tracking_task": (_I, [_P, _I, _I, _P, _I, _D, C.c_char_p, _I]),
    "bmpc_ik_add_state_regularization_cost": (_I, [_P, _I, _I, _D, C.c_char_p, _P, _P, _I]),
    "bmpc_ik_add_state_regularization_cost_single": (_I, [_P, _I, _D, C.c_char_p, _P, _P]),
    "bmpc_ik_add_ctrl_regularization_cost": (_I, [_P, _I, _I, _D, C.c_char_p, _P, _P, _I]),
    "bmpc_ik_add_ctrl_regularization_cost_single": (_I, [_P, _I, _D, C.c_char_p, _P, _P]),
    "bmpc_ik_last_stats": (_I, [_P, _P, _P, _P, _P]),
    "bmpc_ik_workspace_doubles": (_I, [_I]),
    "bmpc_ik_layout": (None, [_I, _P]),
    "bmpc_ik_layout_trace": (None, [_I, _P, _P, _P]),
    "bmpc_ik_selftest_state_ops": (_I, [_P, _P, _P, _I, _P, _P, _P, _P]),
    "bmpc_ik_set_profile": (_I, [_I]),
    "bmpc_ik_set_all_steps": (_I, [_I]),
    "bmpc_ik_set_gains_wave_below": (_I, [_I]),
    "bmpc_ik_set_blocking_waits": (_I, [_I]),
    "bmpc_ik_set_express_capacity": (_I, [_I]),
    "bmpc_ik_set_fused_direct_max": (_I, [_I]),
    "bmpc_ik_set_express_near": (C.c_double, [C.c_double]),
    "bmpc_ik_batch_struct_size": (_I, []),
    "bmpc_ik_kernel_occupancy": (None, [_P]),
    "bmpc_ik_active_list_ints": (C.c_long, [C.c_long]),
    "bmpc_ik_last_profile": (None, [_P]),
    "bmpc_ik_solve_batch_device": (_I, [_P, _P]),
    "bmpc_ik_centroidal_state_device": (_I, [_P, _P, _P, _I, _P]),
    "bmpc_kinodyn_create": (_P, [_P, _D, _I, _I, _I]),
    "bmpc_kinodyn_destroy": (None, [_P]),
    "bmpc_kinodyn_return_dyn": (_P, [_P]),
    "bmpc_kinodyn_return_ik": (_P, [_P]),
    "bmpc_kinodyn_optimize": (_I, [_P, _P, _P, _I, _I]),
    "bmpc_kinodyn_set_com_tracking_weight": (_I, [_P, _D]),
    "bmpc_kinodyn_set_mom_tracking_weight": (_I, [_P, _D]),
    "bmpc_kinodyn_compute_solve_times": (_I, [_P]),
    "bmpc_kinodyn_return_solve_times": (_I, [_P, _P]),
    "bmpc_kinodyn_solve_batch_device": (_I, [_P, _P]),
}

IK_NODE_TASK_DOUBLES = 33


class IkSched(C.Structure):
    """bmpc_ik_sched_t: per-batch scheduling thresholds (0 = process default, < 0 = never)"""
    _fields_ = [("spec_below", C.c_int), ("all_steps_below", C.c_int), ("gains_wave_below", C.c_int), ("express_cap", C.c_int), ("debug_inject", C.c_int)]


class IkBatch(C.Structure):
    """bmpc_ik_batch_t"""
    _fields_ = ([("B", C.c_int), ("n_col", C.c_int), ("maxiter", C.c_int), ("model", C.c_void_p)] +
                [(n, C.c_void_p) for n in ("x0", "dt", "tasks", "state_w", "x_reg", "ctrl_w")] +
                [("s_state_w", C.c_long), ("s_ctrl_w", C.c_long), ("ws", C.c_void_p), ("active", C.c_void_p),
                 ("iters_run", C.c_void_p), ("s_x_reg", C.c_long), ("sn_state_w", C.c_long), ("sn_x_reg", C.c_long),
                 ("sn_ctrl_w", C.c_long), ("active_list", C.c_void_p), ("sched", IkSched)])


class KinoDynBatch(C.Structure):
    """bmpc_kinodyn_batch_t"""
    _fields_ = [("dyn", Batch), ("ik", IkBatch), ("x", C.c_void_p)]


def exported_symbols():
    """Every symbol include/bunmpc.h declares (checked by the CPU tests)."""
    return sorted(_SIGS)


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP
    runtimes in one process do not work (the second sees no device), so when torch is installed
    bind this library to torch's copy whichever of the two is imported first."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        _preload_hip_runtime()
        path = os.environ.get("BUNMPC_LIB")     # side-by-side experiment builds (bunmpc_amd/build.py)
        if not path:
            path = _build.LIB
            if not os.path.exists(path) or (_build.is_stale() and os.path.exists(_build.HIPCC)):
                path = _build.build()
        handle = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.bmpc_batch_struct_size() != C.sizeof(Batch):
            raise ImportError("bmpc_batch_t layout differs between include/bunmpc.h and bunmpc_amd/_lib.py")
        if handle.bmpc_ik_batch_struct_size() != C.sizeof(IkBatch):
            raise ImportError("bmpc_ik_batch_t layout differs between include/bunmpc.h and bunmpc_amd/_lib.py")
        _lib = handle
    return _lib


def check(code):
    if code != OK:
        raise BmpcError(code, lib().bmpc_last_error().decode())
    return code


def last_error():
    return lib().bmpc_last_error().decode()
