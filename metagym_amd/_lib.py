"""ctypes binding of libmetagym_hip.so (see include/metagym_hip.h).

The HIP library IS the product: there is no CPU or PyTorch fallback. If the shared object is
missing and cannot be built with hipcc, importing an environment raises.
"""
import ctypes as C
import os

from . import build as _build

ABI_VERSION = 7

MG_OK = 0


class MetaGymHipError(RuntimeError):
    pass


class QuadrotorConfig(C.Structure):
    """mg_quadrotor_config"""
    _fields_ = [
        ("precision", C.c_double), ("quality", C.c_double),
        ("ct0", C.c_double), ("ct1", C.c_double), ("ct2", C.c_double),
        ("mm", C.c_double), ("jm", C.c_double), ("ra", C.c_double), ("phi", C.c_double),
        ("fail_velocity", C.c_double), ("fail_w", C.c_double), ("fail_range", C.c_double),
        ("min_voltage", C.c_double), ("max_voltage", C.c_double),
        ("dt", C.c_double), ("healthy_reward", C.c_double), ("z_offset", C.c_double),
        ("x_offset", C.c_int64), ("y_offset", C.c_int64),
        ("nt", C.c_int32), ("task", C.c_int32),
        ("inertia", C.c_float * 9), ("drag_m", C.c_float * 9), ("drag_f", C.c_float * 9),
        ("gravity_center", C.c_float * 3), ("prop_coord", C.c_float * 12),
        ("map_d", C.c_void_p), ("map_h", C.c_int32), ("map_w", C.c_int32),
        ("velocity_targets_d", C.c_void_p),
    ]


class QuadrotorState(C.Structure):
    """mg_quadrotor_state (device pointers)"""
    _fields_ = [("pos", C.c_void_p), ("vel", C.c_void_p), ("omega", C.c_void_p),
                ("propw", C.c_void_p), ("rot", C.c_void_p), ("ct", C.c_void_p), ("episode", C.c_void_p)]


class QuadrotorAutoReset(C.Structure):
    """mg_quadrotor_autoreset"""
    _fields_ = [("init_velocity", C.c_float * 3), ("init_angular_velocity", C.c_float * 3),
                ("init_velocity_noisy", C.c_double), ("init_angular_velocity_noisy", C.c_double),
                ("seed", C.c_uint64), ("env_id_base", C.c_uint64)]


class QuadrotorPlan(C.Structure):
    """mg_quadrotor_plan (opaque, caller-owned host memory)"""
    _fields_ = [("opaque", C.c_uint64 * 128)]


class MazeTasks(C.Structure):
    """mg_maze_tasks (device pointers)"""
    _fields_ = [("n", C.c_int32), ("n_tasks", C.c_int32), ("start", C.c_void_p), ("goal", C.c_void_p),
                ("walls", C.c_void_p), ("texts", C.c_void_p), ("food_rewards", C.c_void_p),
                ("food_interval", C.c_void_p), ("scalars", C.c_void_p),
                ("food_cells", C.c_void_p), ("n_food", C.c_void_p), ("max_food", C.c_int32),
                ("cell_slot", C.c_void_p), ("slot_food", C.c_void_p), ("slot_interval", C.c_void_p)]


class MazeSampleParams(C.Structure):
    """mg_maze_sample_params"""
    _fields_ = [("n", C.c_int32), ("allow_loops", C.c_int32), ("n_texts", C.c_int32), ("food_interval", C.c_int32),
                ("has_goal_reward", C.c_int32),
                ("cell_size", C.c_double), ("wall_height", C.c_double), ("agent_height", C.c_double),
                ("step_reward", C.c_double), ("goal_reward", C.c_double), ("food_reward", C.c_double),
                ("initial_life", C.c_double), ("max_life", C.c_double),
                ("food_density", C.c_double), ("crowd_ratio", C.c_double)]


class MazeState(C.Structure):
    """mg_maze_state (device pointers)"""
    _fields_ = [("task_id", C.c_void_p), ("grid", C.c_void_p), ("steps", C.c_void_p), ("ori_idx", C.c_void_p),
                ("ori", C.c_void_p), ("loc", C.c_void_p), ("life", C.c_void_p), ("cur_food", C.c_void_p),
                ("wait_refresh", C.c_void_p), ("revival", C.c_void_p),
                ("food_env_stride", C.c_int64), ("food_cell_stride", C.c_int64), ("food_by_slot", C.c_int32)]


class MazeView(C.Structure):
    """mg_maze_view"""
    _fields_ = [("res_h", C.c_int32), ("res_v", C.c_int32), ("max_vision", C.c_double), ("l_focal", C.c_double),
                ("text_size", C.c_double), ("tan_half_fov", C.c_double), ("collision_dist", C.c_double),
                ("col_cos", C.c_void_p), ("col_sin", C.c_void_p), ("ori_sin", C.c_float * 4),
                ("ori_cos", C.c_float * 4), ("textures", C.c_void_p), ("ceil_texture", C.c_void_p),
                ("n_textures", C.c_int32), ("tex_size", C.c_int32), ("max_ray_records", C.c_int32),
                ("obs_format", C.c_int32), ("uniform_cell_size", C.c_double)]


WALKER_MAX_BODIES, WALKER_MAX_JOINTS, WALKER_MAX_SPHERES, WALKER_MAX_FEET = 16, 24, 128, 6
WALKER_MAX_GEOMS, WALKER_MAX_PAIRS = 24, 128


class WalkerTopology(C.Structure):
    """mg_walker_topology"""
    _fields_ = [("n_bodies", C.c_int32), ("n_joints", C.c_int32), ("n_spheres", C.c_int32), ("n_feet", C.c_int32),
                ("body_parent", C.c_int32 * WALKER_MAX_BODIES), ("joint_body", C.c_int32 * WALKER_MAX_JOINTS),
                ("sphere_body", C.c_int32 * WALKER_MAX_SPHERES), ("foot_body", C.c_int32 * WALKER_MAX_FEET),
                ("n_geoms", C.c_int32), ("n_pairs", C.c_int32), ("geom_body", C.c_int32 * WALKER_MAX_GEOMS),
                ("pair_a", C.c_uint8 * WALKER_MAX_PAIRS), ("pair_b", C.c_uint8 * WALKER_MAX_PAIRS),
                ("sphere_foot", C.c_int8 * WALKER_MAX_SPHERES)]


class WalkerModels(C.Structure):
    """mg_walker_models"""
    _fields_ = [("table", C.c_void_p), ("n_tasks", C.c_int32), ("model_stride", C.c_int32)]


class WalkerParams(C.Structure):
    """mg_walker_params"""
    _fields_ = [("time_step", C.c_double), ("frame_skip", C.c_int32), ("solver_iterations", C.c_int32),
                ("erp", C.c_double), ("limit_erp", C.c_double), ("gravity", C.c_double), ("friction", C.c_double),
                ("alive_z", C.c_double), ("alive_bonus", C.c_double), ("dead_bonus", C.c_double),
                ("initial_z", C.c_double), ("joints_at_limit_cost", C.c_double),
                ("walk_target_x", C.c_double), ("walk_target_y", C.c_double),
                ("max_steps", C.c_int32), ("floor_in_parts", C.c_int32), ("mapping", C.c_int32),
                ("self_collision", C.c_int32), ("self_friction", C.c_double),
                ("auto_reset", C.c_int32), ("seed", C.c_uint64), ("step_index", C.c_uint64),
                ("env_id_base", C.c_uint64), ("torque_f32", C.c_int32), ("height_f32", C.c_int32),
                ("actuation", C.c_int32), ("pd_command", C.c_void_p),
                ("pd_kp", C.c_double * WALKER_MAX_JOINTS), ("pd_kd", C.c_double * WALKER_MAX_JOINTS),
                ("pd_strength", C.c_double * WALKER_MAX_JOINTS), ("pd_limit", C.c_double * WALKER_MAX_JOINTS),
                ("substep_log", C.c_void_p), ("n_terrain_boxes", C.c_int32), ("terrain", C.c_void_p),
                ("sphere_friction", C.c_void_p), ("body_linear_damping", C.c_double), ("body_angular_damping", C.c_double),
                ("pd_kp_env", C.c_void_p), ("pd_kd_env", C.c_void_p), ("ext_wrench", C.c_void_p),
                ("max_coordinate_velocity", C.c_double), ("terrain_id", C.c_void_p), ("n_terrain_tables", C.c_int32),
                ("gravity_env", C.c_void_p), ("foot_friction_env", C.c_void_p), ("reset_pos", C.c_void_p), ("reset_rot", C.c_void_p),
                ("contact_margin", C.c_double), ("sphere_margin_in_table", C.c_int32)]


class JointMajor(object):
    """Layout marker for WalkerBatchEnv.reset(joint_noise=JointMajor(t)): `t` is already float64 `[n_joints][num_envs]` on the env's
    device (the kernel's layout) — taken without the transpose-copy a `[num_envs, n_joints]` tensor needs."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class WalkerState(C.Structure):
    """mg_walker_state (device pointers)"""
    _fields_ = [("task_id", C.c_void_p), ("pos", C.c_void_p), ("rot", C.c_void_p), ("vel", C.c_void_p),
                ("omega", C.c_void_p), ("q", C.c_void_p), ("qd", C.c_void_p), ("potential", C.c_void_p),
                ("feet_contact", C.c_void_p), ("steps", C.c_void_p), ("bad_contacts", C.c_void_p), ("foot_force", C.c_void_p)]


A1_NUM_MOTORS, A1_OBS_DIM = 12, 43
A1_MODE_POSITION, A1_MODE_TORQUE, A1_MODE_HYBRID = 1, 2, 3


class A1ActuatorConfig(C.Structure):
    """mg_a1_actuator_config"""
    _fields_ = [("time_step", C.c_double), ("action_repeat", C.c_int32), ("history_len", C.c_int32),
                ("mode", C.c_int32), ("clip_commands", C.c_int32), ("max_angle_change", C.c_double),
                ("control_latency", C.c_double), ("pd_latency", C.c_double),
                ("control_latency_env", C.c_void_p), ("pd_latency_env", C.c_void_p),
                ("kp", C.c_double * A1_NUM_MOTORS), ("kd", C.c_double * A1_NUM_MOTORS),
                ("kp_env", C.c_void_p), ("kd_env", C.c_void_p),
                ("strength", C.c_double * A1_NUM_MOTORS), ("torque_limit", C.c_double * A1_NUM_MOTORS),
                ("has_torque_limit", C.c_int32)]


class A1ActuatorState(C.Structure):
    """mg_a1_actuator_state (device pointers)"""
    _fields_ = [("history", C.c_void_p), ("count", C.c_void_p), ("head", C.c_void_p),
                ("observed_torque", C.c_void_p), ("control_obs", C.c_void_p)]


A1_ETG_MAX_H, A1_MAX_SEGMENTS = 32, 32
A1_EXTRA_ETG, A1_EXTRA_ETG_OBS, A1_EXTRA_YAW = 1, 2, 4
WALKER_BOX_DOUBLES = 16


class A1EtgConfig(C.Structure):
    """mg_a1_etg_config"""
    _fields_ = [("enabled", C.c_int32), ("H", C.c_int32), ("T", C.c_double), ("T2_ratio", C.c_double),
                ("sigma_sq", C.c_double), ("amp", C.c_double), ("phase", C.c_double * 2), ("omega", C.c_double),
                ("u", (C.c_double * 2) * A1_ETG_MAX_H), ("w", (C.c_double * A1_ETG_MAX_H) * 3), ("b", C.c_double * 3),
                ("act_mode_pose", C.c_int32), ("gallop", C.c_int32), ("etg_weight", C.c_double),
                ("action_space", C.c_int32), ("pose", C.c_double * A1_NUM_MOTORS)]


class A1RewardConfig(C.Structure):
    """mg_a1_reward_config"""
    _fields_ = [("w_torso", C.c_double), ("w_up", C.c_double), ("w_feet", C.c_double), ("w_tau", C.c_double),
                ("w_badfoot", C.c_double), ("w_footcontact", C.c_double), ("reward_p", C.c_double), ("vel_d", C.c_double),
                ("cw_half", C.c_double), ("cw_04", C.c_double), ("n_segments", C.c_int32),
                ("seg", (C.c_double * 5) * A1_MAX_SEGMENTS), ("vel_mode", C.c_int32),
                ("seg_table", C.c_void_p), ("seg_count", C.c_void_p), ("terrain_id", C.c_void_p)]


class A1RewardState(C.Structure):
    """mg_a1_reward_state (device pointers)"""
    _fields_ = [("last_base", C.c_void_p), ("last_base10", C.c_void_p), ("last_foot", C.c_void_p), ("vd2", C.c_void_p),
                ("steps", C.c_void_p)]


A1_SENSOR_OBS_DIM = 37


class A1SensorConfig(C.Structure):
    """mg_a1_sensor_config"""
    _fields_ = [("normal", C.c_int32), ("disp_dt", C.c_double), ("motor_dt", C.c_double)]


class A1SensorState(C.Structure):
    """mg_a1_sensor_state (device pointers)"""
    _fields_ = [("base_last", C.c_void_p), ("base_cur", C.c_void_p), ("yaw", C.c_void_p), ("first_rpy", C.c_void_p),
                ("last_angle", C.c_void_p), ("first", C.c_void_p), ("noise", C.c_void_p)]


A1_FILTER_MAX_HIST = 4


class A1FilterConfig(C.Structure):
    """mg_a1_filter_config"""
    _fields_ = [("hist_len", C.c_int32), ("a", (C.c_double * (A1_FILTER_MAX_HIST + 1)) * A1_NUM_MOTORS),
                ("b", (C.c_double * (A1_FILTER_MAX_HIST + 1)) * A1_NUM_MOTORS)]


# symbol -> (restype, argtypes); tests/test_abi.py checks this list against include/metagym_hip.h
_P = C.c_void_p
SIGNATURES = {
    "mg_abi_version": (C.c_int, []),
    "mg_last_error": (C.c_char_p, []),
    "mg_target_arch": (C.c_char_p, []),
    "mg_selftest_philox": (C.c_int, [_P, _P, C.c_int32, _P]),
    "mg_quadrotor_default_config": (C.c_int, [C.POINTER(QuadrotorConfig)]),
    "mg_quadrotor_velocity_targets": (C.c_int, [C.POINTER(QuadrotorConfig), C.c_int32, _P, _P, _P]),
    "mg_quadrotor_reset": (C.c_int, [C.POINTER(QuadrotorConfig), C.c_int32, C.POINTER(QuadrotorState),
                                     _P, _P, _P, _P, _P]),
    "mg_quadrotor_step": (C.c_int, [C.POINTER(QuadrotorConfig), C.c_int32, C.POINTER(QuadrotorState),
                                    _P, _P, _P, _P, _P, _P, _P]),
    "mg_quadrotor_step_autoreset": (C.c_int, [C.POINTER(QuadrotorConfig), C.c_int32, C.c_int32,
                                              C.POINTER(QuadrotorState), C.POINTER(QuadrotorAutoReset),
                                              _P, _P, _P, _P, _P, _P, _P]),
    "mg_quadrotor_plan_init": (C.c_int, [C.POINTER(QuadrotorPlan), C.POINTER(QuadrotorConfig),
                                         C.POINTER(QuadrotorAutoReset), C.c_int32, C.POINTER(QuadrotorState)]),
    "mg_quadrotor_plan_step": (C.c_int, [C.POINTER(QuadrotorPlan), C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "mg_quadrotor_rollout": (C.c_int, [C.POINTER(QuadrotorConfig), C.c_int32, C.c_int32,
                                       C.POINTER(QuadrotorState), _P, _P, _P, _P, _P, _P, _P]),
    "mg_maze_view_tables": (C.c_int, [C.c_int32, C.c_double, C.c_double, _P, _P]),
    "mg_maze_sample_tasks": (C.c_int, [C.POINTER(MazeSampleParams), C.c_int32, C.c_uint32, _P, _P, _P, _P, _P, _P,
                                       _P, _P, _P]),
    "mg_maze_reset": (C.c_int, [C.POINTER(MazeTasks), C.c_int32, C.c_int32, C.POINTER(MazeState), _P, _P]),
    "mg_maze2d_step": (C.c_int, [C.POINTER(MazeTasks), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 C.POINTER(MazeState), _P, _P, _P, _P, _P, _P]),
    "mg_maze_check_uniform_cell_size": (C.c_int, [C.POINTER(MazeTasks), C.c_double, _P]),
    "mg_maze_forget_tasks": (C.c_int, [C.POINTER(MazeTasks)]),
    "mg_maze3d_step": (C.c_int, [C.POINTER(MazeTasks), C.POINTER(MazeView), C.c_int32, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_int32, C.POINTER(MazeState), _P, _P, _P, _P, _P, _P]),
    "mg_walker_reset": (C.c_int, [C.POINTER(WalkerTopology), C.POINTER(WalkerModels), C.POINTER(WalkerParams),
                                  C.c_int32, C.POINTER(WalkerState), _P, _P, _P, _P]),
    "mg_walker_step": (C.c_int, [C.POINTER(WalkerTopology), C.POINTER(WalkerModels), C.POINTER(WalkerParams),
                                 C.c_int32, C.POINTER(WalkerState), _P, _P, _P, _P, _P, _P]),
    "mg_a1_apply_action": (C.c_int, [C.POINTER(A1ActuatorConfig), C.c_int32, C.POINTER(A1ActuatorState), _P, _P,
                                     C.c_double, _P, _P]),
    "mg_a1_receive_observation": (C.c_int, [C.POINTER(A1ActuatorConfig), C.c_int32, C.POINTER(A1ActuatorState),
                                            _P, _P, _P, _P, _P, _P]),
    "mg_a1_receive_and_apply": (C.c_int, [C.POINTER(A1ActuatorConfig), C.c_int32, C.POINTER(A1ActuatorState), _P, _P, _P, _P,
                                          _P, _P, C.c_double, _P, _P]),
    "mg_a1_receive_log": (C.c_int, [C.POINTER(A1ActuatorConfig), C.c_int32, C.POINTER(A1ActuatorState), _P, C.c_int32, _P]),
    "mg_a1_sensors": (C.c_int, [C.POINTER(A1ActuatorConfig), C.c_int32, C.POINTER(A1ActuatorState),
                                _P, _P, _P, _P, _P, _P]),
    "mg_a1_info": (C.c_int, [C.POINTER(A1ActuatorConfig), C.c_int32, C.POINTER(A1ActuatorState), _P, _P, _P, _P, _P, _P, _P]),
    "mg_a1_etg_action": (C.c_int, [C.POINTER(A1EtgConfig), C.c_int32, _P, _P, _P, _P, _P, _P]),
    "mg_a1_reward_reset": (C.c_int, [C.POINTER(A1RewardConfig), C.c_int32, C.POINTER(A1RewardState), _P, _P, _P, _P, _P]),
    "mg_a1_observation": (C.c_int, [C.POINTER(A1SensorConfig), C.c_int32, C.POINTER(A1SensorState), _P, _P, _P, _P, _P, _P,
                                    _P, _P]),
    "mg_a1_observation_extras": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "mg_a1_action_filter": (C.c_int, [C.POINTER(A1FilterConfig), C.c_int32, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "mg_a1_reward_step": (C.c_int, [C.POINTER(A1RewardConfig), C.c_int32, C.POINTER(A1RewardState), _P, _P, _P, _P, _P,
                                    _P, _P, _P, _P, _P, _P, _P]),
}

_lib = None


def lib_path():
    return os.environ.get("METAGYM_HIP_LIB", _build.LIB_PATH)


def load():
    """Load (building first if the in-tree .so is missing or stale and hipcc exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if "METAGYM_HIP_LIB" not in os.environ and _build.is_stale():
        if os.path.exists(_build.HIPCC):
            _build.build(verbose=False)
        elif not os.path.exists(path):
            raise MetaGymHipError(
                "libmetagym_hip.so is not built and hipcc (%s) is not available. Run "
                "`python -m metagym_amd.build` on a ROCm machine. There is no CPU fallback." % _build.HIPCC)
    # torch-ROCm ships its own libamdhip64 / libhsa-runtime64 with the same SONAME as /opt/rocm's. The
    # process must end up with ONE HIP runtime, and it has to be the one that owns the torch tensors we are
    # handed: import torch first so the library binds to that copy (loading ours first pulls /opt/rocm's
    # runtime in and torch then fails with "no ROCm-capable device").
    import torch  # noqa: F401
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise MetaGymHipError("cannot load %s: %s (no CPU fallback exists)" % (path, e)) from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.mg_abi_version() != ABI_VERSION:
        raise MetaGymHipError("ABI mismatch: library %d, binding %d" % (lib.mg_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != MG_OK:
        msg = load().mg_last_error().decode("utf-8", "replace")
        raise MetaGymHipError("%s failed (%d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL). The tensor must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor handed to the C ABI must be contiguous"
    return C.c_void_p(t.data_ptr())


def canonical_device(device):
    """torch.device(device) with an explicit index for a CUDA device ("cuda" -> the current device, cuda:0 unless the caller
    changed it): tensors report `cuda:0`, so an env that kept the index-less spelling would see every tensor it is handed as
    living "elsewhere" (`t.device == self.device` is False) — a slow path in Quadrotor.step, a refusal in A1Actuators."""
    import torch
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return d


def current_stream(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
