"""ctypes binding of libemloco_hip.so (the C ABI declared in include/*.h).

There is no CPU fallback: if the gfx950 library is missing or no MI355X is visible, every entry
point that needs the device raises `EmlocoError`.  Build the library with `python -m emloco_amd.build`.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMLOCO_LIB") or os.path.join(HERE, "lib", "libemloco_hip.so")     # (EMLOCO_LIB: another build of the library, for A/B runs)

NB, NDOF, MAXC, MAXCAND = 24, 69, 20, 96
SELF_OBS, TRAJ_SAMPLES, TRAJ_VERTS, HEIGHT_POINTS = 368, 15, 101, 1024
TASK_OBS = 2 * TRAJ_SAMPLES + HEIGHT_POINTS
OBS = SELF_OBS + TASK_OBS
AMP_ROW, AMP_STEPS = 206, 15

T_ROOT_STATE, T_DOF_STATE, T_RIGID_BODY, T_CONTACT_FORCE, T_DOF_FORCE, T_PD_TARGET, T_WARM_START = range(7)
POST_ADVANCE, POST_OBS, POST_REWARD, POST_RESET, POST_AMP_SHIFT, POST_AMP_ROW = 1, 2, 4, 8, 16, 32
POST_STEP = 63
POST_SKIP_DONE = 64
POST_AMP_DONE_ONLY = 128


class EmlocoError(RuntimeError):
    pass


class SimParams(C.Structure):
    """EmlocoSimParams (include/emloco_sim.h)."""
    _fields_ = [("n_sub", C.c_int32), ("n_iter", C.c_int32), ("h", C.c_float), ("gravity_z", C.c_float),
                ("contact_offset", C.c_float), ("erp", C.c_float), ("max_depen_vel", C.c_float),
                ("mu", C.c_float), ("ang_damping", C.c_float), ("max_ang_vel", C.c_float),
                ("ground_z", C.c_float), ("cfm", C.c_float), ("warm", C.c_float), ("drive_mode", C.c_int32)]


class ModelDesc(C.Structure):
    """EmlocoModelDesc (include/emloco_sim.h)."""
    _fields_ = [("n_env", C.c_int32), ("parent", C.POINTER(C.c_int32)), ("geom_type", C.POINTER(C.c_int32)),
                ("joint_off", C.POINTER(C.c_float)), ("mass", C.POINTER(C.c_float)), ("com", C.POINTER(C.c_float)),
                ("inertia", C.POINTER(C.c_float)), ("geom_a", C.POINTER(C.c_float)), ("geom_b", C.POINTER(C.c_float)),
                ("geom_r", C.POINTER(C.c_float)), ("kp", C.POINTER(C.c_float)), ("kd", C.POINTER(C.c_float)),
                ("armature", C.POINTER(C.c_float)), ("effort", C.POINTER(C.c_float))]


class SelfCollisionDesc(C.Structure):
    """EmlocoSelfCollisionDesc (include/emloco_sim.h)."""
    _fields_ = [("n_pairs", C.c_int32), ("pairs", C.POINTER(C.c_uint8)), ("cap_a", C.POINTER(C.c_float)),
                ("cap_b", C.POINTER(C.c_float)), ("cap_r", C.POINTER(C.c_float)), ("k", C.c_float), ("c", C.c_float),
                ("max_pen", C.c_float), ("mu", C.c_float), ("n_seg", C.c_int32), ("seg_body", C.POINTER(C.c_uint8))]


class TaskBufs(C.Structure):
    """EmlocoTaskBufs (include/emloco_task.h); pointers are raw device addresses."""
    _fields_ = [("n_env", C.c_int32), ("hf_rows", C.c_int32), ("hf_cols", C.c_int32), ("head_body", C.c_int32),
                ("n_dof_subset", C.c_int32),
                ("dt", C.c_float), ("traj_dur", C.c_float), ("sample_dt", C.c_float), ("hscale", C.c_float),
                ("vscale", C.c_float), ("power_coef", C.c_float), ("fail_dist", C.c_float),
                ("max_episode_length", C.c_float),
                ("rb_state", C.c_void_p), ("dof_state", C.c_void_p), ("dof_force", C.c_void_p),
                ("contact_force", C.c_void_p),
                ("betas", C.c_void_p), ("traj_verts", C.c_void_p), ("heightfield", C.c_void_p),
                ("left_to_right", C.c_void_p), ("contact_body_mask", C.c_void_p), ("key_bodies", C.c_void_p),
                ("dof_subset", C.c_void_p),
                ("progress_buf", C.c_void_p), ("reset_buf", C.c_void_p), ("terminate_buf", C.c_void_p),
                ("obs_buf", C.c_void_p), ("flip_obs_buf", C.c_void_p), ("rew_buf", C.c_void_p),
                ("reward_raw", C.c_void_p), ("amp_obs_buf", C.c_void_p), ("amp_ring", C.c_int32)]


class ResetBufs(C.Structure):
    """EmlocoResetBufs (include/emloco_task.h)."""
    _fields_ = [("flags", C.c_int32), ("n_motions", C.c_int32), ("n_real", C.c_int32), ("n_valid", C.c_int32),
                ("n_dof_subset", C.c_int32), ("hf_rows", C.c_int32), ("hf_cols", C.c_int32),
                ("fixed_x", C.c_float), ("fixed_y", C.c_float), ("dt", C.c_float), ("height_tolerance", C.c_float),
                ("vert_dt", C.c_float), ("dtheta_max", C.c_float), ("speed_min", C.c_float), ("speed_max", C.c_float),
                ("accel_max", C.c_float), ("sharp_prob", C.c_float), ("hybrid_prob", C.c_float),
                ("traj_dur", C.c_float), ("sample_dt", C.c_float), ("hscale", C.c_float), ("vscale", C.c_float),
                ("gts", C.c_void_p), ("grs", C.c_void_p), ("lrs", C.c_void_p), ("gvs", C.c_void_p), ("gavs", C.c_void_p),
                ("dvs", C.c_void_p), ("motion_len", C.c_void_p), ("motion_dt", C.c_void_p), ("motion_nframes", C.c_void_p),
                ("motion_start", C.c_void_p), ("real_traj", C.c_void_p), ("heightfield", C.c_void_p),
                ("valid_x", C.c_void_p), ("valid_y", C.c_void_p), ("betas", C.c_void_p), ("key_bodies", C.c_void_p),
                ("dof_subset", C.c_void_p), ("traj_verts", C.c_void_p), ("inverted", C.c_void_p),
                ("progress_buf", C.c_void_p), ("reset_buf", C.c_void_p), ("terminate_buf", C.c_void_p),
                ("waypoint_traj", C.c_void_p), ("init_pose", C.c_void_p), ("init_vel", C.c_void_p),
                ("amp_obs_buf", C.c_void_p), ("motion_ids", C.c_void_p), ("motion_times", C.c_void_p), ("ground_h", C.c_void_p),
                ("real_pick", C.c_void_p), ("real_pick_key", C.c_uint32), ("amp_ring", C.c_int32)]


RESET_RND = 512
RESET_RANDOM_HEADING, RESET_INIT_HEADING, RESET_HEADING_INVERSION, RESET_ADJUST_ROOT_VEL, RESET_REAL_PATH, RESET_FIXED_LOCATION = 1, 2, 4, 8, 16, 32
RESET_NO_AMP_HISTORY = 64
RND_MOTION, RND_TIME, RND_YAW, RND_SPEED, RND_LOC, RND_REAL, RND_REAL_PICK, RND_INVERSION, RND_HEADING, RND_SPEED0 = range(10)
RND_DTHETA, RND_SHARP, RND_BERN, RND_DSPEED = 16, 116, 216, 316


def _fmix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def real_pick_perm(i, n, key):
    """Row of the real-path table list entry i takes (restates reset_kernels.hip: real_pick_perm, the keyed bijection of
    [0, n) that stands where the reference calls random.sample, traj_generator.py:132)."""
    bits = 2
    while (1 << bits) < n:
        bits += 2
    half = bits >> 1
    mask = (1 << half) - 1
    x = i % n
    while True:
        l, r = x >> half, x & mask
        for rnd in range(4):
            f = _fmix32((r * 0x9E3779B1 + key + rnd * 0x85EBCA6B) & 0xFFFFFFFF) & mask
            l, r = r, l ^ f
        x = (l << half) | r
        if x < n:
            return x


POOL_FLOATS = 512              # EMLOCO_POOL_FLOATS


class ResetPool(C.Structure):   # EmlocoResetPool
    _fields_ = [("k", C.c_int32), ("cur", C.c_void_p), ("cur_tag", C.c_void_p), ("next", C.c_void_p), ("next_tag", C.c_void_p),
                ("next_seed", C.c_uint64)]


def default_sim_params(**kw):
    """Engine parameters of pacer.yaml:93-104 / config.py:143-163 mapped onto EmlocoSimParams."""
    p = dict(n_sub=2, n_iter=4, h=(1.0 / 60.0) / 2, gravity_z=-9.81, contact_offset=0.02, erp=0.2,
             max_depen_vel=10.0, mu=1.0, ang_damping=0.01, max_ang_vel=100.0, ground_z=0.0, cfm=1e-4, warm=1.0, drive_mode=0)
    p.update(kw)
    return SimParams(**p)


# every symbol the headers declare; checked at load time
SYMBOLS_SIM = [
    "emloco_last_error", "emloco_device_count", "emloco_sim_create", "emloco_sim_destroy", "emloco_sim_set_models",
    "emloco_sim_set_self_collision", "emloco_sim_set_ground_heightfield", "emloco_sim_set_ground_mesh_moves",
    "emloco_sim_prepare", "emloco_sim_get_params", "emloco_sim_set_params", "emloco_sim_tensor",
    "emloco_sim_set_pd_targets", "emloco_sim_set_dof_actuation_force", "emloco_sim_step", "emloco_sim_step_subset", "emloco_sim_set_cost_order", "emloco_sim_set_split", "emloco_sim_sync", "emloco_sim_set_root_state_indexed",
    "emloco_sim_set_dof_state_indexed", "emloco_sim_refresh_bodies", "emloco_sim_num_candidates",
    "emloco_sim_last_step_ms", "emloco_sim_enable_timing", "emloco_sim_timing_stats",
]
SYMBOLS_TASK = [
    "emloco_task_post_physics", "emloco_task_amp_rows", "emloco_task_pd_targets", "emloco_task_last_ms",
    "emloco_task_enable_timing", "emloco_task_reset", "emloco_task_reset_seeded", "emloco_task_compact_done", "emloco_task_compact_done_snapshot", "emloco_task_reset_amp_history",
    "emloco_task_traj_reset", "emloco_task_get_heights", "emloco_task_pd_targets_copy", "emloco_task_compact_done_order", "emloco_task_reset_obs", "emloco_task_reset_obs_pooled", "emloco_task_post_physics_returns",
]

_lib = None


def load():
    """Load the library (no device needed for loading; device entry points fail without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime; it must be the one the process binds first, otherwise torch
    # later fails with "No HIP GPUs are available" (two runtimes with the same soname).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise EmlocoError(f"{LIB_PATH} is missing: run `python -m emloco_amd.build` (hipcc, gfx950). "
                          "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name in SYMBOLS_SIM + SYMBOLS_TASK:
        if not hasattr(lib, name):
            raise EmlocoError(f"libemloco_hip.so does not export {name}")
    lib.emloco_last_error.restype = C.c_char_p
    lib.emloco_sim_last_step_ms.restype = C.c_float
    lib.emloco_task_last_ms.restype = C.c_float
    lib.emloco_sim_create.argtypes = [C.POINTER(SimParams), C.c_int, C.POINTER(C.c_void_p)]
    lib.emloco_sim_destroy.argtypes = [C.c_void_p]
    lib.emloco_sim_set_models.argtypes = [C.c_void_p, C.POINTER(ModelDesc)]
    lib.emloco_sim_prepare.argtypes = [C.c_void_p]
    lib.emloco_sim_get_params.argtypes = [C.c_void_p, C.POINTER(SimParams)]
    lib.emloco_sim_set_params.argtypes = [C.c_void_p, C.POINTER(SimParams)]
    lib.emloco_sim_tensor.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.emloco_sim_set_pd_targets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emloco_sim_set_dof_actuation_force.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emloco_sim_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.emloco_sim_set_cost_order.argtypes = [C.c_void_p, C.c_int]
    lib.emloco_sim_set_split.argtypes = [C.c_void_p, C.c_int]
    lib.emloco_sim_step_subset.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.emloco_sim_sync.argtypes = [C.c_void_p, C.c_void_p]
    lib.emloco_sim_set_root_state_indexed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.emloco_sim_set_dof_state_indexed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.emloco_sim_refresh_bodies.argtypes = [C.c_void_p, C.c_void_p]
    lib.emloco_sim_num_candidates.argtypes = [C.c_void_p]
    lib.emloco_sim_last_step_ms.argtypes = [C.c_void_p]
    lib.emloco_sim_enable_timing.argtypes = [C.c_void_p, C.c_int]
    lib.emloco_sim_timing_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    lib.emloco_task_post_physics.argtypes = [C.POINTER(TaskBufs), C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.emloco_task_post_physics_returns.argtypes = [C.POINTER(TaskBufs), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emloco_task_amp_rows.argtypes = [C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_void_p, C.c_void_p]
    lib.emloco_task_pd_targets.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_void_p]
    lib.emloco_task_enable_timing.argtypes = [C.c_int]
    lib.emloco_task_reset.argtypes = [C.c_void_p, C.POINTER(ResetBufs), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.emloco_task_reset_seeded.argtypes = [C.c_void_p, C.POINTER(ResetBufs), C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.emloco_task_traj_reset.argtypes = [C.POINTER(ResetBufs), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emloco_task_get_heights.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emloco_task_compact_done.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.emloco_task_compact_done_snapshot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emloco_task_reset_amp_history.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.emloco_task_pd_targets_copy.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_void_p, C.c_void_p]
    lib.emloco_task_compact_done_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emloco_task_reset_obs_pooled.argtypes = [C.c_void_p, C.POINTER(ResetBufs), C.POINTER(TaskBufs), C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(ResetPool), C.c_void_p]
    lib.emloco_task_reset_obs.argtypes = [C.c_void_p, C.POINTER(ResetBufs), C.POINTER(TaskBufs), C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().emloco_last_error()
        raise EmlocoError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def require_device():
    lib = load()
    if lib.emloco_device_count() < 1:
        raise EmlocoError("no MI355X / HIP device visible: the emloco hot path has no CPU fallback")
    return lib
