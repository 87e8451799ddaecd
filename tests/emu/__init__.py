"""CPU execution of the HIP kernel sources through tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_EXTRA = os.environ.get("EMLOCO_EMU_EXTRA", "").split()          # e.g. -DEMLOCO_SIM_PAIR=1: the experimental two-envs-per-wave rigid-body kernel
_SO = os.path.join(_HERE, "_build", "libemu" + ("_" + "".join(c for c in "".join(_EXTRA) if c.isalnum()) if _EXTRA else "") + ".so")
_SRCS = ["emu_sim.cpp", "emu_task.cpp", "emu_predictor.cpp", "emu_ppo.cpp", "emu_runtime.cpp", "hip/hip_runtime.h"]
def build():
    """g++ over the kernel sources; every file under emloco_amd/csrc and include/ is a dependency.  Built under a file lock into a
    temporary name and renamed, so that the workers of a parallel test run neither build twice at once nor load a half-written file."""
    import fcntl
    csrc, inc = os.path.join(_ROOT, "emloco_amd", "csrc"), os.path.join(_ROOT, "include")
    deps = [os.path.join(_HERE, s) for s in _SRCS] + [os.path.join(d, f) for d in (csrc, inc) for f in os.listdir(d)]
    deps = [d for d in deps if os.path.isfile(d)]

    def stale():
        return not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps)

    if stale():
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        with open(_SO + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if stale():
                cpps = [os.path.join(_HERE, s) for s in _SRCS if s.endswith(".cpp") and os.path.exists(os.path.join(_HERE, s))]
                tmp = _SO + f".{os.getpid()}.tmp"
                subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-ffp-contract=off", "-Wno-psabi",
                                       "-I", _HERE, "-o", tmp] + _EXTRA + cpps + ["-lpthread"])
                os.replace(tmp, _SO)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


class ModelDesc(C.Structure):
    _fields_ = [("n_env", C.c_int32), ("parent", C.POINTER(C.c_int32)), ("geom_type", C.POINTER(C.c_int32)),
                ("joint_off", C.POINTER(C.c_float)), ("mass", C.POINTER(C.c_float)), ("com", C.POINTER(C.c_float)),
                ("inertia", C.POINTER(C.c_float)), ("geom_a", C.POINTER(C.c_float)), ("geom_b", C.POINTER(C.c_float)),
                ("geom_r", C.POINTER(C.c_float)), ("kp", C.POINTER(C.c_float)), ("kd", C.POINTER(C.c_float)),
                ("armature", C.POINTER(C.c_float)), ("effort", C.POINTER(C.c_float))]


def model_desc(arr):
    return ModelDesc(arr["mass"].shape[0], _p(arr["parent"], C.c_int32), _p(arr["geom_type"], C.c_int32),
                     _p(arr["joint_off"]), _p(arr["mass"]), _p(arr["com"]), _p(arr["inertia"]), _p(arr["geom_a"]),
                     _p(arr["geom_b"]), _p(arr["geom_r"]), _p(arr["kp"]), _p(arr["kd"]), _p(arr["armature"]),
                     _p(arr["effort"]))


class SelfCollisionDesc(C.Structure):
    _fields_ = [("n_pairs", C.c_int32), ("pairs", C.POINTER(C.c_uint8)), ("cap_a", C.POINTER(C.c_float)),
                ("cap_b", C.POINTER(C.c_float)), ("cap_r", C.POINTER(C.c_float)), ("k", C.c_float), ("c", C.c_float),
                ("max_pen", C.c_float), ("mu", C.c_float), ("n_seg", C.c_int32), ("seg_body", C.POINTER(C.c_uint8))]


def sim_step(osim, n_calls=1, expect_error=None):
    """Advance an oracle.Sim-shaped state holder with the emulated HIP kernel (same arrays, in place)."""
    desc = model_desc(osim.arr)
    sc = getattr(osim, "sc", None)
    scd = None
    if sc is not None:
        scd = SelfCollisionDesc(int(sc["pairs"].shape[0]), _p(sc["pairs"], C.c_uint8), _p(sc["cap_a"]), _p(sc["cap_b"]),
                                _p(sc["cap_r"]), float(sc["k"]), float(sc["c"]), float(sc["max_pen"]), float(sc.get("mu", 1.0)),
                                int(sc["seg_body"].shape[0]) if sc.get("seg_body") is not None else 0,
                                _p(sc["seg_body"], C.c_uint8) if sc.get("seg_body") is not None else None)
    lib().emu_sim_set_self_collision(C.byref(scd) if scd is not None else None)
    hf = getattr(osim, "hf", None)
    fn = lib().emu_sim_set_heightfield
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
    if hf is not None:
        m = osim.model
        mv = getattr(osim, "hf_mv", None)
        fn(hf.ctypes.data, hf.shape[0], hf.shape[1], m.hf_hs, m.hf_vs, m.hf_ox, m.hf_oy, mv.ctypes.data if mv is not None else None)
    else:
        fn(None, 0, 0, 0.0, 0.0, 0.0, 0.0, None)
    rc = lib().emu_sim_step(C.byref(osim.params), C.byref(desc), _p(osim.root_state), _p(osim.dof_state),
                            _p(osim.pd_target), _p(osim.rb_state), _p(osim.contact_force), _p(osim.dof_force),
                            _p(osim.lambda_ws), C.c_int(n_calls))
    if expect_error is None:
        assert rc == 0
    return rc


def sim_fk(osim):
    desc = model_desc(osim.arr)
    assert lib().emu_sim_fk(C.byref(desc), _p(osim.root_state), _p(osim.dof_state), _p(osim.rb_state)) == 0


# ------------------------------------------------------------------ task kernels
def _vp(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


class TaskHost:
    """Host-memory buffers laid out like the task's device buffers, for running the kernel under emulation."""
    L2R = [0, 5, 6, 7, 8, 1, 2, 3, 4, 9, 10, 11, 12, 13, 19, 20, 21, 22, 23, 14, 15, 16, 17, 18]

    def __init__(self, E, heightfield, dt=1.0 / 30.0, episode_len=168):
        from emloco_amd import _lib as L
        self.L = L
        self.E = E
        f32 = lambda *s: np.zeros(s, np.float32)
        self.rb_state = f32(E, 24, 13); self.rb_state[:, :, 6] = 1
        self.dof_state = f32(E, 69, 2)
        self.dof_force = f32(E, 69)
        self.contact_force = f32(E, 24, 3)
        self.betas = f32(E, 17)
        self.traj_verts = f32(E, 101, 3)
        self.heightfield = np.ascontiguousarray(heightfield, dtype=np.int16)
        self.l2r = np.asarray(self.L2R, np.int32)
        mask = np.zeros(24, np.uint8); mask[[7, 3, 8, 4]] = 1
        self.contact_mask = mask
        self.key_bodies = np.asarray([7, 3, 22, 17], np.int32)
        self.dof_subset = np.concatenate([np.arange(3 * j, 3 * j + 3) for j in range(23) if j not in (3, 7, 17, 22)]).astype(np.int32)
        self.progress = np.zeros(E, np.int64)
        self.reset = np.ones(E, np.int64)
        self.terminate = np.ones(E, np.int64)
        self.obs = f32(E, L.OBS); self.flip_obs = f32(E, L.OBS)
        self.rew = f32(E); self.reward_raw = f32(E, 2)
        self.amp = f32(E, L.AMP_STEPS, L.AMP_ROW)
        vdt = episode_len * dt / 100.0
        self.dt, self.traj_dur = dt, 101 * vdt
        self.episode_len = episode_len

    def bufs(self):
        L = self.L
        return L.TaskBufs(self.E, self.heightfield.shape[0], self.heightfield.shape[1], 13, len(self.dof_subset),
                          self.dt, self.traj_dur, 0.4, 0.1, 0.005, 0.0005, 4.0, float(self.episode_len),
                          _vp(self.rb_state), _vp(self.dof_state), _vp(self.dof_force), _vp(self.contact_force),
                          _vp(self.betas), _vp(self.traj_verts), _vp(self.heightfield), _vp(self.l2r),
                          _vp(self.contact_mask), _vp(self.key_bodies), _vp(self.dof_subset),
                          _vp(self.progress), _vp(self.reset), _vp(self.terminate), _vp(self.obs), _vp(self.flip_obs),
                          _vp(self.rew), _vp(self.reward_raw), _vp(self.amp))

    def post_physics(self, mode, env_ids=None):
        b = self.bufs()
        fn = lib().emu_task_post_physics
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        if env_ids is None:
            rc = fn(C.byref(b), mode, None, 0)
        else:
            ids = np.ascontiguousarray(env_ids, np.int32)
            rc = fn(C.byref(b), mode, _vp(ids), len(ids))
        assert rc == 0


def task_pd_targets(actions, offset, scale, zero_mask):
    actions = np.ascontiguousarray(actions, np.float32)
    out = np.zeros_like(actions)
    fn = lib().emu_task_pd_targets
    fn.argtypes = [C.c_int] + [C.c_void_p] * 5
    fn(actions.shape[0], _vp(actions), _vp(np.ascontiguousarray(offset, np.float32)),
       _vp(np.ascontiguousarray(scale, np.float32)), _vp(np.ascontiguousarray(zero_mask, np.uint8)), _vp(out))
    return out
