"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package
(see oracle/README.md).  The product (emloco_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

NB, NDOF, MAXCAND, MAXC = 24, 69, 96, 20


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("oracle_task.c", "oracle_sim.c", "oracle_math.h", "oracle_sim.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if os.path.exists(os.path.join(_HERE, "oracle_sim.c")):
            build()                              # no-op unless a source is newer than the library
        _lib = C.CDLL(_SO)
    return _lib


def set_threads(n=0):
    """OpenMP threads of the per-env loops (0 = all cores); returns the count in effect.  Results do not depend on it."""
    return int(lib().orc_set_threads(C.c_int(int(n))))


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class SimParams(C.Structure):
    _fields_ = [("n_sub", C.c_int32), ("n_iter", C.c_int32), ("h", C.c_float), ("gravity_z", C.c_float),
                ("contact_offset", C.c_float), ("erp", C.c_float), ("max_depen_vel", C.c_float),
                ("mu", C.c_float), ("ang_damping", C.c_float), ("max_ang_vel", C.c_float),
                ("ground_z", C.c_float), ("cfm", C.c_float), ("warm", C.c_float), ("drive_mode", C.c_int32)]


def default_params(**kw):
    p = dict(n_sub=4, n_iter=4, h=1.0 / 120.0, gravity_z=-9.81, contact_offset=0.02, erp=0.2,
             max_depen_vel=10.0, mu=1.0, ang_damping=0.01, max_ang_vel=100.0, ground_z=0.0, cfm=1e-4, warm=1.0, drive_mode=0)
    p.update(kw)
    return SimParams(**p)


class Model(C.Structure):
    _fields_ = [("parent", C.POINTER(C.c_int32)), ("geom_type", C.POINTER(C.c_int32)),
                ("joint_off", C.POINTER(C.c_float)), ("mass", C.POINTER(C.c_float)),
                ("com", C.POINTER(C.c_float)), ("inertia", C.POINTER(C.c_float)),
                ("geom_a", C.POINTER(C.c_float)), ("geom_b", C.POINTER(C.c_float)),
                ("geom_r", C.POINTER(C.c_float)), ("kp", C.POINTER(C.c_float)), ("kd", C.POINTER(C.c_float)),
                ("armature", C.POINTER(C.c_float)), ("effort", C.POINTER(C.c_float)),
                ("sc_n", C.c_int32), ("sc_pairs", C.POINTER(C.c_uint8)), ("sc_cap_a", C.POINTER(C.c_float)),
                ("sc_cap_b", C.POINTER(C.c_float)), ("sc_cap_r", C.POINTER(C.c_float)), ("sc_k", C.c_float), ("sc_c", C.c_float),
                ("sc_max_pen", C.c_float), ("sc_mu", C.c_float), ("sc_nseg", C.c_int32), ("sc_segbody", C.POINTER(C.c_uint8)),
                ("hf", C.POINTER(C.c_int16)), ("hf_nx", C.c_int32), ("hf_ny", C.c_int32), ("hf_hs", C.c_float), ("hf_vs", C.c_float),
                ("hf_ox", C.c_float), ("hf_oy", C.c_float), ("hf_mv", C.POINTER(C.c_uint8))]


def pack_mesh_moves(move_x, move_y):
    """Vertex moves of the slope-corrected terrain mesh (terrain_utils.py:313-325), each -1 / 0 / +1 cells, as the byte per
    sample oracle_sim.h: hf_mv describes: bits 0-1 move_x + 1, bits 2-3 move_y + 1, bit 4 = some vertex of the 4 x 4 block
    (i - 1 .. i + 2, j - 1 .. j + 2) around cell (i, j) moved."""
    mx, my = np.asarray(move_x).astype(np.int64), np.asarray(move_y).astype(np.int64)
    assert mx.shape == my.shape and mx.ndim == 2 and np.abs(mx).max() <= 1 and np.abs(my).max() <= 1
    moved = ((mx != 0) | (my != 0))
    nx, ny = mx.shape
    pad = np.zeros((nx + 3, ny + 3), bool)
    pad[1:nx + 1, 1:ny + 1] = moved
    flag = np.zeros((nx, ny), bool)
    for a in range(4):
        for b in range(4):
            flag |= pad[a:a + nx, b:b + ny]
    return np.ascontiguousarray((mx + 1) | ((my + 1) << 2) | (flag.astype(np.int64) << 4), dtype=np.uint8)


class Sim:
    """Sequential CPU simulator over the packed model arrays of emloco_amd.model.pack_models()."""

    def __init__(self, packed, params=None, self_collision=None, heightfield=None):
        """`self_collision`: dict from emloco_amd.model.pack_self_collision (pairs, cap_a, cap_b, cap_r, k, c, max_pen) or None.
        `heightfield`: dict(samples int16 [nx][ny], horizontal_scale, vertical_scale, origin_x=0, origin_y=0) or None (plane)."""
        self.arr = {k: np.ascontiguousarray(v) for k, v in packed.items()}
        self.sc = None if not self_collision else {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v)
                                                   for k, v in self_collision.items()}
        self.E = self.arr["mass"].shape[0]
        self.params = params or default_params()
        a = self.arr
        self.model = Model(_p(a["parent"], C.c_int32), _p(a["geom_type"], C.c_int32), _p(a["joint_off"]),
                           _p(a["mass"]), _p(a["com"]), _p(a["inertia"]), _p(a["geom_a"]), _p(a["geom_b"]),
                           _p(a["geom_r"]), _p(a["kp"]), _p(a["kd"]), _p(a["armature"]), _p(a["effort"]))
        if self.sc is not None:
            c = self.sc
            self.model.sc_n = int(c["pairs"].shape[0])
            self.model.sc_pairs = _p(c["pairs"], C.c_uint8)
            self.model.sc_cap_a, self.model.sc_cap_b, self.model.sc_cap_r = _p(c["cap_a"]), _p(c["cap_b"]), _p(c["cap_r"])
            self.model.sc_k, self.model.sc_c, self.model.sc_max_pen = float(c["k"]), float(c["c"]), float(c["max_pen"])
            self.model.sc_mu = float(c.get("mu", 1.0))
            if c.get("seg_body") is not None:
                self._seg_body = np.ascontiguousarray(c["seg_body"], np.uint8)
                self.model.sc_nseg = int(self._seg_body.shape[0])
                self.model.sc_segbody = _p(self._seg_body, C.c_uint8)
        self.hf = None
        if heightfield is not None:
            self.hf = np.ascontiguousarray(heightfield["samples"], dtype=np.int16)
            self.model.hf = _p(self.hf, C.c_int16)
            self.model.hf_nx, self.model.hf_ny = self.hf.shape
            self.model.hf_hs, self.model.hf_vs = float(heightfield["horizontal_scale"]), float(heightfield["vertical_scale"])
            self.model.hf_ox, self.model.hf_oy = float(heightfield.get("origin_x", 0.0)), float(heightfield.get("origin_y", 0.0))
            if heightfield.get("move_x") is not None:
                self.hf_mv = pack_mesh_moves(heightfield["move_x"], heightfield["move_y"])
                assert self.hf_mv.shape == self.hf.shape
                self.model.hf_mv = _p(self.hf_mv, C.c_uint8)
        E = self.E
        self.root_state = np.zeros((E, 13), np.float32)
        self.root_state[:, 6] = 1.0
        self.dof_state = np.zeros((E, NDOF, 2), np.float32)
        self.pd_target = np.zeros((E, NDOF), np.float32)
        self.rb_state = np.zeros((E, NB, 13), np.float32)
        self.contact_force = np.zeros((E, NB, 3), np.float32)
        self.dof_force = np.zeros((E, NDOF), np.float32)
        self.lambda_ws = np.zeros((E, MAXCAND, 3), np.float32)

    def fk(self):
        lib().orc_sim_fk(C.c_int(self.E), C.byref(self.model), _p(self.root_state), _p(self.dof_state), _p(self.rb_state))
        return self.rb_state

    def step(self, n_calls=1):
        for _ in range(n_calls):
            lib().orc_sim_step(C.c_int(self.E), C.byref(self.params), C.byref(self.model), _p(self.root_state),
                               _p(self.dof_state), _p(self.pd_target), _p(self.rb_state), _p(self.contact_force),
                               _p(self.dof_force), _p(self.lambda_ws))

    def free_accel(self, env=0):
        out = np.zeros(75, np.float32)
        lib().orc_sim_free_accel(C.byref(self.params), C.byref(self.model), C.c_int(env), _p(self.root_state),
                                 _p(self.dof_state), _p(self.pd_target), _p(out))
        return out

    def self_contacts(self, env=0):
        """Limb-limb contacts at the current state: rows [bi, bj, point (relative to the root origin) 3, normal 3, F_normal,
        total force on bi 3] (test hook)."""
        info = np.zeros((32, 12), np.float32)
        lib().orc_sim_self_contacts.restype = C.c_int
        n = lib().orc_sim_self_contacts(C.byref(self.params), C.byref(self.model), C.c_int(env), _p(self.root_state),
                                        _p(self.dof_state), _p(info))
        return info[:n]

    def dense_dynamics(self, env=0):
        M = np.zeros((75, 75), np.float64)
        rhs = np.zeros(75, np.float64)
        lib().orc_sim_dense_dynamics(C.byref(self.params), C.byref(self.model), C.c_int(env), _p(self.root_state),
                                     _p(self.dof_state), _p(self.pd_target), _p(M, C.c_double), _p(rhs, C.c_double))
        return M, rhs


# ------------------------------------------------------------------ task functions (oracle_task.c)
def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def self_obs(pos, rot, vel, ang, betas, flip=False):
    pos, rot, vel, ang, betas = map(_f32, (pos, rot, vel, ang, betas))
    E = pos.shape[0]
    out = np.zeros((E, 368), np.float32)
    fn = lib().orc_flip_self_obs if flip else lib().orc_self_obs
    fn(C.c_int(E), _p(pos), _p(rot), _p(vel), _p(ang), _p(betas), _p(out))
    return out


def traj_calc_pos(verts, progress, dt, traj_dur):
    verts, progress = _f32(verts), _i64(progress)
    E = verts.shape[0]
    out = np.zeros((E, 3), np.float32)
    lib().orc_traj_calc_pos(C.c_int(E), _p(verts), _p(progress, C.c_int64), C.c_float(dt), C.c_float(traj_dur), _p(out))
    return out


def traj_samples(verts, progress, dt, traj_dur, sample_dt=0.4):
    verts, progress = _f32(verts), _i64(progress)
    E = verts.shape[0]
    out = np.zeros((E, 15, 3), np.float32)
    lib().orc_traj_samples(C.c_int(E), _p(verts), _p(progress, C.c_int64), C.c_float(dt), C.c_float(traj_dur),
                           C.c_float(sample_dt), _p(out))
    return out


def location_obs(root_states, samples):
    root_states, samples = _f32(root_states), _f32(samples)
    E = root_states.shape[0]
    out = np.zeros((E, 30), np.float32)
    lib().orc_location_obs(C.c_int(E), _p(root_states), _p(samples), _p(out))
    return out


def get_heights(pose7, hf, hscale=0.1, vscale=0.005, heading_q=None, return_index=False):
    """`heading_q` (E,4): use this heading quaternion instead of computing it; `return_index`: also the int64 map indices."""
    pose7 = _f32(pose7)
    hf = np.ascontiguousarray(hf, dtype=np.int16)
    E = pose7.shape[0]
    out = np.empty((E, 1024), np.float32)
    px = py = None
    if return_index:
        px, py = np.empty((E, 1024), np.int64), np.empty((E, 1024), np.int64)
    hq = None if heading_q is None else _f32(heading_q)
    lib().orc_get_heights_ex(C.c_int(E), _p(pose7), None if hq is None else _p(hq), _p(hf, C.c_int16), C.c_int(hf.shape[0]),
                             C.c_int(hf.shape[1]), C.c_float(hscale), C.c_float(vscale), _p(out),
                             None if px is None else _p(px, C.c_int64), None if py is None else _p(py, C.c_int64))
    return (out, px, py) if return_index else out


def get_center_heights(root_states, hf, hscale=0.1, vscale=0.005, return_index=False):
    root_states = _f32(root_states)
    hf = np.ascontiguousarray(hf, dtype=np.int16)
    E = root_states.shape[0]
    out = np.empty((E, 9), np.float32)
    px, py = np.empty((E, 9), np.int64), np.empty((E, 9), np.int64)
    lib().orc_get_center_heights_ex(C.c_int(E), _p(root_states), _p(hf, C.c_int16), C.c_int(hf.shape[0]), C.c_int(hf.shape[1]),
                                    C.c_float(hscale), C.c_float(vscale), _p(out), _p(px, C.c_int64), _p(py, C.c_int64))
    return (out, px, py) if return_index else out


def height_obs(center9, heights):
    center9, heights = _f32(center9), _f32(heights)
    E = center9.shape[0]
    out = np.zeros((E, 1024), np.float32)
    lib().orc_height_obs(C.c_int(E), _p(center9), _p(heights), _p(out))
    return out


def flip_task_obs(task_obs):
    task_obs = _f32(task_obs)
    out = np.zeros_like(task_obs)
    lib().orc_flip_task_obs(C.c_int(task_obs.shape[0]), _p(task_obs), _p(out))
    return out


def reward(root_pos, tar_pos, dof_force, dof_vel, power_coef=0.0005):
    root_pos, tar_pos, dof_force, dof_vel = map(_f32, (root_pos, tar_pos, dof_force, dof_vel))
    E = root_pos.shape[0]
    rew = np.zeros(E, np.float32)
    raw = np.zeros((E, 2), np.float32)
    lib().orc_reward(C.c_int(E), _p(root_pos), _p(tar_pos), _p(dof_force), _p(dof_vel), C.c_float(power_coef),
                     _p(rew), _p(raw))
    return rew, raw


def reset(progress, contact, body_pos, tar_pos, contact_body_ids=(7, 3, 8, 4), max_episode_length=168.0, fail_dist=4.0):
    progress = _i64(progress)
    contact, body_pos, tar_pos = map(_f32, (contact, body_pos, tar_pos))
    ids = np.ascontiguousarray(contact_body_ids, dtype=np.int32)
    E = progress.shape[0]
    rs = np.zeros(E, np.int64)
    tm = np.zeros(E, np.int64)
    lib().orc_reset(C.c_int(E), _p(progress, C.c_int64), _p(contact), _p(ids, C.c_int32), C.c_int(len(ids)),
                    _p(body_pos), _p(tar_pos), C.c_float(max_episode_length), C.c_float(fail_dist),
                    _p(rs, C.c_int64), _p(tm, C.c_int64))
    return rs, tm


def amp_obs(root_pos, root_rot, root_vel, root_ang, dof_pos, dof_vel, key_pos, betas, dof_subset):
    a = list(map(_f32, (root_pos, root_rot, root_vel, root_ang, dof_pos, dof_vel, key_pos, betas)))
    sub = np.ascontiguousarray(dof_subset, dtype=np.int32)
    E = a[0].shape[0]
    out = np.zeros((E, 206), np.float32)
    lib().orc_amp_obs(C.c_int(E), *[_p(x) for x in a], _p(sub, C.c_int32), C.c_int(len(sub)), _p(out))
    return out


def pd_targets(actions, offset, scale, zero_mask):
    actions, offset, scale = map(_f32, (actions, offset, scale))
    zm = np.ascontiguousarray(zero_mask, dtype=np.uint8)
    out = np.zeros_like(actions)
    lib().orc_pd_targets(C.c_int(actions.shape[0]), _p(actions), _p(offset), _p(scale), _p(zm, C.c_uint8), _p(out))
    return out


def vec(name, *arrs, out_width):
    arrs = [_f32(a) for a in arrs]
    n = arrs[0].shape[0]
    out = np.zeros((n, out_width) if out_width > 1 else (n,), np.float32)
    getattr(lib(), "orc_vec_" + name)(C.c_int(n), *[_p(a) for a in arrs], _p(out))
    return out
