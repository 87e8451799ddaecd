"""NativeSim: Python handle on the C-ABI simulator (include/emloco_sim.h).

The sim owns its state buffers in HBM; the torch tensors exposed here are NON-OWNING ALIASES of that
memory (what `gymtorch.wrap_tensor` gives the reference's task code, isaacgym/python/isaacgym/gymtorch.py:60-70):
writing into them edits the simulator state, and a step overwrites them.  PyTorch is plumbing only --
device memory views and the current HIP stream -- the arithmetic is in libemloco_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .model import HumanoidModel, pack_models

_TORCH_TYPESTR = {torch.float32: "<f4", torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1", torch.int16: "<i2"}


class _DevView:
    """Minimal __cuda_array_interface__ carrier (works for HIP memory under PyTorch-ROCm)."""

    def __init__(self, ptr, shape, dtype, owner):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": _TORCH_TYPESTR[dtype],
                                         "data": (int(ptr), False), "version": 2, "strides": None}
        self._owner = owner


def wrap_device_pointer(ptr, shape, dtype, device, owner=None):
    """Zero-copy torch view of raw device memory."""
    return torch.as_tensor(_DevView(ptr, shape, dtype, owner), device=device)


def current_stream_handle(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dptr(t):
    """Raw device address of a contiguous torch tensor (gymtorch.unwrap_tensor requires contiguity, gymtorch.py:97-107)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise L.EmlocoError("tensor handed to the C ABI must be contiguous")
    return C.c_void_p(t.data_ptr())


class NativeSim:
    def __init__(self, models, params=None, device_index=0, self_collision=None, heightfield=None):
        """`self_collision`: None / False = off; True = limb-limb contacts with the defaults of
        `model.pack_self_collision`; a dict from `pack_self_collision(models, ...)` to choose the parameters.
        `heightfield`: None = the plane z = params.ground_z; dict(samples int16 [nx][ny], horizontal_scale, vertical_scale,
        origin_x=0, origin_y=0[, move_x, move_y int8 [nx][ny]]) = height-field ground (emloco_sim_set_ground_heightfield); with the
        vertex moves of the slope-corrected mesh (terrain_utils.mesh_vertex_moves) its vertical faces collide too."""
        lib = L.require_device()
        self.lib = lib
        self.device_index = int(device_index)
        self.device = torch.device("cuda", self.device_index)
        self.params = params or L.default_sim_params()
        if isinstance(models, dict):
            packed = models
        else:
            if isinstance(models, HumanoidModel):
                models = [models]
            packed = pack_models(models)
        self._packed = {k: np.ascontiguousarray(v) for k, v in packed.items()}
        self.num_envs = int(self._packed["mass"].shape[0])
        self._h = C.c_void_p()
        L.check(lib.emloco_sim_create(C.byref(self.params), self.device_index, C.byref(self._h)), "emloco_sim_create")
        p = self._packed
        f = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        desc = L.ModelDesc(self.num_envs, i(p["parent"]), i(p["geom_type"]), f(p["joint_off"]), f(p["mass"]),
                           f(p["com"]), f(p["inertia"]), f(p["geom_a"]), f(p["geom_b"]), f(p["geom_r"]),
                           f(p["kp"]), f(p["kd"]), f(p["armature"]), f(p["effort"]))
        L.check(lib.emloco_sim_set_models(self._h, C.byref(desc)), "emloco_sim_set_models")
        if self_collision:
            if self_collision is True:
                if isinstance(models, dict):
                    raise L.EmlocoError("self_collision=True needs HumanoidModel objects (or pass pack_self_collision(...))")
                from .model import pack_self_collision
                self_collision = pack_self_collision(models)
            sc = self._sc = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in self_collision.items()}
            scd = L.SelfCollisionDesc(int(sc["pairs"].shape[0]), sc["pairs"].ctypes.data_as(C.POINTER(C.c_uint8)), f(sc["cap_a"]),
                                      f(sc["cap_b"]), f(sc["cap_r"]), float(sc["k"]), float(sc["c"]), float(sc["max_pen"]), float(sc.get("mu", 1.0)),
                                      int(sc["seg_body"].shape[0]) if sc.get("seg_body") is not None else 0,
                                      sc["seg_body"].ctypes.data_as(C.POINTER(C.c_uint8)) if sc.get("seg_body") is not None else None)
            lib.emloco_sim_set_self_collision.argtypes = [C.c_void_p, C.POINTER(L.SelfCollisionDesc)]
            L.check(lib.emloco_sim_set_self_collision(self._h, C.byref(scd)), "emloco_sim_set_self_collision")
        if heightfield is not None:
            hf = np.ascontiguousarray(heightfield["samples"], dtype=np.int16)
            if hf.ndim != 2:
                raise L.EmlocoError("heightfield samples must be a 2-D int16 array [nx][ny]")
            lib.emloco_sim_set_ground_heightfield.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                                              C.c_float, C.c_float]
            L.check(lib.emloco_sim_set_ground_heightfield(
                self._h, hf.ctypes.data, hf.shape[0], hf.shape[1], float(heightfield["horizontal_scale"]),
                float(heightfield["vertical_scale"]), float(heightfield.get("origin_x", 0.0)),
                float(heightfield.get("origin_y", 0.0))), "emloco_sim_set_ground_heightfield")
            if heightfield.get("move_x") is not None:          # the slope-corrected mesh: its vertical faces collide
                mx = np.ascontiguousarray(heightfield["move_x"], dtype=np.int8)
                my = np.ascontiguousarray(heightfield["move_y"], dtype=np.int8)
                if mx.shape != hf.shape or my.shape != hf.shape:
                    raise L.EmlocoError("heightfield move_x / move_y must have the samples' shape")
                lib.emloco_sim_set_ground_mesh_moves.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
                L.check(lib.emloco_sim_set_ground_mesh_moves(self._h, mx.ctypes.data, my.ctypes.data), "emloco_sim_set_ground_mesh_moves")
        L.check(lib.emloco_sim_prepare(self._h), "emloco_sim_prepare")
        self.root_state = self._tensor(L.T_ROOT_STATE)
        self.dof_state = self._tensor(L.T_DOF_STATE)
        self.rigid_body_state = self._tensor(L.T_RIGID_BODY)
        self.contact_force = self._tensor(L.T_CONTACT_FORCE)
        self.dof_force = self._tensor(L.T_DOF_FORCE).view(-1)
        self.pd_target = self._tensor(L.T_PD_TARGET)
        self.warm_start = self._tensor(L.T_WARM_START)

    def _tensor(self, kind):
        ptr = C.c_void_p()
        shape = (C.c_int64 * 2)()
        L.check(self.lib.emloco_sim_tensor(self._h, kind, C.byref(ptr), shape), "emloco_sim_tensor")
        return wrap_device_pointer(ptr.value, (shape[0], shape[1]), torch.float32, self.device, owner=self)

    def _stream(self):
        return current_stream_handle(self.device)

    # ------------------------------------------------------------------ gym-level operations
    def set_pd_targets(self, targets):
        L.check(self.lib.emloco_sim_set_pd_targets(self._h, dptr(targets), self._stream()), "emloco_sim_set_pd_targets")

    def set_dof_actuation_force(self, forces):
        """Joint torques [n_env][69] for effort drives (emloco_sim_set_dof_actuation_force); set_pd_targets switches back."""
        if forces.dtype != torch.float32 or forces.numel() != self.num_envs * 69:
            raise L.EmlocoError("actuation forces must be a float32 [n_env][69] tensor")
        L.check(self.lib.emloco_sim_set_dof_actuation_force(self._h, dptr(forces), self._stream()), "emloco_sim_set_dof_actuation_force")

    def step(self, n_calls=1):
        L.check(self.lib.emloco_sim_step(self._h, int(n_calls), self._stream()), "emloco_sim_step")

    def set_cost_order(self, on=True):
        """Longest-first dispatch of the step launch from the per-env durations of the previous one (emloco_sim_set_cost_order)."""
        L.check(self.lib.emloco_sim_set_cost_order(self._h, int(bool(on))), "emloco_sim_set_cost_order")

    def set_split(self, n_parts=2):
        """The substeps of a step as `n_parts` dependent workgroups per env in one launch (emloco_sim_set_split)."""
        L.check(self.lib.emloco_sim_set_split(self._h, int(n_parts)), "emloco_sim_set_split")

    def step_subset(self, n_calls=1, skip=None, ids=None):
        """The step for a subset of the envs on the current stream: `skip` (int64 per env) leaves the flagged envs alone,
        `ids` (int32 device-compacted list, -1 padded) steps exactly the listed ones."""
        if skip is not None:
            assert skip.dtype == torch.int64 and skip.numel() == self.num_envs
        if ids is not None:
            assert ids.dtype == torch.int32
        L.check(self.lib.emloco_sim_step_subset(self._h, int(n_calls), dptr(skip), dptr(ids), 0 if ids is None else int(ids.numel()),
                                                self._stream()), "emloco_sim_step_subset")

    def sync(self):
        L.check(self.lib.emloco_sim_sync(self._h, self._stream()), "emloco_sim_sync")

    def set_root_state_indexed(self, full, env_ids_i32):
        L.check(self.lib.emloco_sim_set_root_state_indexed(self._h, dptr(full), dptr(env_ids_i32), int(env_ids_i32.numel()),
                                                          self._stream()), "emloco_sim_set_root_state_indexed")

    def set_dof_state_indexed(self, full, env_ids_i32):
        L.check(self.lib.emloco_sim_set_dof_state_indexed(self._h, dptr(full), dptr(env_ids_i32), int(env_ids_i32.numel()),
                                                         self._stream()), "emloco_sim_set_dof_state_indexed")

    def refresh_bodies(self):
        L.check(self.lib.emloco_sim_refresh_bodies(self._h, self._stream()), "emloco_sim_refresh_bodies")

    def get_params(self):
        out = L.SimParams()
        L.check(self.lib.emloco_sim_get_params(self._h, C.byref(out)), "emloco_sim_get_params")
        return out

    def set_params(self, p):
        L.check(self.lib.emloco_sim_set_params(self._h, C.byref(p)), "emloco_sim_set_params")
        self.params = p

    def enable_timing(self, on=True, every=1):
        """HIP-event timing of the step launches; `every` = N times every N-th launch only."""
        L.check(self.lib.emloco_sim_enable_timing(self._h, int(every) if on else 0), "emloco_sim_enable_timing")

    def timing_stats(self):
        """(number of step launches, their summed HIP-event duration in ms) since timing was enabled / last read."""
        n, ms = C.c_int(), C.c_float()
        L.check(self.lib.emloco_sim_timing_stats(self._h, C.byref(n), C.byref(ms)), "emloco_sim_timing_stats")
        return n.value, ms.value

    def last_step_ms(self):
        return float(self.lib.emloco_sim_last_step_ms(self._h))

    @property
    def num_candidates(self):
        return int(self.lib.emloco_sim_num_candidates(self._h))

    def close(self):
        if self._h:
            self.lib.emloco_sim_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
