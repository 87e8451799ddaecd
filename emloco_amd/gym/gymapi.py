"""gymapi: the subset of Isaac Gym's Python API on the rollout hot path, over the C ABI.

Reference shim: /root/reference/isaacgym/python/isaacgym/gymapi.py:41-56 (loads the absent binary);
API contract: /root/reference/isaacgym/docs/api/python/gym_py.html.  Call sites are cited per method.
Return conventions follow the reference: bool / None / handles, no exceptions on the hot path
(`create_sim` returns None on failure, base_task.py:238-241).
"""
import os

import numpy as np
import torch

from .. import _lib as L
from ..model import HumanoidModel, load_mjcf

SIM_PHYSX, SIM_FLEX = 0, 1
DOF_MODE_NONE, DOF_MODE_POS, DOF_MODE_VEL, DOF_MODE_EFFORT = 0, 1, 2, 3
UP_AXIS_Y, UP_AXIS_Z = 0, 1
ENV_SPACE, LOCAL_SPACE, GLOBAL_SPACE = 0, 1, 2
MESH_VISUAL, MESH_COLLISION, MESH_VISUAL_AND_COLLISION = 0, 1, 2
DTYPE_FLOAT32, DTYPE_UINT32, DTYPE_UINT64, DTYPE_UINT8, DTYPE_INT16 = 0, 1, 2, 3, 4


class Vec3:
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)

    def __iter__(self):
        return iter((self.x, self.y, self.z))


class Quat:
    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.x, self.y, self.z, self.w = float(x), float(y), float(z), float(w)


class Transform:
    def __init__(self, p=None, r=None):
        self.p = p or Vec3()
        self.r = r or Quat()


class PhysXParams:
    def __init__(self):
        self.solver_type = 1
        self.num_position_iterations = 4
        self.num_velocity_iterations = 0
        self.num_threads = 4
        self.use_gpu = True
        self.num_subscenes = 0
        self.max_gpu_contact_pairs = 1024 * 1024
        self.contact_offset = 0.02
        self.rest_offset = 0.0
        self.bounce_threshold_velocity = 0.2
        self.max_depenetration_velocity = 10.0
        self.default_buffer_size_multiplier = 2.0


class FlexParams:
    def __init__(self):
        self.shape_collision_margin = 0.01
        self.num_outer_iterations = 4
        self.num_inner_iterations = 10
        self.warm_start = 0.25


class SimParams:
    """config.py:141-174 fills this; pacer.yaml:93-104 overrides the physx block."""

    def __init__(self):
        self.dt = 1.0 / 60.0
        self.substeps = 2
        self.up_axis = UP_AXIS_Z
        self.gravity = Vec3(0.0, 0.0, -9.81)
        self.use_gpu_pipeline = True
        self.num_client_threads = 0
        self.physx = PhysXParams()
        self.flex = FlexParams()


class PlaneParams:
    def __init__(self):
        self.normal = Vec3(0.0, 0.0, 1.0)
        self.distance = 0.0
        self.static_friction = 1.0
        self.dynamic_friction = 1.0
        self.restitution = 0.0


class TriangleMeshParams:
    def __init__(self):
        self.nb_vertices = 0
        self.nb_triangles = 0
        self.transform = Transform()
        self.static_friction = 1.0
        self.dynamic_friction = 1.0
        self.restitution = 0.0


class AssetOptions:
    def __init__(self):
        self.angular_damping = 0.5
        self.linear_damping = 0.0
        self.max_angular_velocity = 64.0
        self.density = 1000.0
        self.default_dof_drive_mode = DOF_MODE_NONE
        self.fix_base_link = False


class RigidBodyProperties:
    def __init__(self, mass):
        self.mass = float(mass)


class RigidShapeProperties:
    def __init__(self):
        self.filter = 0
        self.friction = 1.0


class ActuatorProperties:
    def __init__(self, effort):
        self.motor_effort = float(effort)


DOF_PROP_DTYPE = np.dtype([("hasLimits", "?"), ("lower", "f4"), ("upper", "f4"), ("driveMode", "i4"),
                           ("velocity", "f4"), ("effort", "f4"), ("stiffness", "f4"), ("damping", "f4"),
                           ("friction", "f4"), ("armature", "f4")])


class Tensor:
    """Descriptor returned by acquire_*_tensor (what gymtorch.wrap_tensor consumes, gymtorch.py:60-70)."""

    def __init__(self, torch_tensor):
        self._t = torch_tensor
        self.shape = tuple(torch_tensor.shape)
        self.dtype = DTYPE_FLOAT32
        self.device = torch_tensor.device.index if torch_tensor.is_cuda else -1
        self.data_address = torch_tensor.data_ptr()


class Asset:
    def __init__(self, model: HumanoidModel, options: AssetOptions):
        self.model = model
        self.options = options
        self.force_sensors = []


class Env:
    def __init__(self, index):
        self.index = index
        self.actors = []


class Actor:
    def __init__(self, asset, pose, name, group, filt):
        self.asset, self.name, self.group, self.filter = asset, name, group, filt
        self.pose = pose
        m = asset.model
        self.dof_props = np.zeros(m.num_dof, dtype=DOF_PROP_DTYPE)
        self.dof_props["hasLimits"] = True
        self.dof_props["lower"], self.dof_props["upper"] = m.lim_lower, m.lim_upper
        self.dof_props["driveMode"] = asset.options.default_dof_drive_mode
        self.dof_props["stiffness"], self.dof_props["damping"] = m.kp, m.kd
        self.dof_props["armature"], self.dof_props["effort"] = m.armature, m.effort
        self.shape_props = [RigidShapeProperties() for _ in range(m.num_bodies)]


class Sim:
    def __init__(self, device, params):
        self.device_index = device
        self.params = params
        self.envs = []
        self.ground_z = 0.0
        self.friction = 1.0
        self.heightfield = None
        self.native = None
        self.frame_count = 0
        self.tensors = {}


class Gym:
    # ------------------------------------------------------------------ creation (base_task.py:226-243, humanoid.py:428-437)
    def create_sim(self, compute_device=0, graphics_device=-1, type=SIM_PHYSX, params=None):
        try:
            lib = L.load()
            if lib.emloco_device_count() <= compute_device:
                print("*** emloco: no HIP device", compute_device, "-- there is no CPU pipeline")
                return None
        except L.EmlocoError as e:
            print("***", e)
            return None
        return Sim(int(compute_device), params or SimParams())

    def destroy_sim(self, sim):
        if sim.native is not None:
            sim.native.close()
            sim.native = None

    def get_sim_params(self, sim):
        return sim.params

    def set_sim_params(self, sim, params):
        sim.params = params
        if sim.native is not None:
            sim.native.set_params(_native_params(sim))

    def get_frame_count(self, sim):
        return sim.frame_count

    def add_ground(self, sim, plane_params):      # humanoid.py:483-494
        sim.ground_z = -float(plane_params.distance)
        sim.friction = float(plane_params.static_friction)

    def add_triangle_mesh(self, sim, vertices, triangles, params):   # humanoid_pedestrain_terrain.py:868-880
        """Terrain collision.  A flat mesh collides as the plane z = its height.  Any other mesh must be a height-field
        mesh (terrain_utils.convert_heightfield_to_trimesh: vertex i*cols + j carries sample [i][j], triangles
        (v00, v11, v01) / (v00, v10, v11)); it collides as that height field (emloco_sim_set_ground_heightfield).  The grid
        is taken from `params.heightfield` = dict(samples, horizontal_scale, vertical_scale) when the caller attaches it (the
        task does: the slope-corrected mesh moves vertices sideways, so the spacing cannot be read back from it), else it
        is recovered from an uncorrected regular-grid mesh.  General triangle soups are rejected loudly."""
        v = np.asarray(vertices, dtype=np.float32).reshape(-1, 3)
        tri = np.asarray(triangles).reshape(-1, 3)
        sim.friction = float(params.static_friction)
        zmin, zmax = float(v[:, 2].min()), float(v[:, 2].max())
        if zmax - zmin <= 1e-6:
            sim.ground_z = zmin + float(params.transform.p.z)
            sim.heightfield = None
            return
        if float(params.transform.p.z) != 0.0:
            raise NotImplementedError("emloco: a height-field mesh with a vertical offset is not supported")
        hf = getattr(params, "heightfield", None)
        if hf is None:
            cols = int(tri[0, 1]) - 1
            if cols < 2 or v.shape[0] % cols or not np.array_equal(tri[0], [0, cols + 1, 1]):
                raise NotImplementedError("emloco: only height-field meshes (convert_heightfield_to_trimesh layout) collide")
            rows = v.shape[0] // cols
            hs = float(v[cols, 0] - v[0, 0])
            gx, gy = np.meshgrid(np.arange(rows) * hs, np.arange(cols) * hs, indexing="ij")
            if hs <= 0 or np.abs(v[:, 0] - gx.ravel() - v[0, 0]).max() > 1e-4 * hs or np.abs(v[:, 1] - gy.ravel() - v[0, 1]).max() > 1e-4 * hs:
                raise NotImplementedError("emloco: irregular mesh -- attach params.heightfield (samples, horizontal_scale, vertical_scale)")
            z = v[:, 2].astype(np.float64)
            nz = np.abs(z[z != 0])
            vs = float(nz.min()) if nz.size else 1.0
            q = z / vs
            if np.abs(q - np.rint(q)).max() > 1e-3 or np.abs(q).max() > 32767:
                vs = float(zmax - zmin) / 30000.0
                q = z / vs
            hf = dict(samples=np.rint(q).astype(np.int16).reshape(rows, cols), horizontal_scale=hs, vertical_scale=vs,
                      origin_x=float(v[0, 0]), origin_y=float(v[0, 1]))
        hf = dict(hf)
        hf["samples"] = np.ascontiguousarray(hf["samples"], dtype=np.int16)
        if hf["samples"].size != v.shape[0]:
            raise ValueError("params.heightfield does not match the mesh: %d samples vs %d vertices" % (hf["samples"].size, v.shape[0]))
        if hf.get("move_x") is None and os.environ.get("EMLOCO_TERRAIN_RISERS", "1") != "0":
            # the slope-corrected mesh (convert_heightfield_to_trimesh(slope_threshold)): read the whole-cell vertex moves back from
            # the vertices so that the simulator collides with THIS mesh -- vertical risers -- not with the raw grid's one-cell ramps
            from .terrain_utils import mesh_vertex_moves
            try:
                mx, my = mesh_vertex_moves(v, hf["samples"].shape, hf["horizontal_scale"],
                                           (float(hf.get("origin_x", 0.0)), float(hf.get("origin_y", 0.0))))
            except ValueError as e:
                raise NotImplementedError("emloco: the mesh is not the height field's (regular or slope-corrected) mesh: %s" % e)
            if mx.any() or my.any():
                hf["move_x"], hf["move_y"] = mx, my
        hf["origin_x"] = float(hf.get("origin_x", 0.0)) + float(params.transform.p.x)
        hf["origin_y"] = float(hf.get("origin_y", 0.0)) + float(params.transform.p.y)
        sim.heightfield = hf
        sim.ground_z = 0.0

    def load_asset(self, sim, rootpath, filename, options=None):    # humanoid.py:720
        path = os.path.join(rootpath, filename)
        return Asset(load_mjcf(path), options or AssetOptions())

    def create_asset_from_model(self, sim, model, options=None):
        """Extension: an asset straight from a HumanoidModel (synthetic per-env shapes, no XML round trip)."""
        return Asset(model, options or AssetOptions())

    def get_asset_rigid_body_count(self, asset):
        return asset.model.num_bodies

    def get_asset_dof_count(self, asset):
        return asset.model.num_dof

    def get_asset_joint_count(self, asset):
        return asset.model.num_dof

    def get_asset_actuator_properties(self, asset):
        return [ActuatorProperties(e) for e in asset.model.effort]

    def get_asset_dof_properties(self, asset):
        return Actor(asset, Transform(), "", 0, 0).dof_props

    def find_asset_rigid_body_index(self, asset, name):
        return _find_name(asset.model.names, name)

    def create_asset_force_sensor(self, asset, body_idx, pose):
        asset.force_sensors.append(body_idx)
        return len(asset.force_sensors) - 1

    def create_env(self, sim, lower, upper, num_per_row):   # humanoid.py:809
        env = Env(len(sim.envs))
        env.sim = sim
        sim.envs.append(env)
        return env

    def create_actor(self, env, asset, pose, name="", group=0, filter=0, seg_id=0):   # humanoid.py:864
        if env.actors:
            raise NotImplementedError("emloco: one humanoid actor per env (marker / projectile actors are render-only)")
        env.actors.append(Actor(asset, pose, name, group, filter))
        return 0

    def enable_actor_dof_force_sensors(self, env, handle):
        return True

    def get_actor_rigid_body_properties(self, env, handle):
        return [RigidBodyProperties(m) for m in env.actors[handle].asset.model.mass]

    def get_actor_dof_properties(self, env, handle):
        return env.actors[handle].dof_props.copy()

    def set_actor_dof_properties(self, env, handle, props):     # humanoid.py:914
        env.actors[handle].dof_props = np.array(props, dtype=DOF_PROP_DTYPE)
        return True

    def get_actor_rigid_shape_properties(self, env, handle):
        return env.actors[handle].shape_props

    def set_actor_rigid_shape_properties(self, env, handle, props):   # humanoid.py:944
        env.actors[handle].shape_props = props
        return True

    def find_actor_rigid_body_handle(self, env, handle, name):   # humanoid.py:581,1273
        return _find_name(env.actors[handle].asset.model.names, name)

    def set_rigid_body_color(self, *a, **k):
        return None

    # ------------------------------------------------------------------ prepare + tensors (base_task.py:128, humanoid.py:137-148)
    def prepare_sim(self, sim):
        from ..sim import NativeSim
        if not sim.envs or any(not e.actors for e in sim.envs):
            print("*** emloco: prepare_sim needs one actor in every env")
            return False
        models = []
        for e in sim.envs:
            a = e.actors[0]
            m = a.asset.model
            if not (np.array_equal(a.dof_props["stiffness"], m.kp.astype(np.float32)) and
                    np.array_equal(a.dof_props["damping"], m.kd.astype(np.float32))):
                m = m.scaled(1.0, 1.0)
                m.kp = a.dof_props["stiffness"].astype(np.float64)
                m.kd = a.dof_props["damping"].astype(np.float64)
            m.effort = a.dof_props["effort"].astype(np.float64)
            models.append(m)
        # self-collision: create_actor(..., filter=0) enables it (humanoid.py:838-841,864), the per-shape bitmasks written by
        # set_actor_rigid_shape_properties select the pairs (humanoid.py:917-944)
        self_collision = None
        if all(e.actors[0].filter == 0 for e in sim.envs):
            from ..model import pack_self_collision
            filters = [int(getattr(sp, "filter", 0)) for sp in sim.envs[0].actors[0].shape_props]
            self_collision = pack_self_collision(models, filters=filters)
            if self_collision["pairs"].shape[0] == 0:
                self_collision = None
        try:
            sim.native = NativeSim(models, _native_params(sim), sim.device_index, self_collision=self_collision,
                                   heightfield=sim.heightfield)
        except L.EmlocoError as e:
            print("***", e)
            return False
        # the substeps of a launch as dependent workgroups of their own, up to four per env (emloco_sim_set_split: identical
        # results, shorter launch); EMLOCO_SPLIT=1 keeps one workgroup per env
        n_parts = int(os.environ.get("EMLOCO_SPLIT", "4"))
        if n_parts > 1:
            sim.native.set_split(n_parts)
        root = torch.zeros((len(sim.envs), 13), dtype=torch.float32)
        for i, e in enumerate(sim.envs):
            p, r = e.actors[0].pose.p, e.actors[0].pose.r
            root[i, 0:3] = torch.tensor([p.x, p.y, p.z])
            root[i, 3:7] = torch.tensor([r.x, r.y, r.z, r.w])
        sim.native.root_state.copy_(root)
        sim.native.refresh_bodies()
        return True

    def acquire_actor_root_state_tensor(self, sim):
        return Tensor(sim.native.root_state)

    def acquire_dof_state_tensor(self, sim):
        return Tensor(sim.native.dof_state)

    def acquire_rigid_body_state_tensor(self, sim):
        return Tensor(sim.native.rigid_body_state)

    def acquire_net_contact_force_tensor(self, sim):
        return Tensor(sim.native.contact_force)

    def acquire_dof_force_tensor(self, sim):
        return Tensor(sim.native.dof_force)

    def acquire_force_sensor_tensor(self, sim):
        n = sum(len(e.actors[0].asset.force_sensors) for e in sim.envs)
        if "force_sensor" not in sim.tensors:
            sim.tensors["force_sensor"] = torch.zeros((max(n, 1), 6), dtype=torch.float32, device=sim.native.device)
        return Tensor(sim.tensors["force_sensor"])

    # The state tensors are live views of the simulator's buffers: refresh is a no-op that reports success
    def refresh_actor_root_state_tensor(self, sim):
        return True

    refresh_dof_state_tensor = refresh_rigid_body_state_tensor = refresh_net_contact_force_tensor = \
        refresh_dof_force_tensor = refresh_force_sensor_tensor = refresh_actor_root_state_tensor

    # ------------------------------------------------------------------ stepping (humanoid.py:1201-1207, base_task.py:258,792-797)
    def set_dof_position_target_tensor(self, sim, tensor):
        try:
            sim.native.set_pd_targets(_as_torch(tensor))
        except L.EmlocoError:
            return False
        return True

    def set_dof_actuation_force_tensor(self, sim, tensor):    # humanoid.py:1206-1207 (`pdControl: False`: DOF_MODE_EFFORT drives)
        try:
            sim.native.set_dof_actuation_force(_as_torch(tensor))
        except L.EmlocoError:
            return False                                      # the gym API reports failure by value, it never raises
        return True

    def simulate(self, sim):
        sim.native.step(1)
        sim.frame_count += 1

    def simulate_n(self, sim, n_calls):
        """Extension: `n_calls` consecutive gym.simulate calls fused into one launch (base_task.py:792-797 loop)."""
        sim.native.step(n_calls)
        sim.frame_count += n_calls

    def simulate_n_subset(self, sim, n_calls, skip=None, ids=None, count=True):
        """Extension: simulate_n for a subset of the envs (NativeSim.step_subset); the two halves of one step count once."""
        sim.native.step_subset(n_calls, skip=skip, ids=ids)
        if count:
            sim.frame_count += n_calls

    def fetch_results(self, sim, wait=True):
        """Stream-ordered: consumers on torch's current stream need no host wait; `wait=True` keeps the reference's
        blocking semantics only when EMLOCO_BLOCKING_FETCH=1."""
        if wait and os.environ.get("EMLOCO_BLOCKING_FETCH", "0") == "1":
            sim.native.sync()

    def set_actor_root_state_tensor_indexed(self, sim, tensor, actor_ids, n):   # humanoid.py:470-472
        sim.native.set_root_state_indexed(_as_torch(tensor), _as_torch(actor_ids)[:n])
        return True

    def set_dof_state_tensor_indexed(self, sim, tensor, actor_ids, n):   # humanoid.py:473-475
        sim.native.set_dof_state_indexed(_as_torch(tensor), _as_torch(actor_ids)[:n])
        return True

    # ------------------------------------------------------------------ viewer / camera: headless no-ops (base_task.py:136-222)
    def create_viewer(self, *a, **k):
        return None

    def viewer_camera_look_at(self, *a, **k):
        return None

    def create_camera_sensor(self, *a, **k):
        return -1

    def set_light_parameters(self, *a, **k):
        return None

    def clear_lines(self, *a, **k):
        return None

    def add_lines(self, *a, **k):
        return None

    def step_graphics(self, *a, **k):
        return None

    def draw_viewer(self, *a, **k):
        return None

    def sync_frame_time(self, *a, **k):
        return None

    def query_viewer_has_closed(self, *a, **k):
        return False

    def poll_viewer_events(self, *a, **k):
        return None

    def query_viewer_action_events(self, *a, **k):
        return []


def _find_name(names, name):
    low = [n.lower() for n in names]
    return low.index(name.lower()) if name.lower() in low else -1


def _as_torch(t):
    return t._t if isinstance(t, Tensor) else t


def _native_params(sim):
    p = sim.params
    return L.default_sim_params(n_sub=int(p.substeps), n_iter=int(p.physx.num_position_iterations),
                                h=float(p.dt) / int(p.substeps), gravity_z=float(p.gravity.z),
                                contact_offset=float(p.physx.contact_offset),
                                max_depen_vel=float(p.physx.max_depenetration_velocity), mu=float(sim.friction),
                                ang_damping=float(sim.envs[0].actors[0].asset.options.angular_damping) if sim.envs else 0.01,
                                max_ang_vel=float(sim.envs[0].actors[0].asset.options.max_angular_velocity) if sim.envs else 100.0,
                                ground_z=float(sim.ground_z))


_GYM = None


def acquire_gym():    # base_task.py:59
    global _GYM
    if _GYM is None:
        _GYM = Gym()
    return _GYM
