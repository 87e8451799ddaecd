"""Shared scene builders for the parity tests (oracle vs HIP / emulated HIP)."""
import numpy as np

from emloco_amd.model import pack_models, smpl_humanoid


def varied_models(E, seed=0):
    """Synthetic 'AMASS-shaped' humanoids: limb scale U[0.9,1.1], mass scale U[0.7,1.4], PD gains x mass/77
    (humanoid.py:905-911)."""
    rng = np.random.default_rng(seed)
    base = smpl_humanoid()
    out = []
    for e in range(E):
        m = base.scaled(rng.uniform(0.9, 1.1), rng.uniform(0.7, 1.4)) if e else base.scaled(1.0, 1.0)
        s = m.total_mass() / 77.0
        m.kp = m.kp * s
        m.kd = m.kd * s
        out.append(m)
    return out


def scene_state(E, seed=1, height=0.93, perturbed_from=1):
    """root/dof/target arrays: env 0 stands still, the others start perturbed."""
    rng = np.random.default_rng(seed)
    root = np.zeros((E, 13), np.float32)
    root[:, 6] = 1.0
    root[:, 2] = height
    root[:, 0] = np.arange(E) * 0.5 + 50.0
    root[:, 1] = 55.0
    dof = np.zeros((E, 69, 2), np.float32)
    tgt = np.zeros((E, 69), np.float32)
    for e in range(perturbed_from, E):
        dof[e, :, 0] = rng.normal(size=69) * 0.15
        dof[e, :, 1] = rng.normal(size=69) * 0.8
        root[e, 7:13] = rng.normal(size=6) * 0.4
        yaw = rng.uniform(-np.pi, np.pi)
        root[e, 3:7] = [0, 0, np.sin(yaw / 2), np.cos(yaw / 2)]
        tgt[e] = rng.normal(size=69) * 0.15
    return root, dof, tgt


def bumpy_heightfield(n=1100, seed=0, amp=0.08, slope=0.0):
    """int16 field (0.1 m grid, 0.005 m units): smooth random bumps of +-amp metres over a constant slope along x,
    covering the 50..60 m region the test scenes stand in."""
    rng = np.random.default_rng(seed)
    x = np.arange(n)[:, None] * 0.1
    y = np.arange(n)[None, :] * 0.1
    z = slope * (x - 50.0) + 0 * y
    for _ in range(6):
        kx, ky, ph = rng.uniform(0.5, 4.0), rng.uniform(0.5, 4.0), rng.uniform(0, 6.28)
        z = z + (amp / 3) * np.sin(kx * x + ky * y + ph)
    return dict(samples=np.rint(z / 0.005).astype(np.int16), horizontal_scale=0.1, vertical_scale=0.005)


def oracle_sim(models, root, dof, tgt, self_collision=None, heightfield=None, **params):
    import oracle
    if self_collision is True:
        from emloco_amd.model import pack_self_collision
        self_collision = pack_self_collision(models)
    s = oracle.Sim(pack_models(models), oracle.default_params(**params), self_collision=self_collision, heightfield=heightfield)
    s.root_state[:] = root
    s.dof_state[:] = dof
    s.pd_target[:] = tgt
    return s


TRAJ_CASES = {      # golden fixture -> TrajGenerator flags (tests/golden/gen_golden.py)
    "traj_reset_plain": dict(),
    "traj_reset_heading": dict(init_heading=True, heading_inversion=True, adjust_root_vel=True),
    "traj_reset_real1": dict(init_heading=True, heading_inversion=True, real_path=True, jta_path=True),
    "traj_reset_real2": dict(init_heading=True, heading_inversion=True, adjust_root_vel=True, real_path=True, jta_path=True, jrdb_path=True),
    "traj_reset_real2_noadj": dict(init_heading=True, heading_inversion=True, real_path=True, jta_path=True, jrdb_path=True),
}


def traj_rnd_rows(g, E=16):
    """The reference's draws of one TrajGenerator.reset (a golden fixture) laid out as the device's random rows
    (include/emloco_task.h: EMLOCO_RND_*); the Bernoulli outcome becomes a uniform on the right side of sharp_prob."""
    from emloco_amd import _lib as L
    rnd = np.full((E, L.RESET_RND), 0.25, np.float32)
    rnd[:, L.RND_DTHETA:L.RND_DTHETA + 100] = g["r_dtheta"]
    rnd[:, L.RND_SHARP:L.RND_SHARP + 100] = g["r_dtheta_sharp"]
    rnd[:, L.RND_BERN:L.RND_BERN + 100] = np.where(g["bern_sharp"] == 1.0, 0.0, 1.0)
    rnd[:, L.RND_DSPEED:L.RND_DSPEED + 100] = g["r_dspeed"]
    rnd[:, L.RND_HEADING] = g["r_heading"]
    rnd[:, L.RND_SPEED0] = g["r_speed0"]
    if "r_real" in g:
        rnd[:, L.RND_REAL] = g["r_real"]
    if "r_inversion" in g:
        rnd[:, L.RND_INVERSION] = g["r_inversion"]
    return rnd


def traj_real_pick(g, E=16):
    """Per-list-entry real-path rows of a fixture: the i-th env with r_real > 0.5 takes real_rids[i] (traj_generator.py:126-143)."""
    pick = np.zeros(E, np.int32)
    if "real_rids" in g:
        pick[np.nonzero(g["r_real"] > 0.5)[0]] = g["real_rids"]
    return pick


def traj_reset_bufs(flags, g, verts, inverted, real_table=None, real_pick=None, key=0, E=16):
    """EmlocoResetBufs with the trajectory fields only (emloco_task_traj_reset); `verts`, `inverted`, `real_table`, `real_pick`
    are objects with a data pointer already resolved (ints) or None."""
    from emloco_amd import _lib as L
    b = L.ResetBufs()
    fl = (L.RESET_INIT_HEADING if flags.get("init_heading") else 0) | (L.RESET_ADJUST_ROOT_VEL if flags.get("adjust_root_vel") else 0)
    fl |= L.RESET_HEADING_INVERSION if flags.get("heading_inversion") else 0
    fl |= L.RESET_REAL_PATH if flags.get("real_path") else 0
    b.flags = fl
    b.vert_dt, b.dtheta_max, b.speed_min, b.speed_max = float(g["dt_vert"]), 2.0, 0.0005, 3.0
    b.accel_max, b.sharp_prob, b.hybrid_prob = 2.0, 0.02, 0.5
    b.n_real = 0 if real_table is None else int(g["real_table"].shape[0])
    b.real_traj, b.real_pick, b.real_pick_key = real_table, real_pick, key
    b.traj_verts, b.inverted = verts, inverted
    return b


def terrain_index_map(rows=1080, cols=1080):
    """The int16 map of tests/golden/terrain_index.npz (formula of gen_golden.py: terrain_index_map; the fixture stores its checksum)."""
    i = np.arange(rows, dtype=np.int64)[:, None]
    j = np.arange(cols, dtype=np.int64)[None, :]
    coarse = (((i // 8) * 73856093) ^ ((j // 8) * 19349663)) % 600 - 200
    fine = ((i * 83492791) ^ (j * 2971215073)) % 7 - 3
    return (coarse + fine).astype(np.int16)


def corrected_stairs(n=200):
    """A staircase along x (15 cm risers every 30 cm, 4 high, repeating) crossed by a 20 cm trench along y, as the height field and as
    the vertex moves of its slope-corrected mesh (terrain_utils.convert_heightfield_to_trimesh, threshold 0.9)."""
    from emloco_amd.gym import terrain_utils as T
    steps = ((np.arange(n)[:, None] // 3) % 4 * 30 + 0 * np.arange(n)[None, :]).astype(np.int16)
    steps[:, 96:104] -= 40
    verts, _ = T.convert_heightfield_to_trimesh(steps, 0.1, 0.005, 0.9)
    mx, my = T.mesh_vertex_moves(verts, steps.shape, 0.1)
    assert (mx != 0).any() and (my != 0).any()
    return steps, dict(samples=steps, horizontal_scale=0.1, vertical_scale=0.005, move_x=mx, move_y=my)
