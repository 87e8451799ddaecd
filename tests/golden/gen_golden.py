#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own pure-torch code in this container.

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

The reference (/root/reference) is imported read-only through tests/golden/_ref_shim.py; only the
inputs and the outputs it produced are stored (as float32/int64 numpy arrays).  Nothing under
/root/reference is copied.  The .npz files are what the oracle (oracle/) is pinned against in
`tests/test_oracle_golden.py`, and what the HIP path is compared to on the GPU.

Each fixture records `torch_version` because the arithmetic of the reference is torch's.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim  # noqa: E402

OUT = HERE


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    out["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: tuple(v.shape) for k, v in out.items() if k != "torch_version"})


def rand_quat(g, *shape):
    q = torch.randn(*shape, 4, generator=g)
    return q / q.norm(dim=-1, keepdim=True)


def random_body_state(g, E, nb=24):
    """Plausible humanoid-like states: bodies scattered ~1 m around a root somewhere on the map."""
    root = torch.stack([torch.rand(E, generator=g) * 40 + 30, torch.rand(E, generator=g) * 40 + 30,
                        torch.rand(E, generator=g) * 0.4 + 0.7], dim=-1)
    pos = root[:, None, :] + torch.randn(E, nb, 3, generator=g) * 0.4
    pos[:, 0] = root
    rot = rand_quat(g, E, nb)
    # make most roots nearly upright (yaw + small tilt) like the real task, keep a few arbitrary
    yaw = (torch.rand(E, generator=g) * 2 - 1) * np.pi
    tilt = torch.randn(E, 3, generator=g) * 0.1
    qz = torch.stack([torch.zeros(E), torch.zeros(E), torch.sin(yaw / 2), torch.cos(yaw / 2)], -1)
    qt = torch.cat([tilt * 0.5, torch.ones(E, 1)], -1)
    qt = qt / qt.norm(dim=-1, keepdim=True)
    from isaacgym.torch_utils import quat_mul
    up = quat_mul(qz, qt)
    keep = E // 4
    rot[keep:, 0] = up[keep:]
    vel = torch.randn(E, nb, 3, generator=g) * 1.5
    ang = torch.randn(E, nb, 3, generator=g) * 3.0
    return pos.float(), rot.float(), vel.float(), ang.float()



def terrain_index_map(rows=1080, cols=1080):
    """The int16 height map of the terrain_index fixture, from a formula (not stored): 8 x 8 plateaus of hashed height plus
    a hashed +-3 ripple, so neighbouring cells differ almost everywhere and a one-cell index slip changes the height."""
    i = np.arange(rows, dtype=np.int64)[:, None]
    j = np.arange(cols, dtype=np.int64)[None, :]
    coarse = (((i // 8) * 73856093) ^ ((j // 8) * 19349663)) % 600 - 200
    fine = ((i * 83492791) ^ (j * 2971215073)) % 7 - 3
    return (coarse + fine).astype(np.int16)


def gen_terrain_index():
    """A10 at scale (256 envs x (1024 + 9) probes on the 1080 x 1080 map of `small_terrain`): the integer map indices
    world_points_to_map returns (humanoid_pedestrain_terrain.py:1212-1218) and the sampled heights, for array_equal tests."""
    import env.tasks.humanoid_pedestrain_terrain as HPT
    from utils import torch_utils as TU
    g = torch.Generator().manual_seed(4321)
    E = 256
    terr = HPT.Terrain.__new__(HPT.Terrain)
    terr.horizontal_scale, terr.vertical_scale, terr.device = 0.1, 0.005, "cpu"
    terr.heightsamples = torch.from_numpy(terrain_index_map())
    pos = torch.rand(E, 3, generator=g) * torch.tensor([104.0, 104.0, 0.6]) + torch.tensor([1.0, 1.0, 0.8])
    pos[0, 0], pos[1, 1], pos[2, 0], pos[3, 1] = -1.5, -0.7, 108.9, 109.3           # off the map: index clipping
    pos[4, :2] = torch.tensor([50.0, 55.0])                                          # flags.fixed spawn: on cell boundaries
    rot = rand_quat(g, E)
    yaw = (torch.rand(E, generator=g) * 2 - 1) * np.pi
    up = torch.stack([torch.randn(E, generator=g) * 0.05, torch.randn(E, generator=g) * 0.05, torch.sin(yaw / 2), torch.cos(yaw / 2)], -1)
    rot[E // 8:] = (up / up.norm(dim=-1, keepdim=True))[E // 8:]
    rot[5] = torch.tensor([0.0, 0.0, 0.0, 1.0])                                      # axis-aligned grid
    rot[6] = torch.tensor([0.0, 0.0, 1.0, 0.0])
    rot[7] = torch.tensor([0.0, 0.0, 0.70710678, 0.70710678])
    head_pose = torch.cat([pos, rot], 1)
    root_states = torch.cat([pos, rot, torch.zeros(E, 6)], 1)
    fake = SimpleNamespace(cfg={"env": {"terrain": {"terrainType": "trimesh"}}}, num_envs=E, device="cpu", sensor_extent=2,
                           sensor_res=32, smpl_humanoid=True, _has_upright_start=True, velocity_map=False, _divide_group=False,
                           _group_obs=False, _disable_group_obs=False, terrain=terr)
    fake.height_points = HPT.HumanoidPedestrianTerrain.init_square_height_points(fake)
    fake.center_height_points = HPT.HumanoidPedestrianTerrain.init_center_height_points(fake)
    HPT.flags.divide_group = False
    heights = HPT.HumanoidPedestrianTerrain.get_heights(fake, head_pose.clone(), None)
    center = HPT.HumanoidPedestrianTerrain.get_center_heights(fake, root_states.clone(), None)
    # the indices, by the reference's own expressions (get_heights :778-787, get_center_heights :747-750, world_points_to_map)
    heading_rot = TU.calc_heading_quat(rot)
    pts = HPT.quat_apply(heading_rot.repeat(1, 1024).reshape(-1, 4), fake.height_points) + pos.unsqueeze(1)
    px, py = terr.world_points_to_map(pts.clone())
    cpts = HPT.quat_apply_yaw(rot.repeat(1, 9), fake.center_height_points) + pos.unsqueeze(1)
    cpx, cpy = terr.world_points_to_map(cpts.clone())
    hs = terr.heightsamples
    assert torch.equal(torch.min(hs[px, py], hs[px + 1, py + 1]) * terr.vertical_scale, heights.view(-1))
    save("terrain_index", head_pose=head_pose, root_states=root_states, heading_rot=heading_rot,
         px=px.view(E, 1024).short(), py=py.view(E, 1024).short(), cpx=cpx.view(E, 9).short(), cpy=cpy.view(E, 9).short(),
         heights_raw=torch.round(heights / 0.005).short(), center_raw=torch.round(center / 0.005).short(),
         heights_sample=heights[:4], map_checksum=np.array(int(terrain_index_map().astype(np.int64).sum())))


def gen_scheduler():
    """LocoVal learning-rate schedule (pacer/pacer/learning/scheduler.py, stepped once per epoch, common_agent.py:95,209)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_scheduler", f"{_ref_shim.REF}/pacer/pacer/learning/scheduler.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for tag, (w, T, n) in dict(short=(5, 40, 130), locoval=(20, 20000, 80)).items():
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=1e-4)
        sch = mod.CosineAnnealingLR(opt, warmup_epochs=w, max_epochs=T)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(n):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        out["lr_" + tag] = np.array(lrs, np.float64)
        out["cfg_" + tag] = np.array([w, T, n])
    save("locoval_lr_schedule", **out)

def gen_pacer():
    _ref_shim.install_pacer()
    import env.tasks.humanoid as H
    import env.tasks.humanoid_amp as HA
    import env.tasks.humanoid_pedestrain_terrain as HPT
    import env.util.traj_generator as TG
    from utils import torch_utils as TU
    from utils.flags import flags
    import isaacgym.torch_utils as ITU

    g = torch.Generator().manual_seed(1234)

    # ---------------------------------------------------------------- quaternion helpers
    N = 64
    q = rand_quat(g, N)
    q2 = rand_quat(g, N)
    v = torch.randn(N, 3, generator=g)
    em = torch.randn(N, 3, generator=g) * 1.2
    em[:4] = 0.0            # |theta| <= 1e-5 branch -> default axis z
    em[4:8] *= 1e-6
    em[8:12] *= 4.0         # |theta| > pi -> wrapped through atan2(sin, cos)
    t = torch.rand(N, 1, generator=g)
    ang = (torch.rand(N, generator=g) * 2 - 1) * 3.0
    save("quat_utils",
         q=q, q2=q2, v=v, exp_map=em, t=t, angle=ang,
         my_quat_rotate=TU.my_quat_rotate(q, v),
         quat_mul=ITU.quat_mul(q, q2),
         quat_apply=ITU.quat_apply(q, v),
         quat_rotate_inverse=ITU.quat_rotate_inverse(q, v),
         calc_heading=TU.calc_heading(q),
         calc_heading_quat=TU.calc_heading_quat(q),
         calc_heading_quat_inv=TU.calc_heading_quat_inv(q),
         quat_to_tan_norm=TU.quat_to_tan_norm(q),
         exp_map_to_quat=TU.exp_map_to_quat(em),
         quat_to_exp_map=TU.quat_to_exp_map(q),
         slerp=TU.slerp(q, q2, t),
         quat_from_angle_axis=ITU.quat_from_angle_axis(ang, v),
         quat_apply_yaw=HPT.quat_apply_yaw(q.clone(), v),
         normalize_angle=ITU.normalize_angle(ang * 3))

    # ---------------------------------------------------------------- self obs (A7) + flip obs (A11)
    E = 16
    pos, rot, vel, angv = random_body_state(g, E)
    betas = torch.randn(E, 17, generator=g)
    betas[:, 0] = torch.randint(0, 3, (E,), generator=g).float()
    limb = torch.zeros(E, 10)
    obs = H.compute_humanoid_observations_smpl_max(pos, rot, vel, angv, betas, limb,
                                                   True, False, True, True, False)
    l2r = [0, 5, 6, 7, 8, 1, 2, 3, 4, 9, 10, 11, 12, 13, 19, 20, 21, 22, 23, 14, 15, 16, 17, 18]
    fake = SimpleNamespace(_rigid_body_pos=pos, _rigid_body_rot=rot, _rigid_body_vel=vel,
                           _rigid_body_ang_vel=angv, left_to_right_index=l2r, smpl_humanoid=True,
                           humanoid_betas=betas, humanoid_limb_and_weights=limb, _local_root_obs=True,
                           _root_height_obs=False, _has_upright_start=True, _has_shape_obs=True,
                           _has_limb_weight_obs=False, device="cpu")
    flip = H.Humanoid._compute_flip_humanoid_obs(fake, None)
    save("self_obs", body_pos=pos, body_rot=rot, body_vel=vel, body_ang_vel=angv, betas=betas,
         obs=obs, flip_obs=flip)

    # ---------------------------------------------------------------- trajectory generator (A8, A16)
    for k, val in dict(real_path=False, jta_path=False, jrdb_path=False, pred_path=False, fixed_path=False,
                       slow=False, adjust_root_vel=False, init_heading=False, heading_inversion=False,
                       add_noise=False, vru=False).items():
        setattr(flags, k, val)
    dt = 2 * (1.0 / 60.0)
    episode_len = 168
    tg = TG.TrajGenerator(E, episode_len * dt, 101, "cpu", 2.0, 0.0005, 3.0, 2.0, 0.02, None,
                          hybridInitProb=0.5, flags=flags)
    env_ids = torch.arange(E, dtype=torch.long)
    root_vel = torch.randn(E, 3, generator=g)
    torch.manual_seed(77)
    st = torch.get_rng_state()
    tg.reset(env_ids, pos[:, 0].clone(), root_vel.clone())
    verts_plain = tg._verts.clone()
    # replay of the draws, in the reference's call order, so the host mirror can be pinned without
    # depending on torch's generator implementation
    torch.set_rng_state(st)
    r1 = torch.rand([E, 100]); r2 = torch.rand([E, 100])
    bern = torch.bernoulli(0.02 * torch.ones(E, 100))
    r3 = torch.rand([E]); r4 = torch.rand([E, 100]); r5 = torch.rand([E])
    save("traj_reset_plain", init_pos=pos[:, 0], root_vel=root_vel, rng_seed=np.array(77),
         r_dtheta=r1, r_dtheta_sharp=r2, bern_sharp=bern, r_heading=r3, r_dspeed=r4, r_speed0=r5,
         verts=verts_plain, dt_vert=np.array(tg._dt, dtype=np.float64))

    # with init_heading + heading_inversion + adjust_root_vel (LocoVal-training flags)
    flags.init_heading = True
    flags.heading_inversion = True
    flags.adjust_root_vel = True
    tg2 = TG.TrajGenerator(E, episode_len * dt, 101, "cpu", 2.0, 0.0005, 3.0, 2.0, 0.02, None,
                           hybridInitProb=0.5, flags=flags)
    torch.manual_seed(78)
    st = torch.get_rng_state()
    tg2.reset(env_ids, pos[:, 0].clone(), root_vel.clone())
    torch.set_rng_state(st)
    r1 = torch.rand([E, 100]); r2 = torch.rand([E, 100])
    bern = torch.bernoulli(0.02 * torch.ones(E, 100))
    r3 = torch.rand([E]); r4 = torch.rand([E, 100]); r5 = torch.rand([E])
    r6 = torch.rand(E)
    save("traj_reset_heading", init_pos=pos[:, 0], root_vel=root_vel, rng_seed=np.array(78),
         r_dtheta=r1, r_dtheta_sharp=r2, bern_sharp=bern, r_heading=r3, r_dspeed=r4, r_speed0=r5,
         r_inversion=r6, verts=tg2._verts.clone(), inverted=tg2.inverted.clone().long(),
         dt_vert=np.array(tg2._dt, dtype=np.float64))

    # real-world paths (traj_generator.py:120-160; configs[1] runs with --real_path JTA+JRDB): the constructor loads the
    # pickles from hard-coded relative paths, so the generator is built with the flag off and handed the tables the way
    # its constructor stores them (:43-52).  Synthetic tables in the saved-traj format {id: {'pose', 'traj' (>=101,3) f64}}.
    import random as pyrandom

    def fake_paths(n, seed, nv):
        r = np.random.RandomState(seed)
        out = {}
        for i in range(n):
            speed = r.uniform(0.05, 2.5)
            turn = r.uniform(-0.5, 0.5)
            tt = np.arange(nv) * (168 * dt / 100)
            th = r.uniform(-np.pi, np.pi) + turn * tt
            xy = np.cumsum(np.stack([np.cos(th), np.sin(th)], -1) * speed * (168 * dt / 100), 0) + r.uniform(-30, 30, 2)
            z = np.full((nv, 1), r.uniform(-0.05, 0.05))
            out[i] = {"pose": None, "traj": np.concatenate([xy, z], -1)}
        out[0]["traj"][1] = out[0]["traj"][0]                 # a standing start: first segment of zero length
        return out

    jta, jrdb = fake_paths(23, 5, 101), fake_paths(17, 6, 108)
    for tag, use_jrdb, adj in (("real1", False, False), ("real2", True, True), ("real2_noadj", True, False)):
        flags.real_path = False
        flags.init_heading = True
        flags.heading_inversion = True
        flags.adjust_root_vel = adj
        tg3 = TG.TrajGenerator(E, episode_len * dt, 101, "cpu", 2.0, 0.0005, 3.0, 2.0, 0.02, None,
                               hybridInitProb=0.5, flags=flags)
        flags.real_path, flags.jta_path, flags.jrdb_path = True, True, use_jrdb
        tg3.traj_data_jta, tg3.traj_data = jta, [jta]
        if use_jrdb:
            tg3.traj_data_jrdb = jrdb
            tg3.traj_data.append(jrdb)
        torch.manual_seed(90)
        pyrandom.seed(91)
        st, pst = torch.get_rng_state(), pyrandom.getstate()
        tg3.reset(env_ids, pos[:, 0].clone(), root_vel.clone())
        torch.set_rng_state(st)
        pyrandom.setstate(pst)
        r1 = torch.rand([E, 100]); r2 = torch.rand([E, 100])
        bern = torch.bernoulli(0.02 * torch.ones(E, 100))
        r3 = torch.rand([E]); r4 = torch.rand([E, 100]); r5 = torch.rand([E])
        r_real = torch.rand(E)
        n_real = int((r_real > 0.5).sum())
        rids = pyrandom.sample(range(len(jta) + (len(jrdb) if use_jrdb else 0)), n_real)
        r6 = torch.rand(E)
        table = np.stack([v["traj"][:101] for v in jta.values()] + ([v["traj"][:101] for v in jrdb.values()] if use_jrdb else []))
        save("traj_reset_" + tag, init_pos=pos[:, 0], root_vel=root_vel, r_dtheta=r1, r_dtheta_sharp=r2, bern_sharp=bern,
             r_heading=r3, r_dspeed=r4, r_speed0=r5, r_real=r_real, real_rids=np.array(rids, np.int64), r_inversion=r6,
             real_table=table.astype(np.float64), n_jta=np.array(len(jta)), adjust_root_vel=np.array(adj),
             verts=tg3._verts.clone(), inverted=tg3.inverted.clone().long(), dt_vert=np.array(tg3._dt, dtype=np.float64))
    flags.real_path = flags.jta_path = flags.jrdb_path = False
    flags.init_heading = False
    flags.heading_inversion = False
    flags.adjust_root_vel = False

    # calc_pos and _fetch_traj_samples
    progress = torch.randint(0, 168, (E,), generator=g)
    progress[0] = 0
    progress[1] = 167
    progress[2] = 166
    times = progress * dt
    tar = tg.calc_pos(env_ids, times)
    fake = SimpleNamespace(num_envs=E, device="cpu", progress_buf=progress, dt=dt, _num_traj_samples=15,
                           _traj_sample_timestep=0.4, _traj_gen=tg)
    import env.tasks.humanoid_traj as HT
    samples = HT.HumanoidTraj._fetch_traj_samples(fake, None)
    root_states = torch.cat([pos[:, 0], rot[:, 0], vel[:, 0], angv[:, 0]], dim=-1)
    loc_obs = HPT.compute_location_observations(root_states, samples, True)
    save("traj_samples", verts=verts_plain, progress=progress, dt=np.array(dt, dtype=np.float64),
         traj_dur=np.array(tg.get_traj_duration(), dtype=np.float64),
         tar_pos=tar, samples=samples, root_states=root_states, loc_obs=loc_obs)

    # ---------------------------------------------------------------- terrain height sampling (A10)
    terr = HPT.Terrain.__new__(HPT.Terrain)
    terr.horizontal_scale = 0.1
    terr.vertical_scale = 0.005
    terr.device = "cpu"
    rows, cols = 256, 192  # a 25.6 m x 19.2 m random map (non-square on purpose)
    hf = (torch.randint(-200, 400, (rows // 8, cols // 8), generator=g)).short()
    hf = hf.repeat_interleave(8, 0).repeat_interleave(8, 1)
    hf = (hf + torch.randint(-3, 4, (rows, cols), generator=g).short()).short()
    terr.heightsamples = hf
    head_idx = 13
    shift = torch.tensor([-28.0, -30.0, 0.0])   # bring the 30..70 m states onto the small map
    head_pose = torch.cat([(pos[:, head_idx] + shift) * torch.tensor([0.5, 0.4, 1.0]), rot[:, head_idx]], dim=1)
    # push two envs off the map to exercise the index clipping
    head_pose[0, 0] = -3.0
    head_pose[1, 1] = 200.0
    fake_hpt = SimpleNamespace(cfg={"env": {"terrain": {"terrainType": "trimesh"}}}, num_envs=E, device="cpu",
                               sensor_extent=2, sensor_res=32, smpl_humanoid=True, _has_upright_start=True,
                               velocity_map=False, _divide_group=False, _group_obs=False,
                               _disable_group_obs=False, terrain=terr)
    fake_hpt.height_points = HPT.HumanoidPedestrianTerrain.init_square_height_points(fake_hpt)
    fake_hpt.center_height_points = HPT.HumanoidPedestrianTerrain.init_center_height_points(fake_hpt)
    flags.divide_group = False
    HPT.flags.divide_group = False
    heights = HPT.HumanoidPedestrianTerrain.get_heights(fake_hpt, head_pose.clone(), None)
    root_states_t = root_states.clone()
    root_states_t[:, :3] = (root_states_t[:, :3] + shift) * torch.tensor([0.5, 0.4, 1.0])
    center = HPT.HumanoidPedestrianTerrain.get_center_heights(fake_hpt, root_states_t, None)
    center_mean = center.mean(dim=-1, keepdim=True)
    height_obs = torch.clip(center_mean - heights, -3, 3.) * 5
    save("terrain_heights", heightfield=hf, head_pose=head_pose, root_states=root_states_t,
         height_points=fake_hpt.height_points[0], center_height_points=fake_hpt.center_height_points[0],
         heights=heights, center_heights=center, height_obs=height_obs)

    # flip of the task obs (A11)
    task_obs = torch.cat([loc_obs, height_obs], dim=1)
    fake_flip = SimpleNamespace(_num_traj_samples=15, terrain_obs=True, velocity_map=False,
                                num_height_points=1024, _divide_group=False, _group_obs=False)
    flip_task = HPT.HumanoidPedestrianTerrain._compute_flip_task_obs(fake_flip, task_obs.clone(), None)
    save("task_obs_flip", task_obs=task_obs, flip_task_obs=flip_task)

    # ---------------------------------------------------------------- reward (A12) + reset (A13)
    E2 = 64
    pos2, rot2, vel2, ang2 = random_body_state(g, E2)
    tar2 = pos2[:, 0].clone()
    tar2[:, :2] += torch.randn(E2, 2, generator=g) * torch.tensor([0.3, 0.3])
    tar2[:8, :2] += 4.5  # some beyond the 4 m fail distance
    tar2[8, 0] = pos2[8, 0, 0] + 4.0   # on the threshold (dist_sq ~ 16)
    tar2[8, 1] = pos2[8, 0, 1]
    dof_force = torch.randn(E2, 69, generator=g) * 40
    dof_vel = torch.randn(E2, 69, generator=g) * 2
    loc_r = HPT.compute_location_reward(pos2[:, 0], tar2)
    power = torch.abs(torch.multiply(dof_force, dof_vel)).sum(dim=-1)
    pow_r = -0.0005 * power
    contact = torch.zeros(E2, 24, 3)
    contact[:, [3, 4, 7, 8]] = torch.randn(E2, 4, 3, generator=g) * 300  # feet: ignored
    hit = torch.rand(E2, generator=g) < 0.4
    contact[hit, 11] = torch.randn(int(hit.sum()), 3, generator=g) * 40
    contact[9, 13] = torch.tensor([30.0, 40.0, 0.0])       # |F| == 50 exactly: not > 50
    contact[10, 13] = torch.tensor([30.0, 40.0, 0.01])
    progress2 = torch.randint(0, 170, (E2,), generator=g)
    progress2[:4] = torch.tensor([0, 1, 2, 167])
    reset_in = torch.randint(0, 2, (E2,), generator=g)
    center_h = torch.zeros(E2, 1)
    reset, term = HPT.compute_humanoid_reset(reset_in, progress2, contact,
                                             torch.tensor([7, 3, 8, 4]), center_h, pos2, tar2,
                                             168, 4.0, True, torch.full((24,), 0.15), False)
    save("reward_reset", root_pos=pos2[:, 0], body_pos=pos2, tar_pos=tar2, dof_force=dof_force, dof_vel=dof_vel,
         loc_reward=loc_r, power_reward=pow_r, rew=loc_r + pow_r,
         contact=contact, progress=progress2, reset_in=reset_in, reset=reset, terminate=term)

    # ---------------------------------------------------------------- AMP obs row (A14)
    dof_pos = torch.randn(E, 69, generator=g) * 0.6
    dof_velE = torch.randn(E, 69, generator=g) * 2
    key_ids = [7, 3, 22, 17]
    names = ['L_Hip', 'L_Knee', 'L_Ankle', 'L_Toe', 'R_Hip', 'R_Knee', 'R_Ankle', 'R_Toe', 'Torso', 'Spine',
             'Chest', 'Neck', 'Head', 'L_Thorax', 'L_Shoulder', 'L_Elbow', 'L_Wrist', 'L_Hand', 'R_Thorax',
             'R_Shoulder', 'R_Elbow', 'R_Wrist', 'R_Hand']
    subset = np.concatenate([np.arange(i * 3, i * 3 + 3) for i, n in enumerate(names)
                             if n not in ("L_Hand", "R_Hand", "L_Toe", "R_Toe")])
    amp = HA.build_amp_observations_smpl(pos[:, 0], rot[:, 0], vel[:, 0], angv[:, 0], dof_pos, dof_velE,
                                         pos[:, key_ids], betas, limb, torch.from_numpy(subset), True, False,
                                         True, True, False, True)
    save("amp_obs", root_pos=pos[:, 0], root_rot=rot[:, 0], root_vel=vel[:, 0], root_ang_vel=angv[:, 0],
         dof_pos=dof_pos, dof_vel=dof_velE, key_pos=pos[:, key_ids], betas=betas, dof_subset=subset,
         amp_obs=amp)

    # ---------------------------------------------------------------- PD target map (A1)
    lim = np.deg2rad(np.array([180.0] * 69))
    for j, n in enumerate(names):
        if "Shoulder" in n or "Elbow" in n:
            lim[j * 3:j * 3 + 3] = np.deg2rad(720.0)
    fake_pd = SimpleNamespace(_dof_offsets=np.linspace(0, 69, 24).astype(int), _bias_offset=False,
                              dof_limits_lower=torch.tensor(-lim, dtype=torch.float32),
                              dof_limits_upper=torch.tensor(lim, dtype=torch.float32),
                              device="cpu", smpl_humanoid=True, _dof_names=names,
                              _has_smpl_pd_offset=False, _has_upright_start=True)
    H.Humanoid._build_pd_action_offset_scale(fake_pd)
    actions = torch.randn(E, 69, generator=g) * 0.3
    pd_tar = H.Humanoid._action_to_pd_targets(fake_pd, actions)
    for n in ("L_Hand", "R_Hand", "L_Toe", "R_Toe"):
        i = names.index(n) * 3
        pd_tar[:, i:i + 3] = 0
    save("pd_targets", lim_lower=-lim.astype(np.float32), lim_upper=lim.astype(np.float32),
         offset=fake_pd._pd_action_offset, scale=fake_pd._pd_action_scale, actions=actions, pd_tar=pd_tar)

    # ---------------------------------------------------------------- motion-lib frame blend (A16)
    from utils.motion_lib_smpl import MotionLib as MotionLibSMPL
    ml = MotionLibSMPL.__new__(MotionLibSMPL)
    mlen = torch.rand(E, generator=g) * 5 + 2
    nfr = (mlen * 30).long() + 1
    mdt = torch.full((E,), 1 / 30.)
    mtime = torch.rand(E, generator=g) * 8 - 0.5
    i0, i1, blend = ml._calc_frame_blend(mtime, mlen, nfr, mdt)
    save("frame_blend", time=mtime, length=mlen, num_frames=nfr, dt=mdt, idx0=i0, idx1=i1, blend=blend)

    # ---------------------------------------------------------------- LocoVal MLP (A18 / B7)
    from learning.value_pose_net import ValuePoseNet
    torch.manual_seed(5)
    net = ValuePoseNet(use_pose=True, use_vel=True)
    B = 8
    traj = torch.cumsum(torch.randn(B, 13, 3, generator=g) * 0.3 + torch.tensor([0.5, 0.1, 0.0]), dim=1)
    traj[:, 0] = 0
    traj[0, 1, 0] = 0.0   # exercises the epsilon guard on x
    pose = torch.randn(B, 24, 3, generator=g) * 0.3
    velB = torch.randn(B, 2, generator=g)
    traj_req = traj.clone().requires_grad_(True)
    pose_in = pose.clone()
    value, loss = net.calc_embodied_motion_loss(traj_req, pose_in, velB.clone())
    loss.backward()
    grads = {("grad_" + k.replace(".", "_")): p.grad for k, p in net.named_parameters()}
    # sum-reduction fit used while training LocoVal in the rollout (amp_continuous_value.py:123-145)
    net.zero_grad()
    target = torch.rand(B, 1, generator=g)
    v2 = net(traj.clone(), pose.clone(), velB.clone())
    fit_loss = torch.nn.MSELoss(reduction="sum")(v2, target)
    fit_loss.backward()
    fgrads = {("fitgrad_" + k.replace(".", "_")): p.grad for k, p in net.named_parameters()}
    save("locoval", traj=traj, pose=pose, vel=velB, pose_after_inplace=pose_in,
         value=value, loss=loss, grad_traj=traj_req.grad, target=target, fit_value=v2, fit_loss=fit_loss,
         **{k.replace(".", "_"): p for k, p in net.state_dict().items()}, **grads, **fgrads)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "pacer"
    if which == "pacer":
        gen_pacer()
        gen_terrain_index()
        gen_scheduler()
    elif which == "scheduler":
        gen_scheduler()
    elif which == "terrain_index":
        _ref_shim.install_pacer()
        gen_terrain_index()
    elif which == "predictor":
        from gen_golden_predictor import gen_predictor
        gen_predictor()
