"""GPU parity: the fused post-physics kernel (through the C ABI) against the reference's golden vectors
and the CPU oracle.  Integer masks bit-exact; fp32 observations within 2e-5 abs / 1e-5 rel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _scene(golden):
    from emloco_amd import _lib as L
    from emloco_amd.post_physics import PostPhysics
    g, gt, gs = golden("self_obs"), golden("terrain_heights"), golden("traj_samples")
    E = 16
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    T = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    rb = np.concatenate([g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"]], -1)
    t = dict(
        rb_state=T(rb), dof_state=T(rng.normal(size=(E, 69, 2))), dof_force=T(rng.normal(size=(E, 69)) * 30),
        contact_force=T(np.zeros((E, 24, 3))), betas=T(g["betas"]), traj_verts=T(gs["verts"]),
        progress_buf=T(gs["progress"], torch.int64), reset_buf=torch.ones(E, dtype=torch.int64, device=dev),
        terminate_buf=torch.ones(E, dtype=torch.int64, device=dev),
        obs_buf=torch.zeros(E, L.OBS, device=dev), flip_obs_buf=torch.zeros(E, L.OBS, device=dev),
        rew_buf=torch.zeros(E, device=dev), reward_raw=torch.zeros(E, 2, device=dev),
        amp_obs_buf=T(rng.normal(size=(E, L.AMP_STEPS, L.AMP_ROW))),
    )
    t["contact_force"][:, 11] = T(rng.normal(size=(E, 3)) * 40)
    pp = PostPhysics(dev)
    dt = 1.0 / 30.0
    bufs = pp.make_bufs(n_env=E, heightfield=T(gt["heightfield"], torch.int16), dt=dt, traj_dur=101 * (168 * dt / 100.0),
                        sample_dt=0.4, hscale=0.1, vscale=0.005, power_coef=0.0005, fail_dist=4.0,
                        max_episode_length=168.0, **t)
    return pp, bufs, t, (g, gt, gs), dt


def test_post_physics_matches_reference_and_oracle(golden):
    import oracle
    from emloco_amd import _lib as L
    pp, bufs, t, (g, gt, gs), dt = _scene(golden)
    amp_before = t["amp_obs_buf"].clone()
    prog_before = t["progress_buf"].clone()
    pp.run(bufs, L.POST_STEP & ~L.POST_ADVANCE)
    torch.cuda.synchronize()
    obs, fobs = t["obs_buf"].cpu().numpy(), t["flip_obs_buf"].cpu().numpy()
    tol = dict(rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(obs[:, :368], g["obs"], **tol)          # reference golden
    np.testing.assert_allclose(fobs[:, :368], g["flip_obs"], **tol)    # reference golden
    root_states = np.concatenate([g["body_pos"][:, 0], g["body_rot"][:, 0], g["body_vel"][:, 0], g["body_ang_vel"][:, 0]], -1)
    loc = oracle.location_obs(root_states, gs["samples"])
    np.testing.assert_allclose(obs[:, 368:398], loc, **tol)
    head = np.concatenate([g["body_pos"][:, 13], g["body_rot"][:, 13]], -1)
    hf = gt["heightfield"]
    ho = oracle.height_obs(oracle.get_center_heights(root_states, hf), oracle.get_heights(head, hf))
    np.testing.assert_array_equal(obs[:, 398:], ho)                    # index work: bit-exact vs the oracle, no tolerated cell flips
    task = np.concatenate([obs[:, 368:398], obs[:, 398:]], 1)
    np.testing.assert_array_equal(fobs[:, 368:], oracle.flip_task_obs(task))
    tar = oracle.traj_calc_pos(gs["verts"], gs["progress"], dt, 101 * (168 * dt / 100.0))
    rew, raw = oracle.reward(g["body_pos"][:, 0], tar, t["dof_force"].cpu().numpy(), t["dof_state"][:, :, 1].cpu().numpy())
    np.testing.assert_allclose(t["rew_buf"].cpu().numpy(), rew, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(t["reward_raw"].cpu().numpy(), raw, rtol=1e-5, atol=1e-6)
    rs, tm = oracle.reset(gs["progress"], t["contact_force"].cpu().numpy(), g["body_pos"], tar)
    assert t["reset_buf"].dtype == torch.int64
    np.testing.assert_array_equal(t["reset_buf"].cpu().numpy(), rs)        # bit-exact
    np.testing.assert_array_equal(t["terminate_buf"].cpu().numpy(), tm)    # bit-exact
    assert torch.equal(t["progress_buf"], prog_before)
    amp = oracle.amp_obs(g["body_pos"][:, 0], g["body_rot"][:, 0], g["body_vel"][:, 0], g["body_ang_vel"][:, 0],
                         t["dof_state"][:, :, 0].cpu().numpy(), t["dof_state"][:, :, 1].cpu().numpy(),
                         g["body_pos"][:, [7, 3, 22, 17]], g["betas"], pp.dof_subset.cpu().numpy())
    np.testing.assert_allclose(t["amp_obs_buf"][:, 0].cpu().numpy(), amp, **tol)
    assert torch.equal(t["amp_obs_buf"][:, 1:], amp_before[:, :-1])           # history shift is a pure copy


def test_reset_masks_bit_exact_on_reference_fixture(golden):
    """The reference's own reset fixture (64 envs incl. threshold cases) through the HIP kernel."""
    from emloco_amd import _lib as L
    from emloco_amd.post_physics import PostPhysics
    g = golden("reward_reset")
    E = 64
    dev = torch.device("cuda", 0)
    T = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    rb = np.zeros((E, 24, 13), np.float32)
    rb[:, :, :3] = g["body_pos"]
    rb[:, :, 6] = 1
    # a straight 2-vertex-per-segment polyline whose point at time progress*dt is the fixture's target:
    # put every vertex at the target so calc_pos returns it whatever the phase
    verts = np.repeat(g["tar_pos"][:, None, :], 101, axis=1)
    dof_state = np.zeros((E, 69, 2), np.float32)
    dof_state[:, :, 1] = g["dof_vel"]
    t = dict(rb_state=T(rb), dof_state=T(dof_state), dof_force=T(g["dof_force"]), contact_force=T(g["contact"]),
             betas=torch.zeros(E, 17, device=dev), traj_verts=T(verts), progress_buf=T(g["progress"], torch.int64),
             reset_buf=T(g["reset_in"], torch.int64), terminate_buf=torch.zeros(E, dtype=torch.int64, device=dev),
             obs_buf=torch.zeros(E, L.OBS, device=dev), flip_obs_buf=torch.zeros(E, L.OBS, device=dev),
             rew_buf=torch.zeros(E, device=dev), reward_raw=torch.zeros(E, 2, device=dev),
             amp_obs_buf=torch.zeros(E, L.AMP_STEPS, L.AMP_ROW, device=dev))
    pp = PostPhysics(dev)
    bufs = pp.make_bufs(n_env=E, heightfield=torch.zeros(8, 8, dtype=torch.int16, device=dev), dt=1 / 30., traj_dur=5.6,
                        sample_dt=0.4, hscale=0.1, vscale=0.005, power_coef=0.0005, fail_dist=4.0, max_episode_length=168.0, **t)
    pp.run(bufs, L.POST_REWARD | L.POST_RESET)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t["reset_buf"].cpu().numpy(), g["reset"])
    np.testing.assert_array_equal(t["terminate_buf"].cpu().numpy(), g["terminate"])
    np.testing.assert_allclose(t["reward_raw"][:, 0].cpu().numpy(), g["loc_reward"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(t["reward_raw"][:, 1].cpu().numpy(), g["power_reward"], rtol=2e-5, atol=1e-6)


def test_indexed_obs_only_touches_listed_envs(golden):
    from emloco_amd import _lib as L
    pp, bufs, t, _, _ = _scene(golden)
    ids = torch.tensor([2, 9], dtype=torch.int32, device=t["obs_buf"].device)
    pp.run(bufs, L.POST_OBS, ids)
    torch.cuda.synchronize()
    touched = (t["obs_buf"].abs().sum(1) > 0).cpu().numpy()
    assert touched.tolist() == [i in (2, 9) for i in range(16)]


def test_terrain_map_indices_bit_exact_vs_reference_golden(golden):
    """A10 at scale on the MI355X (emloco_task_get_heights: the device functions the fused post-physics kernel evaluates):
    int64 map indices and heights array_equal to the reference's over 256 poses x (1024 + 9) probes, and the fused kernel's
    height observations equal to the oracle's on the same poses."""
    import ctypes as C
    import oracle
    from helpers import terrain_index_map
    from emloco_amd import _lib as L
    from emloco_amd.sim import current_stream_handle
    g = golden("terrain_index")
    lib = L.require_device()
    dev = torch.device("cuda", 0)
    E = 256
    hf = torch.from_numpy(terrain_index_map()).to(dev)
    st = current_stream_handle(dev)
    pose = torch.from_numpy(g["head_pose"]).to(dev).contiguous()
    h = torch.zeros(E, 1024, device=dev)
    px, py = torch.zeros(E, 1024, dtype=torch.int64, device=dev), torch.zeros(E, 1024, dtype=torch.int64, device=dev)
    L.check(lib.emloco_task_get_heights(hf.data_ptr(), 1080, 1080, 0.1, 0.005, pose.data_ptr(), E, 1, h.data_ptr(), px.data_ptr(),
                                        py.data_ptr(), st), "emloco_task_get_heights")
    root7 = torch.from_numpy(np.ascontiguousarray(g["root_states"][:, :7])).to(dev)
    c = torch.zeros(E, 9, device=dev)
    cx, cy = torch.zeros(E, 9, dtype=torch.int64, device=dev), torch.zeros(E, 9, dtype=torch.int64, device=dev)
    L.check(lib.emloco_task_get_heights(hf.data_ptr(), 1080, 1080, 0.1, 0.005, root7.data_ptr(), E, 0, c.data_ptr(), cx.data_ptr(),
                                        cy.data_ptr(), st), "emloco_task_get_heights")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(px.cpu().numpy(), g["px"])
    np.testing.assert_array_equal(py.cpu().numpy(), g["py"])
    np.testing.assert_array_equal(np.round(h.cpu().numpy() / 0.005).astype(np.int16), g["heights_raw"])
    np.testing.assert_array_equal(cx.cpu().numpy(), g["cpx"])
    np.testing.assert_array_equal(cy.cpu().numpy(), g["cpy"])
    np.testing.assert_array_equal(np.round(c.cpu().numpy() / 0.005).astype(np.int16), g["center_raw"])
    hh, opx, opy = oracle.get_heights(g["head_pose"], terrain_index_map(), return_index=True)
    np.testing.assert_array_equal(h.cpu().numpy(), hh)


def _height_obs_scene(E, dev, hf, head_pose, root_states):
    """Fused post-physics kernel on poses taken from a terrain fixture: body 0 = root, body 13 = head."""
    from emloco_amd import _lib as L
    from emloco_amd.post_physics import PostPhysics
    T = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    rb = np.zeros((E, 24, 13), np.float32)
    rb[:, :, 6] = 1
    rb[:, 0, :7] = root_states[:, :7]
    rb[:, 13, :7] = head_pose
    t = dict(rb_state=T(rb), dof_state=torch.zeros(E, 69, 2, device=dev), dof_force=torch.zeros(E, 69, device=dev),
             contact_force=torch.zeros(E, 24, 3, device=dev), betas=torch.zeros(E, 17, device=dev),
             traj_verts=torch.zeros(E, 101, 3, device=dev), progress_buf=torch.zeros(E, dtype=torch.int64, device=dev),
             reset_buf=torch.zeros(E, dtype=torch.int64, device=dev), terminate_buf=torch.zeros(E, dtype=torch.int64, device=dev),
             obs_buf=torch.zeros(E, L.OBS, device=dev), flip_obs_buf=torch.zeros(E, L.OBS, device=dev), rew_buf=torch.zeros(E, device=dev),
             reward_raw=torch.zeros(E, 2, device=dev), amp_obs_buf=torch.zeros(E, L.AMP_STEPS, L.AMP_ROW, device=dev))
    pp = PostPhysics(dev)
    bufs = pp.make_bufs(n_env=E, heightfield=T(hf, torch.int16), dt=1 / 30.0, traj_dur=5.656, sample_dt=0.4, hscale=0.1, vscale=0.005,
                        power_coef=0.0005, fail_dist=4.0, max_episode_length=168.0, **t)
    pp.run(bufs, L.POST_OBS)
    torch.cuda.synchronize()
    return t["obs_buf"].cpu().numpy(), t["flip_obs_buf"].cpu().numpy()


def test_fused_height_observations_equal_the_reference_golden(golden):
    """A10 inside the fused kernel: the 1024 height observations clip(mean(centre) - h, -3, 3) * 5
    (humanoid_pedestrain_terrain.py:427-437) equal the reference's element for element -- the 16-pose fixture with its stored
    observations (array_equal against torch's own output) and the 256-pose index fixture (observations rebuilt from the
    reference's integer heights)."""
    import oracle
    from helpers import terrain_index_map
    dev = torch.device("cuda", 0)
    gt = golden("terrain_heights")
    obs, fobs = _height_obs_scene(16, dev, gt["heightfield"], gt["head_pose"], gt["root_states"])
    np.testing.assert_array_equal(obs[:, 398:], gt["height_obs"])
    g = golden("terrain_index")
    obs, fobs = _height_obs_scene(256, dev, terrain_index_map(), g["head_pose"], g["root_states"])
    exp = oracle.height_obs(g["center_raw"].astype(np.float32) * np.float32(0.005), g["heights_raw"].astype(np.float32) * np.float32(0.005))
    np.testing.assert_array_equal(obs[:, 398:], exp)
    np.testing.assert_array_equal(fobs[:, 398:].reshape(256, 32, 32), exp.reshape(256, 32, 32)[:, :, ::-1])
