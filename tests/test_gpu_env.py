"""GPU: the task / VecEnv mirror end to end (create -> reset -> step), checked against the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _make_env(num_envs=64, extra=()):
    from emloco_amd.run import create_rlgpu_env, fill_flags
    from emloco_amd.utils.config import get_args, load_cfg
    args = get_args(["--num_envs", str(num_envs), "--seed", "3", *extra])
    cfg, cfg_train, _ = load_cfg(args)
    fill_flags(args)
    return create_rlgpu_env(args, cfg, cfg_train)


def test_reset_then_step_matches_oracle_recompute():
    import oracle
    from helpers import oracle_sim
    env = _make_env(64, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel"])
    task = env.task
    E = 64
    assert task.num_obs == 1422 and task.num_actions == 69
    obs = env.reset(torch.arange(E, device=task.device))
    torch.cuda.synchronize()
    assert obs.shape == (E, 1422) and torch.isfinite(obs).all()
    assert (task.progress_buf == 0).all() and (task.reset_buf == 0).all()
    # reset obs == oracle self obs of the state the sim holds
    rb = task._rigid_body_state.view(E, 24, 13).cpu().numpy()
    betas = task.humanoid_betas.cpu().numpy()
    o = oracle.self_obs(rb[:, :, 0:3], rb[:, :, 3:7], rb[:, :, 7:10], rb[:, :, 10:13], betas)
    np.testing.assert_allclose(obs[:, :368].cpu().numpy(), o, rtol=1e-5, atol=2e-5)
    # lowest collision point of every reset humanoid sits 2 cm above the ground
    low = task._lowest_point(torch.arange(E, device=task.device)).cpu().numpy()
    np.testing.assert_allclose(low, 0.02, atol=2e-4)
    # LocoVal inputs captured at reset (vec_task_wrappers.py:50-66)
    wp = env.get_waypoint_traj()
    assert wp.shape == (E, 15, 3) and torch.allclose(wp[:, 0], torch.zeros(E, 3, device=wp.device))
    assert env.get_init_pose().shape == (E, 24, 3) and env.get_init_vel().shape == (E, 2)

    # one env.step vs the oracle driven with the same state, targets and models
    root0 = task._root_states.cpu().numpy().copy()
    dof0 = task._dof_state.view(E, 69, 2).cpu().numpy().copy()
    actions = torch.randn(E, 69, device=task.device) * 0.05
    models = [e.actors[0].asset.model for e in task.sim.envs]
    for m, e in zip(models, task.sim.envs):   # per-env PD gains as set through set_actor_dof_properties
        pass
    obs2, rew, done, info = env.step(actions)
    torch.cuda.synchronize()
    tgt = task._pd_targets.cpu().numpy()
    zero = task._pd_zero_mask.cpu().numpy()
    exp_t = oracle.pd_targets(actions.cpu().numpy(), task._pd_action_offset.cpu().numpy(), task._pd_action_scale.cpu().numpy(), zero)
    np.testing.assert_array_equal(tgt, exp_t)
    import copy
    ms = []
    for e in task.sim.envs:
        m = copy.copy(e.actors[0].asset.model)
        m.kp = e.actors[0].dof_props["stiffness"].astype(np.float64)
        m.kd = e.actors[0].dof_props["damping"].astype(np.float64)
        ms.append(m)
    assert task.sim.native._sc["pairs"].shape[0] == 286 and task.sim.native._sc["seg_body"].shape[0] == 26    # has_self_collision: True (pacer.yaml) reached the simulator: 26 segments (ankle boxes as two capsules)
    osim = oracle_sim(ms, root0, dof0, tgt, self_collision=task.sim.native._sc, n_sub=4)
    osim.step(1)
    np.testing.assert_array_equal(task._rigid_body_state.view(E, 24, 13).cpu().numpy(), osim.rb_state)   # bit-exact physics
    assert (task.progress_buf == 1).all()
    assert set(info) >= {"terminate", "reward_raw", "flip_obs", "obs", "amp_obs"}
    assert info["amp_obs"].shape == (E, 15 * 206)
    verts = task._traj_gen._verts.cpu().numpy()
    tar = oracle.traj_calc_pos(verts, task.progress_buf.cpu().numpy(), task.dt, task._traj_gen.get_traj_duration())
    r, raw = oracle.reward(osim.rb_state[:, 0, :3], tar, osim.dof_force, osim.dof_state[:, :, 1])
    np.testing.assert_allclose(rew.cpu().numpy(), r, rtol=1e-5, atol=1e-6)
    rs, tm = oracle.reset(task.progress_buf.cpu().numpy(), osim.contact_force, osim.rb_state[:, :, :3], tar)
    np.testing.assert_array_equal(done.cpu().numpy(), rs)
    np.testing.assert_array_equal(info["terminate"].cpu().numpy(), tm)


def test_rollout_with_resets_runs_an_episode():
    env = _make_env(256, ["--random_heading", "--init_heading"])
    task = env.task
    E = 256
    env.reset(torch.arange(E, device=task.device))
    n_done = 0
    for k in range(170):
        done_ids = task.reset_buf.nonzero(as_tuple=False).flatten()
        n_done += len(done_ids)
        if len(done_ids):
            env.reset(done_ids)
        obs, rew, done, info = env.step(torch.randn(E, 69, device=task.device) * 0.055)
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert n_done >= E          # every env finished at least one episode (168-step cap or early termination)
    assert int(task.progress_buf.max()) <= 168


def test_locoval_rollout_fits_value_function():
    """A17/A18: discounted-return bookkeeping + LocoVal fit through the fused LocoVal kernels."""
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    from emloco_amd.run import RLGPUEnv
    env = RLGPUEnv(_make_env(128, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel",
                                   "--input_init_pose", "--input_init_vel"]))
    agent = LocoValRollout(env, horizon_length=16)
    w0 = agent.valuenet._network.fc1.weight.detach().clone()
    for _ in range(6):
        agent.play_steps()
    torch.cuda.synchronize()
    assert agent.vnet_fits > 0 and np.isfinite(agent.vnet_loss)
    assert not torch.equal(w0, agent.valuenet._network.fc1.weight)      # the optimiser really stepped
    assert agent.frames == 6 * 16 * 128
    # checkpoint in the reference's layout and naming (common_agent.py:252,264): a plain state_dict
    import os, tempfile
    with tempfile.TemporaryDirectory() as d:
        p1 = agent.save(os.path.join(d, "Humanoid"))
        p2 = agent.save(os.path.join(d, "Humanoid"), 25000)
        assert p1.endswith("Humanoid_valuenet.pth") and p2.endswith("Humanoid_valuenet_00025000.pth")
        sd = torch.load(p2)
        assert list(sd) == ["_network.fc1.weight", "_network.fc1.bias", "_network.fc2.weight", "_network.fc2.bias", "_network.fc3.weight", "_network.fc3.bias"]
        agent2 = LocoValRollout(env, horizon_length=16)
        agent2.restore(p2)
        assert torch.equal(agent2.valuenet._network.fc1.weight, agent.valuenet._network.fc1.weight)


def test_rollout_with_frozen_policy_and_disc_reward(tmp_path):
    """Config 3: the frozen policy (A19) acts, the AMP discriminator supplies the style reward (A17), LocoVal is fitted (A18);
    the networks load from an rl_games-layout checkpoint."""
    from emloco_amd.learning.amp_policy import AMPPolicyBundle
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    from emloco_amd.run import RLGPUEnv
    env = RLGPUEnv(_make_env(128, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel",
                                   "--input_init_pose", "--input_init_vel"]))
    task = env.env.task
    b0 = AMPPolicyBundle(task, seed=1)
    with torch.no_grad():                                   # make the checkpoint differ from a fresh initialisation
        b0.a2c_network.mu.bias.add_(0.05)
        b0.running_mean_std.running_mean.add_(0.1)
    ck = {"model": {"a2c_network." + k: v.cpu() for k, v in b0.a2c_network.state_dict().items()},
          "running_mean_std": b0.running_mean_std.state_dict(), "amp_input_mean_std": b0.amp_input_mean_std.state_dict()}
    path = str(tmp_path / "Humanoid.pth")
    torch.save(ck, path)
    bundle = AMPPolicyBundle(task, checkpoint=path, deterministic=True)
    obs = task.obs_buf.clone()
    a = bundle.policy(obs).clone()
    assert torch.equal(a, AMPPolicyBundle(task, checkpoint=path, deterministic=True).policy(obs))    # checkpoint round trip
    assert a.shape == (128, 69) and float(a.abs().max()) <= 1.0
    r = bundle.disc_reward(task._amp_obs_buf)
    assert r.shape == (128,) and torch.isfinite(r).all() and float(r.min()) >= 0.0
    # the packed runner (FrozenDisc: one normalise launch + three GEMMs on preallocated buffers) against the network's modules
    assert torch.allclose(r, bundle.disc_reward_modules(task._amp_obs_buf), rtol=1e-5, atol=1e-6)
    assert bundle.eval_critic(obs).shape == (128, 1)
    agent = LocoValRollout(env, horizon_length=16, policy=bundle.policy, disc_reward=bundle.disc_reward,
                           inversion_penalty_scale=0.3)
    for _ in range(4):
        agent.play_steps()
    torch.cuda.synchronize()
    assert agent.frames == 4 * 16 * 128 and np.isfinite(agent.vnet_loss)


def test_reset_done_equals_reset_of_nonzero():
    """The sync-free `reset_done()` (device-compacted id list) leaves the same state as `reset(reset_buf.nonzero())` with
    the same random rows -- two identically seeded envs, one per path, compared bit for bit after several resets."""
    from emloco_amd import _lib as L
    args = ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel"]
    envs = [_make_env(96, args), _make_env(96, args)]
    dev = envs[0].task.device
    g = torch.Generator(device=dev)
    for k in range(40):
        g.manual_seed(100 + k)
        act = torch.randn(96, 69, device=dev, generator=g) * 0.3
        g.manual_seed(500 + k)
        rnd = torch.rand(96, L.RESET_RND, device=dev, generator=g)
        ta, tb = envs[0].task, envs[1].task
        if k == 0:
            ta.reset_buf[:] = 1
            tb.reset_buf[:] = 1
        if k % 7 == 3:
            ta.reset_buf[5:40:3] = 1                         # force a batch of resets besides the natural terminations
            tb.reset_buf[5:40:3] = 1
        done = ta.reset_buf.nonzero(as_tuple=False).flatten()
        if done.numel():
            ta._fused_reset_envs(done, rnd=rnd[:done.numel()].contiguous())
        tb.reset_done(rnd=rnd)
        envs[0].step(act)
        envs[1].step(act)
    torch.cuda.synchronize()
    ta, tb = envs[0].task, envs[1].task
    for name in ("_root_states", "_dof_state", "obs_buf", "_amp_obs_buf", "progress_buf", "reset_buf", "rew_buf", "waypoint_traj",
                 "init_pose", "init_vel", "_motion_start_times", "_sampled_motion_ids"):
        a, b = getattr(ta, name), getattr(tb, name)
        assert torch.equal(a, b), name
    assert torch.equal(ta._traj_gen._traj_verts, tb._traj_gen._traj_verts) if hasattr(ta._traj_gen, "_traj_verts") else True


def test_seeded_reset_done_draws_uniform_rows_on_the_device():
    """reset_done() without explicit rows: the rows come from the stateless device generator, only for the finished envs;
    same seed and call count -> same bytes, the values are uniform on [0, 1), and the resets it produces are as varied as
    torch.rand's (spawn headings over the whole circle, motion ids over the library)."""
    from emloco_amd import _lib as L
    args = ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel"]
    torch.manual_seed(7)
    a = _make_env(256, args)
    torch.manual_seed(7)
    b = _make_env(256, args)
    for env in (a, b):
        env.task.reset_buf[:] = 1
        env.task.reset_done()
    torch.cuda.synchronize()
    ta, tb = a.task, b.task
    assert ta._rnd_seed0 == tb._rnd_seed0 and torch.equal(ta._rnd_ws, tb._rnd_ws)
    assert torch.equal(ta._root_states, tb._root_states) and torch.equal(ta._sampled_motion_ids, tb._sampled_motion_ids)
    u = ta._rnd_ws
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 0.005 and abs(float(u.var()) - 1.0 / 12.0) < 0.003
    assert abs(float(torch.corrcoef(torch.stack([u[:, :-1].flatten(), u[:, 1:].flatten()]))[0, 1])) < 0.01     # neighbours in a row
    assert abs(float(torch.corrcoef(torch.stack([u[:-1].flatten(), u[1:].flatten()]))[0, 1])) < 0.01           # neighbouring rows
    yaw = 2 * torch.atan2(ta._root_states[:, 5], ta._root_states[:, 6])
    assert float(yaw.min()) < -2.0 and float(yaw.max()) > 2.0 and ta._sampled_motion_ids.unique().numel() > 20
    # next call: a different seed, and only the finished envs' rows are touched
    before = ta._rnd_ws.clone()
    ta.reset_buf[:] = 0
    ta.reset_buf[10:20] = 1
    ta.reset_done()
    torch.cuda.synchronize()
    assert not torch.equal(ta._rnd_ws[:10], before[:10]) and torch.equal(ta._rnd_ws[10:], before[10:])
    assert (ta.progress_buf[10:20] == 0).all()


def test_fused_reset_matches_host_mirror():
    """The three-kernel device reset against the host-side torch mirror of the reference's reset path."""
    from emloco_amd import _lib as L
    from emloco_amd.utils.flags import flags
    env = _make_env(64, ["--init_heading", "--heading_inversion", "--adjust_root_vel"])
    task = env.task
    E = 64
    dev = task.device
    ids = torch.arange(E, device=dev)
    torch.manual_seed(0)
    rnd = torch.rand(E, L.RESET_RND, device=dev)
    task.progress_buf[:] = 55
    task._fused_reset_envs(ids, rnd=rnd)
    torch.cuda.synchronize()
    assert (task.progress_buf == 0).all() and (task.reset_buf == 0).all() and (task._terminate_buf == 0).all()
    # motion sample: same (motion id, time) through the host motion library
    mid, mt = task._reset_motion_ids, task._reset_motion_times
    ref = task._motion_lib.get_motion_state_smpl(mid, mt)
    np.testing.assert_allclose(task._dof_pos.cpu().numpy(), ref["dof_pos"].cpu().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(task._dof_vel.cpu().numpy(), ref["dof_vel"].cpu().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(task._humanoid_root_states[:, 3:7].cpu().numpy(), ref["root_rot"].cpu().numpy(), rtol=1e-4, atol=2e-5)
    assert torch.allclose(task._humanoid_root_states[:, 0], torch.full((E,), 50.0, device=dev))     # flags.fixed placement (:588)
    low = task._lowest_point(ids).cpu().numpy()
    np.testing.assert_allclose(low, 0.02, atol=2e-4)
    # trajectory: host TrajGenerator with the same draws
    from emloco_amd.env.util.traj_generator import TrajGenerator
    tg = TrajGenerator(E, task.max_episode_length * task.dt, 101, dev, 2.0, task._speed_min, task._speed_max, task._accel_max,
                       task._sharp_turn_prob, None, hybridInitProb=task._hybrid_init_prob, flags=flags)
    draws = dict(r_dtheta=rnd[:, L.RND_DTHETA:L.RND_DTHETA + 100], r_dtheta_sharp=rnd[:, L.RND_SHARP:L.RND_SHARP + 100],
                 bern_sharp=(rnd[:, L.RND_BERN:L.RND_BERN + 100] < task._sharp_turn_prob).float(), r_heading=rnd[:, L.RND_HEADING],
                 r_dspeed=rnd[:, L.RND_DSPEED:L.RND_DSPEED + 100], r_speed0=rnd[:, L.RND_SPEED0], r_inversion=rnd[:, L.RND_INVERSION])
    root = task._humanoid_root_states
    tg.reset(ids, root[:, 0:3].clone(), root[:, 7:10].clone(), draws={k: v.clone() for k, v in draws.items()})
    np.testing.assert_allclose(task._traj_gen._verts.cpu().numpy(), tg._verts.cpu().numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_array_equal(task.inverted.cpu().numpy(), tg.inverted.cpu().numpy())            # bit-exact mask
    # LocoVal inputs and AMP history
    np.testing.assert_allclose(task.waypoint_traj.cpu().numpy(), task._fetch_traj_samples(ids).cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(task.init_pose.cpu().numpy(), task._rigid_body_pos.cpu().numpy(), rtol=0, atol=0)
    n = 14
    mids = torch.tile(mid.unsqueeze(-1), [1, n]).view(-1)
    mts = (mt.unsqueeze(-1) - task.dt * (torch.arange(n, device=dev) + 1)).view(-1)
    rows = task._amp_rows_from_motion(mids, mts, task.humanoid_betas.unsqueeze(1).expand(-1, n, -1).reshape(-1, 17)).view(E, n, 206)
    np.testing.assert_allclose(task._amp_obs_buf[:, 1:].cpu().numpy(), rows.cpu().numpy(), rtol=1e-4, atol=2e-5)
    assert torch.isfinite(task.obs_buf).all()


def test_amp_agent_train_epoch_on_the_rollout():
    """configs[1] end to end at toy size: PPO + AMP training epoch (rollout with the policy in the loop, GAE, disc reward,
    minibatch updates on the MFMA GEMMs) runs, changes the policy and keeps everything finite; checkpoint round trip."""
    import yaml
    from emloco_amd.learning.amp_agent import AMPAgent
    from emloco_amd.learning.amp_policy import DEFAULT_CFG
    from emloco_amd.run import RLGPUEnv
    env = RLGPUEnv(_make_env(64, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel"]))
    cfg = yaml.safe_load(open(DEFAULT_CFG))
    cfg["params"]["network"]["mlp"]["units"] = [256, 128]
    cfg["params"]["network"]["disc"]["units"] = [128, 64]
    cfg["params"]["config"].update(horizon_length=8, minibatch_size=128, amp_minibatch_size=128, amp_batch_size=64,
                                   amp_obs_demo_buffer_size=512, amp_replay_buffer_size=512, mini_epochs=2)
    agent = AMPAgent(env, cfg)
    w0 = agent.a2c_network.mu.weight.detach().clone()
    d0 = agent.a2c_network._disc_logits.weight.detach().clone()
    for _ in range(2):
        info = agent.train_epoch()
    assert all(np.isfinite(v) for k, v in info.items() if isinstance(v, float)), info
    assert not torch.equal(w0, agent.a2c_network.mu.weight) and not torch.equal(d0, agent.a2c_network._disc_logits.weight)
    assert agent.frame == 2 * 8 * 64 and info["fps_step"] > 0
    assert float(agent.running_mean_std.count) > 1.0            # the observation statistics were updated in train mode


def test_ppo_loss_heads_give_the_torch_losses_and_gradients_inside_the_agent():
    """The PPO learner's fused loss heads (learning/ppo_heads.py, csrc/ppo_kernels.hip) against the torch expressions they replace, INSIDE
    the agent: `compute_loss` on one real minibatch of a rollout with the heads on and off -- every reported scalar, the total loss and the
    gradient of every parameter (through the actor, critic and discriminator networks) agree to the GEMMs' tolerance; the KL the
    heads report equals policy_kl."""
    import yaml
    from emloco_amd.learning.amp_agent import AMPAgent, policy_kl
    from emloco_amd.learning.amp_policy import DEFAULT_CFG
    from emloco_amd.run import RLGPUEnv
    env = RLGPUEnv(_make_env(64, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel"]))
    cfg = yaml.safe_load(open(DEFAULT_CFG))
    cfg["params"]["network"]["mlp"]["units"] = [256, 128]
    cfg["params"]["network"]["disc"]["units"] = [128, 64]
    cfg["params"]["config"].update(horizon_length=8, minibatch_size=128, amp_minibatch_size=128, amp_batch_size=64,
                                   amp_obs_demo_buffer_size=512, amp_replay_buffer_size=512, mini_epochs=1)
    os.environ["EMLOCO_PPO_GRAPH"] = "0"
    try:
        agent = AMPAgent(env, cfg)
    finally:
        os.environ.pop("EMLOCO_PPO_GRAPH")
    assert agent._fused_heads and agent._flat_adam
    agent.train_epoch()                                              # fills the dataset, moves the policy off its initial point
    agent.set_eval()                                                 # (frozen normalisers: two evaluations of one minibatch must see the same inputs)
    d = {k: (v[:128] if v is not None else None) for k, v in agent.dataset.items()}
    d["advantages"] = d["advantages"] * 3.0                          # (spread the ratios' effect; clipping happens on both sides)
    from emloco_amd.learning.amp_agent import amp_dropout_mask
    masks = None
    if agent._amp_dropout:
        torch.manual_seed(0)
        masks = amp_dropout_mask(agent._amp_minibatch_size, agent.task._num_amp_obs_steps, d["amp_obs"].shape[1] // agent.task._num_amp_obs_steps,
                                 device=d["amp_obs"].device)
    res = {}
    for heads in (True, False):
        agent._fused_heads = heads
        agent.bucket.zero()
        loss, info, mu, sigma = agent.compute_loss(d, dropout_masks=masks)
        loss.backward()
        kl = agent._head_kl if heads else policy_kl(mu, sigma, d["mu"], d["sigma"])
        assert (agent._head_kl is not None) == heads
        res[heads] = (loss.detach().clone(), {k: v.detach().clone() for k, v in info.items()}, agent.bucket.grads.clone(), kl.detach().clone())
    agent._fused_heads = True
    (l1, i1, g1, k1), (l0, i0, g0, k0) = res[True], res[False]
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    assert abs(float(l1) - float(l0)) <= 2e-5 * abs(float(l0)) + 1e-6, (float(l1), float(l0))
    for k in i0:
        assert abs(float(i1[k]) - float(i0[k])) <= 1e-4 * abs(float(i0[k])) + 1e-6, (k, float(i1[k]), float(i0[k]))
    assert abs(float(k1) - float(k0)) <= 1e-4 * abs(float(k0)) + 1e-7
    assert (g1 - g0).abs().max().item() <= 2e-4 * g0.abs().max().item(), ((g1 - g0).abs().max().item(), g0.abs().max().item())


def test_rollout_on_shaped_terrain_collides_with_the_heightfield():
    """terrainProportions with slopes / stairs / obstacles (curriculum layout; no stepping stones: like the reference, a spawn
    next to their 10 m deep gaps averages the gap into its ground height and starts below the stones): the task builds the
    height-field mesh, the sim collides with it, resets place the humanoids on the local ground, the 32 x 32 height
    observations see the relief, and nobody sinks through the terrain during a rollout."""
    from emloco_amd.run import create_rlgpu_env, fill_flags
    from emloco_amd.utils.config import get_args, load_cfg
    E = 64
    args = get_args(["--num_envs", str(E), "--seed", "3"])
    cfg, cfg_train, _ = load_cfg(args)
    fill_flags(args)
    from emloco_amd.utils.flags import flags
    flags.fixed = False                                               # run.py pins every spawn to (50, 55); spread them over the map here
    cfg["env"]["terrain"].update(terrainProportions=[0.3, 0.0, 0.25, 0.25, 0.1, 0.0, 0.0, 0.1], numLevels=2, numTerrains=6,
                                 mapLength=8., mapWidth=8., curriculum=True)
    np.random.seed(3)
    try:
        _shaped_terrain_body(create_rlgpu_env(args, cfg, cfg_train), E)
    finally:
        flags.fixed = True


def _shaped_terrain_body(env, E):
    task = env.task
    terr = task.terrain
    assert not terr.is_flat and task.sim.heightfield is not None
    assert task.sim.heightfield["samples"].shape == terr.height_field_raw.shape
    # round 5: the sim collides with the slope-corrected mesh the task built (vertical risers), its vertex moves read back from the mesh
    mv = task.sim.heightfield
    assert mv.get("move_x") is not None and (mv["move_x"] != 0).any() and (mv["move_y"] != 0).any()
    vx = terr.vertices.reshape(terr.tot_rows, terr.tot_cols, 3)[:, :, 0]
    np.testing.assert_allclose(vx, (np.arange(terr.tot_rows)[:, None] + mv["move_x"]) * terr.horizontal_scale, atol=1e-4)
    ids = torch.arange(E, device=task.device)
    obs = env.reset(ids)
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all()
    hobs = obs[:, 368 + 30:368 + 30 + 1024]
    assert hobs.std(dim=1).max().item() > 0.05                       # some humanoids look at relief
    hs = terr.heightsamples.to(task.device).float() * terr.vertical_scale

    def ground_under(p):                                              # bilinear-free lookup: min of the cell's diagonal corners
        px = (p[..., 0] / terr.horizontal_scale).long().clamp(0, hs.shape[0] - 2)
        py = (p[..., 1] / terr.horizontal_scale).long().clamp(0, hs.shape[1] - 2)
        return torch.minimum(hs[px, py], hs[px + 1, py + 1])

    root = task._root_states
    assert ((root[:, 2] - ground_under(root)) > 0.5).all()            # spawned standing on the local ground, not at z = 0.9
    assert (ground_under(root).abs() > 0.05).any()                    # ... and some of them on raised / lowered ground
    low_gap = []
    for k in range(45):
        obs, rew, done, info = env.step(torch.randn(E, 69, device=task.device) * 0.1)
        rb = task._rigid_body_state.view(E, 24, 13)
        low_gap.append((rb[..., 2] - ground_under(rb)).min().item())
        env.reset_done()
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(task._rigid_body_state).all()
    assert min(low_gap) > -0.25                                       # body origins stay above the terrain (step edges: one riser of slack)
    cf = task._contact_forces.view(E, 24, 3)
    assert cf[..., 2].sum(1).max().item() > 300                       # the terrain carries them


@pytest.mark.parametrize("name", ["traj_reset_plain", "traj_reset_heading", "traj_reset_real1", "traj_reset_real2", "traj_reset_real2_noadj"])
def test_traj_reset_on_device_matches_reference_golden(golden, name):
    """TrajGenerator.reset on the MI355X (emloco_task_traj_reset) on the reference's own draws: real-path branch, one / two
    datasets, with / without adjust_root_vel (traj_generator.py:60-237). Vertices within 1e-4 m, inversion mask bit-exact."""
    from helpers import TRAJ_CASES, traj_real_pick, traj_rnd_rows
    from emloco_amd.env.util.traj_generator import TrajGenerator
    from emloco_amd.utils.flags import Flags
    g = golden(name)
    base = dict(real_path=False, jta_path=False, jrdb_path=False, pred_path=False, fixed_path=False, slow=False,
                adjust_root_vel=False, init_heading=False, heading_inversion=False, add_noise=False, vru=False)
    base.update(TRAJ_CASES[name])
    data = None
    if "real_table" in g:
        n_jta = int(g["n_jta"])
        tables = [g["real_table"][:n_jta]] + ([g["real_table"][n_jta:]] if g["real_table"].shape[0] > n_jta else [])
        data = [{i: {"pose": None, "traj": t[i]} for i in range(len(t))} for t in tables]
    dev = torch.device("cuda", 0)
    tg = TrajGenerator(16, 168 * (2 / 60.0), 101, dev, 2.0, 0.0005, 3.0, 2.0, 0.02, None, hybridInitProb=0.5, flags=Flags(base),
                       traj_data=data)
    tg.inverted[:] = True
    tg.reset_on_device(torch.arange(16, device=dev), torch.from_numpy(g["init_pos"]).to(dev), torch.from_numpy(g["root_vel"]).to(dev),
                       rnd=torch.from_numpy(traj_rnd_rows(g)).to(dev),
                       real_pick=torch.from_numpy(traj_real_pick(g)) if data is not None else None)
    torch.cuda.synchronize()
    err = np.abs(tg._verts.cpu().numpy() - g["verts"]).max()
    assert err < 1e-4, err
    if "inverted" in g:
        np.testing.assert_array_equal(tg.show_inverted().long().cpu().numpy(), g["inverted"])


def _synthetic_paths(n, seed, nv=101):
    r = np.random.RandomState(seed)
    out = {}
    for i in range(n):
        th = r.uniform(-np.pi, np.pi) + r.uniform(-0.4, 0.4) * np.arange(nv) * 0.056
        xy = np.cumsum(np.stack([np.cos(th), np.sin(th)], -1) * r.uniform(0.02, 0.14), 0) + r.uniform(-20, 20, 2)
        out[i] = {"pose": None, "traj": np.concatenate([xy, np.full((nv, 1), 1e-3 * i)], -1)}
    return out


def test_fused_reset_real_path_matches_host_mirror():
    """configs[1]'s reset: the RESET_REAL_PATH branch of the fused device reset (reset_kernels.hip: reset_trajectory) inside
    the full three-launch reset, against the host mirror that tests/test_host_logic.py pins to the reference
    (traj_generator.py:120-160): same rows (keyed permutation restated on the host), vertices <= 1e-4, masks bit-exact,
    rows distinct within the call."""
    from emloco_amd import _lib as L
    from emloco_amd.run import create_rlgpu_env, fill_flags
    from emloco_amd.utils.config import get_args, load_cfg
    from emloco_amd.utils.flags import flags
    from emloco_amd.env.util.traj_generator import TrajGenerator
    E = 64
    args = get_args(["--num_envs", str(E), "--seed", "3", "--random_heading", "--init_heading", "--heading_inversion",
                     "--adjust_root_vel", "--real_path", "JTA+JRDB"])
    cfg, cfg_train, _ = load_cfg(args)
    data = [_synthetic_paths(50, 1), _synthetic_paths(37, 2, nv=104)]
    cfg["env"]["traj_data"] = data
    fill_flags(args)
    try:
        env = create_rlgpu_env(args, cfg, cfg_train)
        task = env.task
        dev = task.device
        ids = torch.arange(E, device=dev)
        torch.manual_seed(0)
        rnd = torch.rand(E, L.RESET_RND, device=dev)
        if task._reset_bufs is None:
            task._reset_bufs = task._make_reset_bufs()
        key = 0xC0FFEE
        task._reset_bufs.real_pick_key = key
        task._fused_reset_envs(ids, rnd=rnd)
        torch.cuda.synchronize()
        real = (rnd[:, L.RND_REAL] > 0.5).cpu().numpy()
        assert 20 < real.sum() < 44
        rows = [L.real_pick_perm(i, 87, key) for i in range(E)]
        tg = TrajGenerator(E, task.max_episode_length * task.dt, 101, dev, 2.0, task._speed_min, task._speed_max, task._accel_max,
                           task._sharp_turn_prob, None, hybridInitProb=task._hybrid_init_prob, flags=flags, traj_data=data)
        draws = dict(r_dtheta=rnd[:, L.RND_DTHETA:L.RND_DTHETA + 100], r_dtheta_sharp=rnd[:, L.RND_SHARP:L.RND_SHARP + 100],
                     bern_sharp=(rnd[:, L.RND_BERN:L.RND_BERN + 100] < task._sharp_turn_prob).float(), r_heading=rnd[:, L.RND_HEADING],
                     r_dspeed=rnd[:, L.RND_DSPEED:L.RND_DSPEED + 100], r_speed0=rnd[:, L.RND_SPEED0], r_inversion=rnd[:, L.RND_INVERSION],
                     r_real=rnd[:, L.RND_REAL])
        draws = {k: v.clone() for k, v in draws.items()}
        draws["real_rids"] = [rows[i] for i in range(E) if real[i]]
        assert len(set(draws["real_rids"])) == len(draws["real_rids"])
        root = task._humanoid_root_states
        tg.reset(ids, root[:, 0:3].clone(), root[:, 7:10].clone(), draws=draws)
        got, exp = task._traj_gen._verts.cpu().numpy(), tg._verts.cpu().numpy()
        assert np.abs(got - exp).max() < 1e-4, np.abs(got - exp).max()
        # the real envs carry their row's z signature (1e-3 * row within its dataset), the others the flat polyline
        z = got[:, 5, 2]
        assert (z[~real] == 0).all()
        exp_z = np.array([1e-3 * (r if r < 50 else r - 50) for r in rows], np.float32)
        np.testing.assert_array_equal(z[real], exp_z[real])
        np.testing.assert_array_equal(task.inverted.cpu().numpy(), tg.inverted.cpu().numpy())
        np.testing.assert_allclose(task.waypoint_traj.cpu().numpy(), task._fetch_traj_samples(ids).cpu().numpy(), rtol=1e-5, atol=1e-5)
        # the seeded path (what reset_done / bench.py run): a fresh permutation per call, rows distinct within a call
        seen = []
        for _ in range(2):
            task.reset_buf[:] = 1
            task.reset_done()
            torch.cuda.synchronize()
            zz = task._traj_gen._verts[:, 5, 2].cpu().numpy()
            took = zz[zz != 0]
            assert 15 < took.size < 50
            sig = np.sort(np.round(took * 1e3).astype(int))
            # a dataset-local signature appears at most twice (once per dataset) when rows are distinct
            assert np.bincount(sig).max() <= 2
            seen.append(zz.copy())
        assert not np.array_equal(seen[0], seen[1])
    finally:
        fill_flags(get_args(["--num_envs", "1"]))


def test_bench_config_rollout_step_matches_oracle():
    """BASELINE configs[1] exactly as bench.py builds it (4096 envs, random_heading, init_heading + inversion,
    adjust_root_vel, JTA+JRDB real paths, self-collision): after a reset and 12 rollout steps with resets in between,
    one env.step of a slice of envs is recomputed by the oracle from the state, warm-start impulses and targets the
    simulator held -- body states bit-exact, reward / reset / terminate masks equal."""
    import copy
    import sys, os
    import oracle
    from helpers import oracle_sim
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from emloco_amd.utils.config import get_args
    from emloco_amd.run import fill_flags
    E = 4096
    try:
        env = bench.make_env(E, 0)
        task = env.task
        dev = task.device
        env.reset(torch.arange(E, device=dev))
        bench.stagger_episodes(env, steps=12, seed=5)
        g = torch.Generator(device=dev)
        g.manual_seed(3)
        for k in range(12):
            env.reset_done()
            env.step(torch.randn(E, 69, device=dev, generator=g) * float(np.exp(-2.9)))
        env.reset_done()
        torch.cuda.synchronize()
        assert int((task.progress_buf == 0).sum()) > 50              # some envs were just reset through the real-path branch
        sl = np.r_[0:24, 2040:2056, 4072:4096]                       # first / middle / last waves of the launch
        just_reset = np.nonzero((task.progress_buf == 0).cpu().numpy())[0][:8]
        sl = np.unique(np.concatenate([sl, just_reset]))
        root0 = task._root_states.cpu().numpy()[sl].copy()
        dof0 = task._dof_state.view(E, 69, 2).cpu().numpy()[sl].copy()
        lws0 = task.sim.native.warm_start.view(E, -1, 3).cpu().numpy()[sl].copy()
        prog0 = task.progress_buf.cpu().numpy()[sl].copy()
        act = torch.randn(E, 69, device=dev, generator=g) * float(np.exp(-2.9))
        obs, rew, done, info = env.step(act)
        torch.cuda.synchronize()
        ms = []
        for i in sl:
            e = task.sim.envs[int(i)]
            m = copy.copy(e.actors[0].asset.model)
            m.kp = e.actors[0].dof_props["stiffness"].astype(np.float64)
            m.kd = e.actors[0].dof_props["damping"].astype(np.float64)
            ms.append(m)
        sc = getattr(task.sim.native, "_sc", None)
        assert sc is not None                                         # has_self_collision: True (pacer.yaml:17)
        sc_sl = {k: (v[sl] if isinstance(v, np.ndarray) and v.shape[:1] == (E,) else v) for k, v in sc.items()}
        osim = oracle_sim(ms, root0, dof0, task._pd_targets.cpu().numpy()[sl], self_collision=sc_sl, n_sub=4)
        osim.lambda_ws[:] = lws0
        osim.step(1)
        rb = task._rigid_body_state.view(E, 24, 13).cpu().numpy()[sl]
        assert np.array_equal(rb, osim.rb_state), "bench-config env.step is not bit-exact against the oracle"
        assert np.array_equal(task.sim.native.warm_start.view(E, -1, 3).cpu().numpy()[sl], osim.lambda_ws)
        betas = task.humanoid_betas.cpu().numpy()[sl]
        o = oracle.self_obs(rb[:, :, 0:3], rb[:, :, 3:7], rb[:, :, 7:10], rb[:, :, 10:13], betas)
        np.testing.assert_allclose(obs[:, :368].cpu().numpy()[sl], o, rtol=1e-5, atol=2e-5)
        tar = oracle.traj_calc_pos(task._traj_gen._verts.cpu().numpy()[sl], prog0 + 1, task.dt, task._traj_gen.get_traj_duration())
        rs, tm = oracle.reset(prog0 + 1, osim.contact_force, osim.rb_state[:, :, :3], tar)
        np.testing.assert_array_equal(done.cpu().numpy()[sl], rs)
        np.testing.assert_array_equal(info["terminate"].cpu().numpy()[sl], tm)
        r, _ = oracle.reward(rb[:, 0, :3], tar, osim.dof_force, osim.dof_state[:, :, 1])
        np.testing.assert_allclose(rew.cpu().numpy()[sl], r, rtol=1e-5, atol=1e-6)
    finally:
        fill_flags(get_args(["--num_envs", "1"]))


def test_observations_on_a_side_stream_give_the_same_rollout():
    """task.overlap_obs: the step's launch is split -- progress / reward / reset flags on the caller's stream, observations and
    AMP rows of the live envs on a side stream (POST_SKIP_DONE) while the caller's stream resets the finished envs (whose
    terminal AMP rows the flags launch writes, POST_AMP_DONE_ONLY) -- and the loop calls wait_obs() before reading.  Two identically seeded envs, one per mode, forced and natural resets: every buffer
    bit-equal after every step."""
    from emloco_amd import _lib as L
    args = ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel"]
    envs = [_make_env(96, args), _make_env(96, args)]
    envs[1].task.overlap_obs = True
    dev = envs[0].task.device
    g = torch.Generator(device=dev)
    names = ("_root_states", "_dof_state", "obs_buf", "_flip_obs_buf", "_amp_obs_buf", "progress_buf", "reset_buf", "rew_buf",
             "reward_raw", "_terminate_buf", "waypoint_traj", "init_pose", "init_vel")
    for k in range(30):
        g.manual_seed(100 + k)
        act = torch.randn(96, 69, device=dev, generator=g) * 0.3
        g.manual_seed(500 + k)
        rnd = torch.rand(96, L.RESET_RND, device=dev, generator=g)
        for e in envs:
            t = e.task
            if k == 0:
                t.reset_buf[:] = 1
            if k % 7 == 3:
                t.reset_buf[5:40:3] = 1
            t.reset_done(rnd=rnd)
            t.wait_obs()
            e.step(act)
        envs[1].task.wait_obs()
        torch.cuda.synchronize()
        for name in names:
            a, b = getattr(envs[0].task, name), getattr(envs[1].task, name)
            # observation rows of finished envs are rebuilt by the next reset_done; their terminal AMP rows (what
            # infos['amp_obs'] hands the discriminator for the last step of an episode) must be there in both schedules
            live = envs[0].task.reset_buf == 0
            if name in ("obs_buf", "_flip_obs_buf"):
                assert torch.equal(a[live], b[live]), (k, name)
            else:
                assert torch.equal(a, b), (k, name)
    assert getattr(envs[1].task, "_obs_stream", None) is not None


@pytest.mark.gpu
@pytest.mark.parametrize("seeded,pool,amp_early", [(False, 64, False), (True, 64, False), (True, 8, False), (True, 0, False), (True, 64, True)])
def test_fused_chain_gives_the_same_rollout(seeded, pool, amp_early, monkeypatch):
    """task.fused_chain: post_physics_step launches only progress / reward / flags (+ the terminal AMP rows of the finished envs),
    reset_done() is two launches -- emloco_task_compact_done_order (compaction + flag snapshot + the next step's dispatch order) and
    emloco_task_reset_obs (reset chain of the finished envs, their AMP history, their observations AND the deferred observation /
    AMP pass of the envs that did not finish).  Two identically seeded envs, one per schedule, forced and natural resets, random
    rows supplied or drawn on the device from the same seeds: every buffer bit-equal after every reset_done, the terminal AMP rows
    after every step.  With device-drawn rows the fused launch also keeps a pool of pre-drawn episodes (EmlocoResetPool: `pool`
    entries drawn one call ahead, copied by the reset chain; 8 < the 12 forced resets: entries beyond the pool take the direct path;
    0: no pool) -- the separate kernels of the other env know nothing of it, the bytes must not differ."""
    from emloco_amd import _lib as L
    monkeypatch.setenv("EMLOCO_RESET_POOL", str(pool))
    args = ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel"]
    torch.manual_seed(11)
    envs = [_make_env(96, args), _make_env(96, args)]
    envs[1].task.fused_chain = True
    envs[1].task.fused_amp_early = amp_early        # AMP rows of every env in the flags launch (a discriminator reads them before the resets)
    for e in envs:
        e.task.sim.native.set_cost_order(True)
    dev = envs[0].task.device
    g = torch.Generator(device=dev)
    names = ("_root_states", "_dof_state", "_rigid_body_state", "obs_buf", "_flip_obs_buf", "_amp_obs_buf", "progress_buf", "reset_buf",
             "rew_buf", "reward_raw", "_terminate_buf", "waypoint_traj", "init_pose", "init_vel", "inverted")
    for k in range(30):
        g.manual_seed(100 + k)
        act = torch.randn(96, 69, device=dev, generator=g) * 0.3
        g.manual_seed(500 + k)
        rnd = None if seeded else torch.rand(96, L.RESET_RND, device=dev, generator=g)
        for e in envs:
            t = e.task
            if k == 0:
                t.reset_buf[:] = 1
            if k % 7 == 3:
                t.reset_buf[5:40:3] = 1
            t.reset_done(rnd=rnd)
        assert envs[1].task._obs_deferred == 0
        torch.cuda.synchronize()
        for name in names:
            a, b = getattr(envs[0].task, name), getattr(envs[1].task, name)
            assert torch.equal(a, b), (k, name, "after reset_done")
        for e in envs:
            e.step(act)
        assert (envs[1].task._obs_deferred != 0) and envs[0].task._obs_deferred == 0
        torch.cuda.synchronize()
        done = envs[0].task.reset_buf != 0
        assert torch.equal(envs[0].task.reset_buf, envs[1].task.reset_buf), k
        assert torch.equal(envs[0].task._amp_obs_buf[done], envs[1].task._amp_obs_buf[done]), (k, "terminal AMP rows")
        if amp_early:
            assert torch.equal(envs[0].task._amp_obs_buf, envs[1].task._amp_obs_buf), (k, "AMP observations of every env right after the step")
        for name in ("_rigid_body_state", "rew_buf", "progress_buf", "_terminate_buf"):
            assert torch.equal(getattr(envs[0].task, name), getattr(envs[1].task, name)), (k, name, "after step")
    assert int((envs[0].task.progress_buf == 0).sum()) < 96            # natural / forced resets happened, not only the first
    if seeded and pool > 0:                                            # the pool was in use: the last launch tagged its draw
        t = envs[1].task
        assert int((t._pool_tag[t._pool_flip] != 0).sum()) >= min(pool, 32)
    # a caller that never calls reset_done still gets its observations: wait_obs() launches the deferred pass
    for e in envs:
        e.task.wait_obs()
    torch.cuda.synchronize()
    live = envs[0].task.reset_buf == 0
    for name in ("obs_buf", "_flip_obs_buf", "_amp_obs_buf"):
        assert torch.equal(getattr(envs[0].task, name)[live], getattr(envs[1].task, name)[live]), name


@pytest.mark.gpu
@pytest.mark.parametrize("obs_stream", [True, False])
def test_reset_chain_on_a_second_stream_gives_the_same_rollout(obs_stream):
    """task.overlap_reset: `reset_done(); step(a)` issued as two chains -- the caller's stream steps the envs that did not
    finish (emloco_sim_step_subset with the flag snapshot), a second stream resets the finished ones, builds their
    observations and steps them over the compacted id list; step() joins before the post-physics launch.  Two identically
    seeded envs, one per mode (the second also with the observation side stream), forced and natural resets, the sim state
    and the warm-start impulses included: every buffer bit-equal after every step, and the reset envs' rows are valid after
    wait_reset().  160 envs: the id-list launch (128 workgroups) strides over the list when every env resets."""
    from emloco_amd import _lib as L
    args = ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel"]
    envs = [_make_env(160, args), _make_env(160, args)]
    envs[1].task.overlap_reset = True
    envs[1].task.overlap_obs = obs_stream            # the AMP history back-fill rides on the observation stream when there is one
    dev = envs[0].task.device
    g = torch.Generator(device=dev)
    names = ("_root_states", "_dof_state", "_rigid_body_state", "_contact_forces", "obs_buf", "_flip_obs_buf", "_amp_obs_buf",
             "progress_buf", "reset_buf", "rew_buf", "reward_raw", "_terminate_buf", "waypoint_traj", "init_pose", "init_vel")
    for k in range(40):
        g.manual_seed(100 + k)
        act = torch.randn(160, 69, device=dev, generator=g) * 0.3
        g.manual_seed(500 + k)
        rnd = torch.rand(160, L.RESET_RND, device=dev, generator=g)
        obs_after_reset = []
        for e in envs:
            t = e.task
            if k == 0:
                t.reset_buf[:] = 1
            if k % 7 == 3:
                t.reset_buf[5:40:3] = 1
            t.reset_done(rnd=rnd)
            t.wait_obs()
            if k % 5 == 0:                                          # a consumer between reset_done() and step()
                t.wait_reset()
                obs_after_reset.append(t.obs_buf.clone())
            e.step(act)
        envs[1].task.wait_obs()
        torch.cuda.synchronize()
        if obs_after_reset:
            assert torch.equal(obs_after_reset[0], obs_after_reset[1]), k
        for name in names:
            a, b = getattr(envs[0].task, name), getattr(envs[1].task, name)
            # observation rows of finished envs are rebuilt by the next reset_done; their terminal AMP rows (what
            # infos['amp_obs'] hands the discriminator for the last step of an episode) must be there in both schedules
            live = envs[0].task.reset_buf == 0
            if name in ("obs_buf", "_flip_obs_buf"):
                assert torch.equal(a[live], b[live]), (k, name)
            else:
                assert torch.equal(a, b), (k, name)
        assert torch.equal(envs[0].task.sim.native.warm_start, envs[1].task.sim.native.warm_start), k
        assert envs[0].task.sim.frame_count == envs[1].task.sim.frame_count
    assert getattr(envs[1].task, "_hp_stream", None) is not None


@pytest.mark.gpu
def test_locoval_loop_with_the_reset_chain_beside_the_step_fits_the_same_network():
    """The headline loop of bench.py in both schedules: LocoValRollout(overlap_reset=False / True) with a policy that does not
    read the observations, same seeds -- after 64 steps with natural resets the fitted LocoVal parameters, the episode count
    and the simulator state are bit-equal."""
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    from emloco_amd.run import RLGPUEnv
    E, outs = 256, []
    for ov in (False, True):
        env = RLGPUEnv(_make_env(E, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel",
                                     "--input_init_pose", "--input_init_vel"]))
        task = env.env.task
        g = torch.Generator(device=task.device)
        g.manual_seed(77)
        pool = torch.randn(8, E, 69, device=task.device, generator=g) * 0.3
        k = [0]

        def pol(obs):
            k[0] += 1
            return pool[k[0] % 8]
        pol.reads_obs = False
        torch.manual_seed(5)
        agent = LocoValRollout(env, horizon_length=8, policy=pol, overlap_reset=ov)
        for _ in range(8):
            agent.play_steps()
        n_fit = agent.fitted_episodes                      # waits for the fit stream
        task.wait_reset()
        torch.cuda.synchronize()
        assert task.overlap_reset == ov
        outs.append((torch.cat([p.detach().reshape(-1) for p in agent.valuenet.parameters()]).clone(), n_fit,
                     task._root_states.clone(), task._dof_state.clone(), task.progress_buf.clone(), task.rew_buf.clone()))
    assert outs[0][1] == outs[1][1] and outs[0][1] > 20
    for a, b in zip(outs[0], outs[1]):
        if torch.is_tensor(a):
            assert torch.equal(a, b)


@pytest.mark.gpu
def test_locoval_fit_in_groups_of_steps_gives_the_per_step_network_after_every_epoch(monkeypatch):
    """EMLOCO_FIT_EVERY (round 5): the fits of k rollout steps issued together behind one event, in step order, from a ring of 2 k
    staging sets -- against the per-step loop (k = 1), same seeds: after EVERY epoch (horizon 8, not a multiple of k = 3: the epoch's
    end flushes a short group, k = 4 and k = 8: whole groups) the LocoVal parameters, the fit counters and the loss are bit-equal, and
    so are the return accumulators and the simulator state at the end.  The learning rate moves every epoch (warm-up of 2 epochs)."""
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    from emloco_amd.run import RLGPUEnv
    E, outs = 256, []
    for every in ("1", "3", "4", "8"):
        monkeypatch.setenv("EMLOCO_FIT_EVERY", every)
        env = RLGPUEnv(_make_env(E, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel",
                                     "--input_init_pose", "--input_init_vel"]))
        task = env.env.task
        g = torch.Generator(device=task.device)
        g.manual_seed(77)
        pool = torch.randn(8, E, 69, device=task.device, generator=g) * 0.3
        k = [0]

        def pol(obs):
            k[0] += 1
            return pool[k[0] % 8]
        torch.manual_seed(5)
        agent = LocoValRollout(env, horizon_length=8, policy=pol, overlap_reset=False, warmup_epochs=3, max_epochs=40)
        assert agent._fit_every == int(every) and agent._nbuf >= 2 * int(every) or every == "1"
        per_epoch = []
        for _ in range(8):
            loss = agent.play_steps()                      # (reads the loss: flushes and waits for the fit stream)
            per_epoch.append((torch.cat([p.detach().reshape(-1) for p in agent.valuenet.parameters()]).clone(), agent.vnet_fits,
                              agent.fitted_episodes, float(loss), float(agent.vnet_optimizer.param_groups[0]["lr"])))
        assert not agent._pending
        torch.cuda.synchronize()
        a = agent.acc
        outs.append((per_epoch, a.current_rewards.clone(), a.current_lengths.clone(), a.current_combined_rewards.clone(),
                     a.discount_coefs.clone(), task._root_states.clone(), task.progress_buf.clone()))
        agent.detach()
    ref = outs[0]
    assert ref[0][-1][2] > 20 and len({e[4] for e in ref[0]}) > 2              # episodes were fitted, the rate moved
    for o in outs[1:]:
        for (w0, f0, n0, l0, r0), (w1, f1, n1, l1, r1) in zip(ref[0], o[0]):
            assert torch.equal(w0, w1) and (f0, n0, l0, r0) == (f1, n1, l1, r1)
        for x, y in zip(ref[1:], o[1:]):
            assert torch.equal(x, y)


@pytest.mark.gpu
def test_state_dict_of_the_value_net_mid_epoch_waits_for_the_pending_fits(monkeypatch):
    """With the fits issued in groups (EMLOCO_FIT_EVERY = 4) up to three steps' fits are not issued yet between two steps: a host-side
    reader that does not go through the loop -- `valuenet.state_dict()` for a checkpoint -- flushes them first (round 6: a state_dict
    pre-hook) and sees the per-step loop's network of that step."""
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    from emloco_amd.run import RLGPUEnv
    E, snaps = 256, []
    for every in ("1", "4"):
        monkeypatch.setenv("EMLOCO_FIT_EVERY", every)
        env = RLGPUEnv(_make_env(E, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel",
                                     "--input_init_pose", "--input_init_vel"]))
        task = env.env.task
        g = torch.Generator(device=task.device)
        g.manual_seed(78)
        pool = torch.randn(8, E, 69, device=task.device, generator=g) * 0.3
        k, holder, got = [0], [], []

        def pol(obs):
            k[0] += 1
            if k[0] in (11, 22):                            # mid-epoch (horizon 8): behind 2 resp. 5 steps of the epoch
                got.append((len(holder[0]._pending), {n: v.clone() for n, v in holder[0].valuenet.state_dict().items()}, len(holder[0]._pending)))
            return pool[k[0] % 8]
        torch.manual_seed(6)
        agent = LocoValRollout(env, horizon_length=8, policy=pol, overlap_reset=False, warmup_epochs=3, max_epochs=40)
        holder.append(agent)
        for _ in range(3):
            agent.play_steps()
        snaps.append(got)
        agent.detach()
    assert any(before > 0 for before, _, _ in snaps[1]) and all(after == 0 for _, _, after in snaps[1])     # fits WERE pending, and were flushed
    for (_, a, _), (_, b, _) in zip(snaps[0], snaps[1]):
        assert a.keys() == b.keys() and all(torch.equal(a[n], b[n]) for n in a)


@pytest.mark.gpu
def test_return_bookkeeping_inside_the_flags_launch_fits_the_same_network(monkeypatch):
    """The LocoVal return bookkeeping as part of the task's flags launch (emloco_task_post_physics_returns; LocoValRollout without a
    discriminator attaches its EmlocoLocoValStep to the task) against the launch of its own (EMLOCO_RETURNS_IN_FLAGS=0): same seeds,
    64 steps with natural resets -- fitted LocoVal parameters, episode count, return accumulators and simulator state bit-equal."""
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    from emloco_amd.run import RLGPUEnv
    E, outs = 256, []
    for in_flags in ("0", "1"):
        monkeypatch.setenv("EMLOCO_RETURNS_IN_FLAGS", in_flags)
        env = RLGPUEnv(_make_env(E, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel",
                                     "--input_init_pose", "--input_init_vel"]))
        task = env.env.task
        g = torch.Generator(device=task.device)
        g.manual_seed(77)
        pool = torch.randn(8, E, 69, device=task.device, generator=g) * 0.3
        k = [0]

        def pol(obs):
            k[0] += 1
            return pool[k[0] % 8]
        torch.manual_seed(5)
        agent = LocoValRollout(env, horizon_length=8, policy=pol, overlap_reset=False)
        assert agent._returns_in_flags == (in_flags == "1") and (task._returns_hook is not None) == (in_flags == "1")
        for _ in range(8):
            agent.play_steps()
        n_fit = agent.fitted_episodes                      # waits for the fit stream
        torch.cuda.synchronize()
        a = agent.acc
        outs.append((torch.cat([p.detach().reshape(-1) for p in agent.valuenet.parameters()]).clone(), n_fit,
                     a.current_rewards.clone(), a.current_lengths.clone(), a.current_combined_rewards.clone(), a.discount_coefs.clone(),
                     task._root_states.clone(), task.progress_buf.clone(), task.rew_buf.clone()))
        agent.detach()
        assert task._returns_hook is None
    assert outs[0][1] == outs[1][1] and outs[0][1] > 20
    for a, b in zip(outs[0], outs[1]):
        if torch.is_tensor(a):
            assert torch.equal(a, b)


@pytest.mark.gpu
def test_discriminator_beside_the_next_step_fits_the_same_network(monkeypatch):
    """configs[2] with the AMP discriminator OFF the chain between two rigid-body steps (LocoValRollout's deferred mode: the flags
    launch stages the step's return bookkeeping, one launch takes the GEMM operand out of the AMP observations, the side stream runs
    the discriminator, emloco_locoval_returns_finish and the fit beside the resets / policy / next step) against the sequential order
    (EMLOCO_DEFER_DISC=0: discriminator and bookkeeping between env.step and the next reset, as amp_continuous_value.py:63-145
    has them): same seeds, frozen policy reading the observations, 64 steps with natural resets -- fitted LocoVal parameters,
    episode count, return accumulators, last style rewards and simulator state bit-equal."""
    from emloco_amd.learning.amp_policy import AMPPolicyBundle
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    from emloco_amd.run import RLGPUEnv
    E, outs = 256, []
    for defer in ("0", "1"):
        monkeypatch.setenv("EMLOCO_DEFER_DISC", defer)
        env = RLGPUEnv(_make_env(E, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel",
                                     "--input_init_pose", "--input_init_vel"]))
        task = env.env.task
        torch.manual_seed(11)
        bundle = AMPPolicyBundle(task, seed=3, deterministic=False)
        with torch.no_grad():                              # livelier actions than a fresh network's: episodes end by falling too
            bundle.frozen.mu_b.add_(torch.randn_like(bundle.frozen.mu_b) * 0.3)
        torch.manual_seed(5)
        agent = LocoValRollout(env, horizon_length=8, policy=bundle.policy, disc_reward=bundle.disc_reward, overlap_reset=False)
        assert (agent._disc_halves is not None) == (defer == "1")
        assert (task._returns_hook is not None) == (defer == "1") and task.fused_chain and task.fused_amp_early
        for _ in range(8):
            agent.play_steps()
        n_fit = agent.fitted_episodes                      # waits for the fit stream
        torch.cuda.synchronize()
        a = agent.acc
        outs.append((torch.cat([p.detach().reshape(-1) for p in agent.valuenet.parameters()]).clone(), n_fit,
                     a.current_rewards.clone(), a.current_lengths.clone(), a.current_combined_rewards.clone(), a.discount_coefs.clone(),
                     bundle.frozen_disc.logits.clone(), task._root_states.clone(), task.progress_buf.clone(), task.rew_buf.clone()))
        agent.detach()
    assert outs[0][1] == outs[1][1] and outs[0][1] > 20
    for a, b in zip(outs[0], outs[1]):
        if torch.is_tensor(a):
            assert torch.equal(a, b)


@pytest.mark.gpu
def test_rollout_from_an_amass_pickle(tmp_path):
    """`--motion_file <pkl>` in the reference's AMASS format (utils/motion_lib_smpl.py; the clips of tests/golden/motion_amass.npz):
    the task builds one clip per env on that env's skeleton, resets sample reference states from it (host path and the fused device
    reset chain), the reset poses stand on the ground, the AMP demonstration batch comes from the clips, and the rollout runs."""
    import joblib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "motion_amass.npz"))
    clips = {f"clip{i}": {"pose_quat_global": g[f"c{i}_pose_quat_global"], "root_trans_offset": torch.from_numpy(g[f"c{i}_root_trans_offset"]),
                          "pose_aa": g[f"c{i}_pose_aa"], "beta": g[f"c{i}_beta"], "gender": "neutral", "fps": int(g[f"c{i}_fps"])} for i in range(2)}
    path = str(tmp_path / "amass_isaac_synthetic.pkl")
    joblib.dump(clips, path)
    from emloco_amd.utils.motion_lib_smpl import MotionLib
    E = 48
    env = _make_env(E, ["--random_heading", "--init_heading", "--adjust_root_vel", "--motion_file", path])
    task = env.task
    lib = task._motion_lib
    assert isinstance(lib, MotionLib) and lib.num_motions() == E and set(lib._curr_motion_ids.tolist()) <= {0, 1}
    assert lib.gts.is_cuda and lib.gts.shape[1:] == (24, 3)
    obs = env.reset(torch.arange(E, device=task.device))
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all()
    low = task._lowest_point(torch.arange(E, device=task.device))
    assert float(low.abs().max()) < 0.03                                 # grounded by the lowest collision point
    demo = task.fetch_amp_obs_demo(32)
    assert demo.shape == (32, task.get_num_amp_obs()) and torch.isfinite(demo).all() and float(demo.abs().max()) > 0.1
    for k in range(40):                                                  # natural resets through the fused device chain
        if hasattr(env, "reset_done"):
            env.reset_done()
        obs, rew, done, info = env.step(torch.randn(E, 69, device=task.device) * 0.3)
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(task._amp_obs_buf).all() and torch.isfinite(rew).all()
    assert int(task._sampled_motion_ids.max()) < E


@pytest.mark.gpu
def test_amp_agent_optimiser_step_as_a_hip_graph_equals_the_eager_step(monkeypatch):
    """AMPAgent's update step captured once as a HIP graph (static minibatch buffers, dropout draw and shuffled indices filled from the
    host, Adam with device-side step counters) and replayed, against the same agent issuing every launch eagerly: same seeds, two
    epochs of 2 x 4 minibatches -- network weights, observation statistics and the epoch's averaged losses agree to float rounding,
    and the graph really replays."""
    import yaml
    from emloco_amd.learning.amp_agent import AMPAgent
    from emloco_amd.learning.amp_policy import DEFAULT_CFG
    from emloco_amd.run import RLGPUEnv
    outs = []
    monkeypatch.setenv("EMLOCO_PPO_GRAPH", "1")              # both agents get the capturable Adam (the default Adam kernel's last-bit
    for mode in ("0", "1"):                                  # differences flip the sign of near-zero updates: lr per step, not rounding)
        torch.manual_seed(21)
        np.random.seed(21)
        env = RLGPUEnv(_make_env(64, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel"]))
        cfg = yaml.safe_load(open(DEFAULT_CFG))
        cfg["params"]["network"]["mlp"]["units"] = [256, 128]
        cfg["params"]["network"]["disc"]["units"] = [128, 64]
        cfg["params"]["config"].update(horizon_length=8, minibatch_size=128, amp_minibatch_size=128, amp_batch_size=64,
                                       amp_obs_demo_buffer_size=512, amp_replay_buffer_size=512, mini_epochs=2)
        agent = AMPAgent(env, cfg, seed=4)
        agent.use_graph = mode == "1"
        torch.manual_seed(33)
        infos = [agent.train_epoch() for _ in range(2)]
        assert (agent._graph is not None) == (mode == "1")
        outs.append((torch.cat([p.detach().reshape(-1) for p in agent.a2c_network.parameters()]).clone(), agent.running_mean_std.running_mean.clone(),
                     agent._amp_input_mean_std.running_var.clone(), infos[-1]))
    (w0, m0, v0, i0), (w1, m1, v1, i1) = outs
    assert torch.isfinite(w1).all() and (w0 - w1).abs().max().item() <= 2e-5 * w0.abs().max().item() + 1e-6
    assert torch.allclose(m0, m1, rtol=1e-9, atol=1e-9) and torch.allclose(v0, v1, rtol=1e-9, atol=1e-9)
    # (the epoch's averaged losses: the weights above agree to 2e-5 of their scale, i.e. up to Adam's +-lr on elements whose gradient is
    # rounding noise; the discriminator's loss on 128 samples moves by up to ~0.3 % with that)
    for k in ("actor_loss", "critic_loss", "disc_loss", "kl", "b_loss"):
        assert abs(i0[k] - i1[k]) <= 5e-3 * abs(i0[k]) + 1e-5, (k, i0[k], i1[k])


@pytest.mark.gpu
def test_amp_history_as_a_ring_holds_the_reference_layout_on_demand(monkeypatch):
    """The AMP history as a ring (HumanoidAMP.enable_amp_ring: the step moves a head instead of 14 rows per env; LocoValRollout without
    a discriminator switches it on) against the shifted layout (EMLOCO_AMP_RING=0): same seeds, 40 steps with natural resets -- the
    reference's (E, 15 x 206) tensor rebuilt from the ring equals the shifted buffer after every step, resets' back-filled history
    included; detach() leaves the task in the reference's layout; simulator state and fit are untouched by the choice."""
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    from emloco_amd.run import RLGPUEnv
    E, runs = 192, []
    for ring in ("0", "1"):
        monkeypatch.setenv("EMLOCO_AMP_RING", ring)
        env = RLGPUEnv(_make_env(E, ["--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel",
                                     "--input_init_pose", "--input_init_vel"]))
        task = env.env.task
        g = torch.Generator(device=task.device)
        g.manual_seed(41)
        pool = torch.randn(8, E, 69, device=task.device, generator=g) * 0.4
        k = [0]

        def pol(obs):
            k[0] += 1
            return pool[k[0] % 8]
        torch.manual_seed(9)
        agent = LocoValRollout(env, horizon_length=8, policy=pol, overlap_reset=False)
        assert task.amp_ring == (ring == "1")
        snaps, resets = [], 0
        for t in range(40):
            agent.step_once()
            if t % 3 != 2 or t == 39:                      # (the last step leaves its pass pending: detach() must take it in the ring's layout)
                continue
            task.wait_obs()                                # the live envs' observation / AMP pass is deferred into the next reset: take it now
            resets += int((task.reset_buf != 0).sum().item())
            snaps.append(task.amp_obs_logical().clone())
        assert (task.extras["amp_obs"] is None) == (ring == "1")
        n_fit = agent.fitted_episodes
        agent.detach()
        task.wait_obs()                                    # (the shifted-layout run still has its last pass pending)
        assert not task.amp_ring and task.extras["amp_obs"] is not None
        runs.append((snaps, task._amp_obs_buf.clone().view(E, -1), task._root_states.clone(), n_fit, resets, task._amp_head))
    assert runs[0][4] == runs[1][4] and runs[0][4] > 10 and runs[0][3] == runs[1][3]
    for t, (a, b) in enumerate(zip(runs[0][0], runs[1][0])):
        assert torch.equal(a, b), f"step {t}"
    assert torch.equal(runs[0][1], runs[1][1])                                                    # after detach: the reference's layout
    assert torch.equal(runs[0][2], runs[1][2])
