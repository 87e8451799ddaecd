"""CPU execution of the HIP kernel sources (tests/emu) against the oracle, numpy and the reference's golden vectors.

There is no GPU in the build container; the emulator compiles emloco_amd/csrc/*_kernels.hip with g++ and runs one
workgroup as 64/256 lock-step fibers (wave shuffles, ballots, LDS, barriers and the fp32 MFMA are emulated), so the
kernels' logic is covered by the `-m "not gpu"` suite.  The same checks run on the real MI355X in tests/test_gpu_*.py.
"""
import ctypes as C

import numpy as np
import pytest

import emu
import oracle
from helpers import oracle_sim, scene_state, varied_models


def ATTN_P8(p):
    """The drop probability the fused attention's mask realises: p in 1/256ths (csrc/attention_kernels.hip: at_drop_thr8)."""
    return min(max(int(p * 256.0 + 0.5), 1 if p > 0 else 0), 255) / 256.0


def P(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def test_sim_step_kernel_is_bit_exact_vs_oracle():
    E = 3
    models = varied_models(E, seed=2)
    root, dof, tgt = scene_state(E, seed=4)
    a = oracle_sim(models, root, dof, tgt, n_sub=4)
    b = oracle_sim(models, root, dof, tgt, n_sub=4)
    for _ in range(2):
        a.step(1)
        emu.sim_step(b, 1)
    for name in ("root_state", "dof_state", "rb_state", "contact_force", "dof_force", "lambda_ws"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert np.abs(a.contact_force).max() > 50


@pytest.mark.parametrize("parts", [2, 4])
def test_sim_step_kernel_split_launch_is_the_fused_step(parts, monkeypatch):
    """emloco_sim_set_split: the 4 substeps as 2 (or 4) dependent workgroups per env, each continuing from the registers its
    predecessor published -- the emulated kernel stays on the oracle's bytes (fused 4-substep step), contacts, warm start and
    self-collision included."""
    monkeypatch.setenv("EMLOCO_EMU_PARTS", str(parts))
    E = 3
    models = varied_models(E, seed=31)
    root, dof, tgt = scene_state(E, seed=32)
    a = oracle_sim(models, root, dof, tgt, self_collision=True, n_sub=4)
    b = oracle_sim(models, root, dof, tgt, self_collision=True, n_sub=4)
    for _ in range(4):
        a.step(1)
        emu.sim_step(b, 1)
    for name in ("root_state", "dof_state", "rb_state", "contact_force", "dof_force", "lambda_ws"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert np.abs(a.contact_force).max() > 50


def test_sim_step_kernel_split_launch_lost_handover_raises_the_error_word(monkeypatch):
    """A part whose predecessor never publishes its flag must not go on from stale hand-over state: its bounded wait runs
    out, the device error word gets EMLOCO_ERR_PART_TIMEOUT and the env's step is abandoned (state tensors untouched); the
    other envs are stepped as usual.  A stale flag of an earlier launch with another part count can never match a tag."""
    monkeypatch.setenv("EMLOCO_EMU_PARTS", "4")
    E = 3
    models = varied_models(E, seed=31)
    root, dof, tgt = scene_state(E, seed=32)
    a = oracle_sim(models, root, dof, tgt, self_collision=True, n_sub=4)
    b = oracle_sim(models, root, dof, tgt, self_collision=True, n_sub=4)
    a.step(1)
    emu.sim_step(b, 1)
    before = {n: getattr(b, n).copy() for n in ("root_state", "dof_state", "rb_state")}
    monkeypatch.setenv("EMLOCO_EMU_POISON", "1")
    a.step(1)
    assert emu.sim_step(b, 1, expect_error=True) == 1
    for name in ("root_state", "dof_state", "rb_state"):
        x, y = getattr(a, name).reshape(E, -1), getattr(b, name).reshape(E, -1)
        assert np.array_equal(x[[0, 2]], y[[0, 2]]), name                          # the others: the oracle's bytes
        assert np.array_equal(y[1], before[name].reshape(E, -1)[1]), name         # the poisoned env: abandoned, not corrupted


def test_sim_step_kernel_effort_drives_are_bit_exact_vs_oracle():
    """drive_mode = 1 (gym.set_dof_actuation_force_tensor, humanoid.py:1203-1207): joint torques instead of position targets,
    some beyond the effort limit, contacts and limb-limb contacts active -- the emulated kernel stays on the oracle's bytes."""
    E = 3
    models = varied_models(E, seed=41)
    root, dof, _ = scene_state(E, seed=42)
    rng = np.random.default_rng(43)
    torque = (rng.normal(size=(E, 69)) * 120.0).astype(np.float32)
    torque[:, ::7] *= 8.0                                               # beyond the 500 N m limit here and there
    a = oracle_sim(models, root, dof, torque, self_collision=True, n_sub=4, drive_mode=1)
    b = oracle_sim(models, root, dof, torque, self_collision=True, n_sub=4, drive_mode=1)
    for _ in range(3):
        a.step(1)
        emu.sim_step(b, 1)
    for name in ("root_state", "dof_state", "rb_state", "contact_force", "dof_force", "lambda_ws"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert np.isfinite(a.rb_state).all()
    lim = np.stack([m.effort for m in models]).astype(np.float32)
    assert np.array_equal(a.dof_force, np.clip(torque, -lim, lim))      # what the drives applied = the command within the limit


def test_sim_step_kernel_fallen_humanoids_are_bit_exact_vs_oracle():
    """Humanoids lying on the ground, pressed into it: more candidates than contact slots (the shallowest are dropped), limb-limb
    contacts, contact bodies at several tree depths in the Gram build.  (Pelvis contacts -- tree depth 0 -- come up in the
    168-step episodes of tests/test_gpu_sim.py on the hardware.)"""
    E = 3
    models = varied_models(E, seed=5)
    root, dof, tgt = scene_state(E, seed=6, height=0.06, perturbed_from=0)
    for e in range(E):                                   # lying on the back / side / front
        ax = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (-1.0, 0.0, 0.0)][e]
        s = np.sin(np.pi / 4)
        root[e, 3:7] = [ax[0] * s, ax[1] * s, ax[2] * s, np.cos(np.pi / 4)]
    a = oracle_sim(models, root, dof, tgt, self_collision=True, n_sub=4)
    b = oracle_sim(models, root, dof, tgt, self_collision=True, n_sub=4)
    for _ in range(3):
        a.step(1)
        emu.sim_step(b, 1)
    for name in ("root_state", "dof_state", "rb_state", "contact_force", "dof_force", "lambda_ws"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    cf = np.abs(a.contact_force).sum(-1)
    assert ((cf > 0).sum(1) >= 1).all()


def test_sim_step_kernel_with_self_collision_is_bit_exact_vs_oracle():
    """phase 1b (limb-limb penalty contacts): folding ragdolls (drives off, limbs thrown together) keep the emulated kernel
    and the oracle on identical bytes, and the contacts really fire"""
    E = 3
    models = varied_models(E, seed=5)
    root, dof, tgt = scene_state(E, seed=6, perturbed_from=0)
    dof[:, :, 0] *= 4.0                         # strongly bent limbs -> overlapping capsules
    dof[:, :, 1] *= 2.0
    for m in models:
        m.kp = m.kp * 0.05                      # weak drives: the limbs keep colliding
    a = oracle_sim(models, root, dof, tgt, self_collision=True, n_sub=4)
    b = oracle_sim(models, root, dof, tgt, self_collision=True, n_sub=4)
    c = oracle_sim(models, root, dof, tgt, n_sub=4)
    for _ in range(3):
        a.step(1)
        emu.sim_step(b, 1)
        c.step(1)
    for name in ("root_state", "dof_state", "rb_state", "contact_force", "dof_force", "lambda_ws"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert not np.array_equal(a.dof_state, c.dof_state)        # self-collision changed the motion


def test_sim_step_kernel_with_saturated_drives_is_bit_exact_vs_oracle():
    """targets far enough away that drives hit their effort limit: the second factorise / solve pass runs in some envs and
    not in others; bytes equal the oracle and the reported drive torque sits exactly on the limit somewhere"""
    E = 4
    models = varied_models(E, seed=11)
    root, dof, tgt = scene_state(E, seed=12)
    a = oracle_sim(models, root, dof, tgt, n_sub=1)
    b = oracle_sim(models, root, dof, tgt, n_sub=1)
    rng = np.random.default_rng(13)
    eff = np.stack([m.effort for m in models]).astype(np.float32)
    hit = False
    for _ in range(6):
        big = (rng.normal(size=(E, 69)) * 2.0).astype(np.float32)
        big[0] = tgt[0]                              # env 0 keeps standing (single pass), the others saturate
        a.pd_target[:] = big
        b.pd_target[:] = big
        a.step(1)
        emu.sim_step(b, 1)
        for name in ("root_state", "dof_state", "rb_state", "contact_force", "dof_force", "lambda_ws"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name
        hit = hit or bool((np.abs(a.dof_force[1:]) == eff[1:]).any())
        assert not (np.abs(a.dof_force[0]) == eff[0]).any()
    assert hit
    wmax = np.abs(a.rb_state[:, :, 10:13]).max()
    assert np.isfinite(a.rb_state).all() and 48.0 < wmax < 150      # 2 rad moves within a few substeps: above the link-speed cap
                                                                    # (0.4 rad / substep = 48 rad/s), so the limiter branch ran too


def test_sim_step_kernel_on_a_heightfield_is_bit_exact_vs_oracle():
    """height-field ground (sloped + bumpy): per-contact normals and tangent frames; emulated kernel == oracle bytes, and
    the terrain really changes the motion relative to the plane"""
    from helpers import bumpy_heightfield
    E = 3
    models = varied_models(E, seed=7)
    root, dof, tgt = scene_state(E, seed=8)
    hf = bumpy_heightfield(seed=3, amp=0.1, slope=0.15)
    a = oracle_sim(models, root, dof, tgt, heightfield=hf, n_sub=4)
    b = oracle_sim(models, root, dof, tgt, heightfield=hf, n_sub=4)
    c = oracle_sim(models, root, dof, tgt, n_sub=4)
    for _ in range(3):
        a.step(1)
        emu.sim_step(b, 1)
        c.step(1)
    for name in ("root_state", "dof_state", "rb_state", "contact_force", "dof_force", "lambda_ws"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert np.abs(a.contact_force).max() > 50 and np.abs(a.contact_force[..., :2]).max() > 5      # tilted contact forces
    assert not np.array_equal(a.rb_state, c.rb_state)


def test_sim_fk_kernel_matches_oracle():
    E = 2
    models = varied_models(E, seed=5)
    root, dof, tgt = scene_state(E, seed=6, perturbed_from=0)
    a = oracle_sim(models, root, dof, tgt)
    b = oracle_sim(models, root, dof, tgt)
    a.fk()
    emu.sim_fk(b)
    assert np.array_equal(a.rb_state, b.rb_state)


def test_post_physics_kernel_matches_reference_golden(golden):
    from emloco_amd import _lib as L
    g, gt, gs = golden("self_obs"), golden("terrain_heights"), golden("traj_samples")
    E = 16
    th = emu.TaskHost(E, gt["heightfield"])
    th.rb_state[:] = np.concatenate([g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"]], -1)
    th.betas[:] = g["betas"]
    th.traj_verts[:] = gs["verts"]
    th.progress[:] = gs["progress"] - 1          # ADVANCE brings it to the fixture's value
    rng = np.random.default_rng(0)
    th.dof_state[:] = rng.normal(size=(E, 69, 2))
    th.dof_force[:] = rng.normal(size=(E, 69)) * 30
    th.contact_force[:, 11] = rng.normal(size=(E, 3)) * 40
    th.amp[:] = rng.normal(size=th.amp.shape)
    amp_before = th.amp.copy()
    th.post_physics(L.POST_STEP)
    np.testing.assert_array_equal(th.progress, gs["progress"])
    np.testing.assert_allclose(th.obs[:, :368], g["obs"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(th.flip_obs[:, :368], g["flip_obs"], rtol=1e-5, atol=2e-5)
    root_states = np.concatenate([g["body_pos"][:, 0], g["body_rot"][:, 0], g["body_vel"][:, 0], g["body_ang_vel"][:, 0]], -1)
    np.testing.assert_allclose(th.obs[:, 368:398], oracle.location_obs(root_states, gs["samples"]), rtol=1e-5, atol=2e-5)
    head = np.concatenate([g["body_pos"][:, 13], g["body_rot"][:, 13]], -1)
    hf = gt["heightfield"]
    ho = oracle.height_obs(oracle.get_center_heights(root_states, hf), oracle.get_heights(head, hf))
    np.testing.assert_array_equal(th.obs[:, 398:], ho)                 # height obs: index work, bit-exact vs the oracle
    np.testing.assert_array_equal(th.flip_obs[:, 368:], oracle.flip_task_obs(th.obs[:, 368:]))
    tar = oracle.traj_calc_pos(gs["verts"], gs["progress"], th.dt, th.traj_dur)
    rew, raw = oracle.reward(g["body_pos"][:, 0], tar, th.dof_force, th.dof_state[:, :, 1])
    np.testing.assert_allclose(th.rew, rew, rtol=1e-5, atol=1e-6)
    rs, tm = oracle.reset(gs["progress"], th.contact_force, g["body_pos"], tar)
    np.testing.assert_array_equal(th.reset, rs)
    np.testing.assert_array_equal(th.terminate, tm)
    amp = oracle.amp_obs(g["body_pos"][:, 0], g["body_rot"][:, 0], g["body_vel"][:, 0], g["body_ang_vel"][:, 0],
                         th.dof_state[:, :, 0], th.dof_state[:, :, 1], g["body_pos"][:, [7, 3, 22, 17]], g["betas"], th.dof_subset)
    np.testing.assert_allclose(th.amp[:, 0], amp, rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(th.amp[:, 1:], amp_before[:, :-1])


def test_fused_launch_observation_role_equals_the_post_physics_launch(golden):
    """reset_obs_kernel with an empty finished-env list: the workgroups behind the reset / history roles run the deferred observation /
    AMP pass of every env whose flag-snapshot entry is zero -- the same bytes as post_physics_kernel(OBS | AMP_SHIFT | AMP_ROW |
    SKIP_DONE) on the same state, skipped envs untouched (the split `task.fused_chain` relies on; the reset roles need a simulator
    and are compared on the GPU: tests/test_gpu_env.py)."""
    from emloco_amd import _lib as L
    g, gt, gs = golden("self_obs"), golden("terrain_heights"), golden("traj_samples")
    E = 16
    rng = np.random.default_rng(3)
    hosts = []
    for _ in range(2):
        th = emu.TaskHost(E, gt["heightfield"])
        th.rb_state[:] = np.concatenate([g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"]], -1)
        th.betas[:] = g["betas"]
        th.traj_verts[:] = gs["verts"]
        th.progress[:] = gs["progress"]
        hosts.append(th)
    dof, amp0 = rng.normal(size=(E, 69, 2)), rng.normal(size=hosts[0].amp.shape)
    skip = np.zeros(E, np.int64); skip[[2, 7, 8]] = 1
    for th in hosts:
        th.dof_state[:] = dof; th.amp[:] = amp0
        th.obs[:] = -5.0; th.flip_obs[:] = -6.0
        th.reset[:] = skip
    mode = L.POST_OBS | L.POST_AMP_SHIFT | L.POST_AMP_ROW
    hosts[0].post_physics(mode | L.POST_SKIP_DONE)
    b = hosts[1].bufs()
    ids = np.full(E + 1, -1, np.int32); ids[E] = 0
    fn = emu.lib().emu_reset_obs_live
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 4
    assert fn(C.byref(b), mode, P(skip), P(ids), E, 5, 3, 14) == 0       # 5 reset slots, 3 x 14 history workgroups ahead of / behind the role
    for name in ("obs", "flip_obs", "amp", "progress", "reset", "rew"):
        np.testing.assert_array_equal(getattr(hosts[0], name), getattr(hosts[1], name), err_msg=name)
    assert np.all(hosts[1].obs[[2, 7, 8]] == -5.0) and np.all(hosts[1].obs[0] != -5.0)


def test_flags_launch_with_the_return_bookkeeping_equals_the_two_launches(golden):
    """post_physics_returns_kernel (emloco_task_post_physics_returns: the LocoVal return bookkeeping of an env right behind its reward
    and reset flag, one launch) against post_physics_kernel followed by locoval_returns_kernel on its outputs: task buffers, return
    accumulators, staged LocoVal inputs, targets and weights byte for byte, over several steps with episodes ending."""
    from emloco_amd import _lib as L
    from emloco_amd.predictor.ops import LocoValStep
    g, gt, gs = golden("self_obs"), golden("terrain_heights"), golden("traj_samples")
    E = 16
    rng = np.random.default_rng(9)
    mode = L.POST_ADVANCE | L.POST_REWARD | L.POST_RESET | L.POST_AMP_SHIFT | L.POST_AMP_ROW | L.POST_AMP_DONE_ONLY
    f = lambda *s: np.zeros(s, np.float32)
    sides = []
    for _ in range(2):
        th = emu.TaskHost(E, gt["heightfield"], episode_len=6)
        th.rb_state[:] = np.concatenate([g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"]], -1)
        th.betas[:] = g["betas"]
        th.traj_verts[:] = gs["verts"]
        th.progress[:] = 0; th.reset[:] = 0; th.terminate[:] = 0
        st = dict(cr=f(E), cl=f(E), cc=f(E), dc=np.ones(E, np.float32), traj13=f(E, 13, 3), pose=f(E, 24, 3), vel=f(E, 2), target=f(E), weight=f(E),
                  wp=rng.normal(size=(E, 15, 3)).astype(np.float32), ip=rng.normal(size=(E, 24, 3)).astype(np.float32), iv=rng.normal(size=(E, 2)).astype(np.float32))
        sides.append((th, st))
    for k in ("wp", "ip", "iv"):
        sides[1][1][k][:] = sides[0][1][k]
    p = lambda a: a.ctypes.data
    steps = [LocoValStep(E, 4, 0.99, 0.3, -10.0, 100.0, p(st["cr"]), p(st["cl"]), p(st["cc"]), p(st["dc"]), p(st["wp"]), p(st["ip"]), p(st["iv"]),
                         p(st["traj13"]), p(st["pose"]), p(st["vel"]), p(st["target"]), p(st["weight"])) for _, st in sides]
    lib = emu.lib()
    lib.emu_locoval_returns.argtypes = [C.c_void_p] * 5
    lib.emu_task_post_physics_returns.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    inv = (rng.random(E) < 0.4).astype(np.uint8)
    emitted = 0
    for t in range(9):
        dof, frc = rng.normal(size=(E, 69, 2)).astype(np.float32), (rng.normal(size=(E, 69)) * 30).astype(np.float32)
        cf = np.zeros((E, 24, 3), np.float32); cf[rng.random(E) < 0.2, 11] = 80.0          # some envs fall
        for th, _ in sides:
            th.dof_state[:] = dof; th.dof_force[:] = frc; th.contact_force[:] = cf
            th.progress[th.reset != 0] = 0                                                   # a finished env starts over
        (tha, sta), (thb, stb) = sides
        tha.post_physics(mode)
        rew, dones = np.ascontiguousarray(tha.rew), np.ascontiguousarray(tha.reset)
        lib.emu_locoval_returns(C.addressof(steps[0]), p(rew), None, p(dones), p(inv))
        bb = thb.bufs()
        assert lib.emu_task_post_physics_returns(C.byref(bb), mode, C.addressof(steps[1]), p(inv)) == 0
        for name in ("progress", "reset", "terminate", "rew", "reward_raw", "amp"):
            np.testing.assert_array_equal(getattr(tha, name), getattr(thb, name), err_msg=f"{name} step {t}")
        for k in sta:
            np.testing.assert_array_equal(sta[k], stb[k], err_msg=f"{k} step {t}")
        emitted += int((sta["weight"] != 0).sum())
    assert emitted > 0 and int((sides[0][0].reset != 0).sum()) >= 0


def test_pd_targets_kernel_matches_reference_golden(golden):
    g = golden("pd_targets")
    zero = np.zeros(69, np.uint8)
    for j in (3, 7, 17, 22):
        zero[3 * j:3 * j + 3] = 1
    np.testing.assert_allclose(emu.task_pd_targets(g["actions"], g["offset"], g["scale"], zero), g["pd_tar"], rtol=1e-6, atol=1e-6)


def _gemm(A, B, ta, tb, m, n, k, batch=1, bias=None, flags=0, ksplit=1, alpha=1.0):
    lib = emu.lib()
    out = np.zeros((batch, m, n), np.float32)
    ws = np.zeros((max(ksplit, 1), batch, m, n), np.float32)
    lib.emu_gemm_f32.argtypes = [C.c_int] * 4 + [C.c_float, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_int,
                                                 C.c_long, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.emu_gemm_f32(batch, m, n, k, alpha, P(A), A.shape[-1], A[0].size if batch > 1 else 0, ta, P(B), B.shape[-1],
                     B[0].size if batch > 1 else 0, tb, P(out), n, m * n, P(bias), flags, ksplit, P(ws))
    return out


def test_mfma_gemm_kernel_all_layouts():
    rng = np.random.default_rng(0)
    m, n, k = 150, 70, 37       # ragged: exercises every tile edge
    A = rng.normal(size=(1, m, k)).astype(np.float32)
    B = rng.normal(size=(1, n, k)).astype(np.float32)
    ref = A[0].astype(np.float64) @ B[0].astype(np.float64).T
    tol = dict(rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(_gemm(A, B, 0, 0, m, n, k)[0], ref, **tol)
    np.testing.assert_allclose(_gemm(A, B, 0, 0, m, n, k, ksplit=3)[0], ref, **tol)
    At, Bt = np.ascontiguousarray(A.transpose(0, 2, 1)), np.ascontiguousarray(B.transpose(0, 2, 1))
    np.testing.assert_allclose(_gemm(At, Bt, 1, 1, m, n, k)[0], ref, **tol)
    np.testing.assert_allclose(_gemm(At, B, 1, 0, m, n, k, alpha=0.5)[0], 0.5 * ref, **tol)
    bias = rng.normal(size=n).astype(np.float32)
    np.testing.assert_allclose(_gemm(A, B, 0, 0, m, n, k, bias=bias, flags=3)[0], np.maximum(ref + bias, 0), **tol)
    Bn = rng.normal(size=(1, 32, k)).astype(np.float32)          # n <= 32 takes the 128x32 tile variant
    np.testing.assert_allclose(_gemm(A, Bn, 0, 0, m, 32, k)[0], A[0].astype(np.float64) @ Bn[0].astype(np.float64).T, **tol)
    # 16-byte aligned operands (k, m, n multiples of 4): the vector-load paths, with ragged row / k tails
    m2, n2, k2 = 148, 136, 44
    A2 = rng.normal(size=(1, m2, k2)).astype(np.float32)
    B2 = rng.normal(size=(1, n2, k2)).astype(np.float32)
    ref2 = A2[0].astype(np.float64) @ B2[0].astype(np.float64).T
    np.testing.assert_allclose(_gemm(A2, B2, 0, 0, m2, n2, k2)[0], ref2, **tol)
    A2t, B2t = np.ascontiguousarray(A2.transpose(0, 2, 1)), np.ascontiguousarray(B2.transpose(0, 2, 1))
    np.testing.assert_allclose(_gemm(A2t, B2t, 1, 1, m2, n2, k2)[0], ref2, **tol)
    np.testing.assert_allclose(_gemm(A2, B2t, 0, 1, m2, n2, k2, ksplit=2)[0], ref2, **tol)
    # long reduction (k > 256): the 32-deep stage variant, both tile shapes, aligned and ragged k
    for k3, n3 in ((300, 40), (290, 24), (521, 133)):
        A3 = rng.normal(size=(1, 70, k3)).astype(np.float32)
        B3 = rng.normal(size=(1, n3, k3)).astype(np.float32)
        ref3 = A3[0].astype(np.float64) @ B3[0].astype(np.float64).T
        np.testing.assert_allclose(_gemm(A3, B3, 0, 0, 70, n3, k3)[0], ref3, rtol=1e-5, atol=1e-4)
        B3t = np.ascontiguousarray(B3.transpose(0, 2, 1))
        np.testing.assert_allclose(_gemm(A3, B3t, 0, 1, 70, n3, k3)[0], ref3, rtol=1e-5, atol=1e-4)
    Ab = rng.normal(size=(3, 40, 20)).astype(np.float32)
    Bb = rng.normal(size=(3, 50, 20)).astype(np.float32)
    np.testing.assert_allclose(_gemm(Ab, Bb, 0, 0, 40, 50, 20, batch=3), np.einsum("bmk,bnk->bmn", Ab, Bb), **tol)


def test_gemm_epilogue_fuses_relu_dropout_backward_and_bias_gradient():
    """emloco_gemm_relu_bwd's kernel: C = (A . B) o [y > 0] * scale with the 64-row column sums of C (the partials of the bias
    gradient), ragged m / n, both layouts of B, both stage depths: C is BIT-equal to the plain GEMM followed by the mask."""
    lib = emu.lib()
    rng = np.random.default_rng(12)
    for m, n, k, tb in ((150, 70, 36, 1), (200, 136, 44, 0), (130, 260, 300, 1)):
        A = rng.normal(size=(m, k)).astype(np.float32)
        B = rng.normal(size=(k, n) if tb else (n, k)).astype(np.float32)
        y = np.maximum(rng.normal(size=(m, n)), 0).astype(np.float32)
        scale = np.float32(1 / 0.9)
        Cm = np.full((m, n), 7.0, np.float32)
        nparts = 2 * ((m + 127) // 128)
        part = np.zeros((nparts, n), np.float32)
        lib.emu_gemm_relu_bwd(m, n, k, P(A), k, P(B), n if tb else k, tb, P(Cm), P(y), C.c_float(scale), P(part))
        plain = _gemm(A[None], B[None], 0, tb, m, n, k)[0]
        want = np.where(y > 0, plain * scale, np.float32(0)).astype(np.float32)
        assert np.array_equal(Cm, want)
        sums = np.stack([want[r:r + 64].astype(np.float64).sum(0) for r in range(0, 64 * nparts, 64)])
        np.testing.assert_allclose(part, sums, rtol=1e-5, atol=2e-4)


def test_mfma_gemm_kernel_split_mode_is_fp32_class():
    """EMLOCO_GEMM_SPLIT: fp32 operands as three bf16 pieces each, six piece products on v_mfma_f32_32x32x16_bf16.  The emulated
    kernel (same source, same LDS image, same thread maps) against float64: all four layouts with ragged rows / columns / k, split
    k, bias + ReLU, a long reduction, values over a wide range of magnitudes -- inside the plain fp32 kernel's own tolerance, and
    the pieces must rebuild every operand exactly (a wrong slot, plane or k pairing shows as an error of 2^-8, not 2^-24)."""
    rng = np.random.default_rng(21)
    SPLIT = 1024
    lib_ = emu.lib()
    lib_.emu_gemm_split_image_words.restype = C.c_long
    # (256 x 256 x 80 and 256 x 128 x 16: every tile inside both operands, whole stages -- the main loop without clamps and masks, round 5)
    for m, n, k in ((148, 136, 44), (72, 260, 520), (256, 128, 16), (256, 256, 80)):
        A = (rng.normal(size=(1, m, k)) * np.exp(rng.uniform(-3, 3, size=(1, m, k)))).astype(np.float32)
        B = (rng.normal(size=(1, n, k)) * np.exp(rng.uniform(-3, 3, size=(1, n, k)))).astype(np.float32)
        ref = A[0].astype(np.float64) @ B[0].astype(np.float64).T
        mag = np.abs(A[0]).astype(np.float64) @ np.abs(B[0]).astype(np.float64).T
        At, Bt = np.ascontiguousarray(A.transpose(0, 2, 1)), np.ascontiguousarray(B.transpose(0, 2, 1))
        plain_err = np.max(np.abs(_gemm(A, B, 0, 0, m, n, k)[0] - ref) / mag)
        for (a, b, ta, tb) in ((A, B, 0, 0), (At, Bt, 1, 1), (A, Bt, 0, 1), (At, B, 1, 0)):
            got = _gemm(a, b, ta, tb, m, n, k, flags=SPLIT)[0]
            err = np.max(np.abs(got - ref) / mag)
            assert err < 4 * plain_err + 2e-7, (m, n, k, ta, tb, err, plain_err)     # (a misplaced piece would show as 2^-8 = 4e-3)
        got = _gemm(A, Bt, 0, 1, m, n, k, flags=SPLIT, ksplit=2)[0]
        assert np.max(np.abs(got - ref) / mag) < 4 * plain_err + 2e-7
        # the 64 x 64 tile the launcher picks for launches too small to fill the chip with 128 x 128 tiles (gemm_split_small_kernel):
        # same pieces, same six products per 16 k -- the same bar, and the SAME BITS as the 128 x 128 tile (the reduction order of an
        # output element does not depend on the tile it sits in)
        SMALL = 1 << 20
        for (a, b, ta, tb) in ((A, B, 0, 0), (At, Bt, 1, 1), (A, Bt, 0, 1), (At, B, 1, 0)):
            got = _gemm(a, b, ta, tb, m, n, k, flags=SPLIT | SMALL)[0]
            assert np.max(np.abs(got - ref) / mag) < 4 * plain_err + 2e-7, ("small tile", m, n, k, ta, tb)
            assert np.array_equal(got, _gemm(a, b, ta, tb, m, n, k, flags=SPLIT)[0]), ("small tile bits", m, n, k, ta, tb)
        got = _gemm(At, Bt, 1, 1, m, n, k, flags=SPLIT | SMALL, ksplit=3)[0]
        assert np.max(np.abs(got - ref) / mag) < 4 * plain_err + 2e-7
        # B as its PIECE IMAGE (EMLOCO_GEMM_B_SPLITIMG, round 5: a weight cut once by gemm_split_pack_kernel instead of by every
        # workgroup): from either layout of the matrix, on both tiles, with split k -- the SAME BITS as the matrix itself
        IMG = 2048
        words = lib_.emu_gemm_split_image_words(n, k)
        for Bsrc, trans, ld in ((B, 0, k), (Bt, 1, n)):
            img = np.zeros((1, words), np.uint32)
            lib_.emu_gemm_split_pack(P(Bsrc), n, k, ld, trans, C.c_void_p(img.ctypes.data))
            want = _gemm(A, B, 0, 0, m, n, k, flags=SPLIT)[0]
            assert np.array_equal(_gemm(A, img.view(np.float32), 0, 0, m, n, k, flags=SPLIT | IMG)[0], want), ("image", m, n, k, trans)
            assert np.array_equal(_gemm(A, img.view(np.float32), 0, 0, m, n, k, flags=SPLIT | IMG | SMALL)[0], want), ("image, small tile", m, n, k, trans)
            assert np.array_equal(_gemm(A, img.view(np.float32), 0, 0, m, n, k, flags=SPLIT | IMG, ksplit=2)[0],
                                  _gemm(A, B, 0, 0, m, n, k, flags=SPLIT, ksplit=2)[0]), ("image, split k", m, n, k, trans)
        bias = rng.normal(size=n).astype(np.float32)
        got = _gemm(A, B, 0, 0, m, n, k, bias=bias, flags=SPLIT | 3, alpha=0.5)[0]
        np.testing.assert_allclose(got, np.maximum(0.5 * ref + bias, 0), rtol=0, atol=2e-6 * float(mag.max()))
        # EMLOCO_GEMM_SPLIT2 (round 6: the backward's gradient products): TWO pieces per operand, the three piece products above 2^-16 --
        # the dropped ones (a2 b2, a1 b3, a3 b1) are each at most 2^-16 of |a| |b|: error under 3 x 2^-16 of the magnitude sum, on every
        # layout, both tiles (same bits), with split k; and well above the three-piece error (the kernel really runs on two)
        SPLIT2 = 4096
        for (a, b, ta, tb) in ((A, B, 0, 0), (At, Bt, 1, 1), (A, Bt, 0, 1), (At, B, 1, 0)):
            got = _gemm(a, b, ta, tb, m, n, k, flags=SPLIT | SPLIT2)[0]
            err = np.max(np.abs(got - ref) / mag)
            assert 4 * plain_err + 2e-7 < err < 3 * 2.0 ** -16, ("two pieces", m, n, k, ta, tb, err)
            assert np.array_equal(got, _gemm(a, b, ta, tb, m, n, k, flags=SPLIT | SPLIT2 | SMALL)[0]), ("two pieces, small tile bits", m, n, k, ta, tb)
        got = _gemm(At, Bt, 1, 1, m, n, k, flags=SPLIT | SPLIT2, ksplit=3)[0]
        assert np.max(np.abs(got - ref) / mag) < 3 * 2.0 ** -16
    # operands the split mode does not serve (unaligned, narrow n) fall back to the plain kernel: bit-equal to it
    A = rng.normal(size=(1, 70, 37)).astype(np.float32)
    B = rng.normal(size=(1, 50, 37)).astype(np.float32)
    assert np.array_equal(_gemm(A, B, 0, 0, 70, 50, 37, flags=SPLIT), _gemm(A, B, 0, 0, 70, 50, 37))
    # the fused backward epilogue in split mode: mask, scale and the bias-gradient partials of what was written
    lib = emu.lib()
    for m, n, k, tb in ((150, 72, 36, 1), (200, 136, 44, 0)):
        A = rng.normal(size=(m, k)).astype(np.float32)
        B = rng.normal(size=(k, n) if tb else (n, k)).astype(np.float32)
        y = np.maximum(rng.normal(size=(m, n)), 0).astype(np.float32)
        scale = np.float32(1 / 0.9)
        Cm = np.full((m, n), 7.0, np.float32)
        nparts = 2 * ((m + 127) // 128)
        part = np.zeros((nparts, n), np.float32)
        lib.emu_gemm_relu_bwd_ex(m, n, k, P(A), k, P(B), n if tb else k, tb, P(Cm), P(y), C.c_float(scale), P(part), SPLIT)
        plain = _gemm(A[None], B[None], 0, tb, m, n, k, flags=SPLIT)[0]
        want = np.where(y > 0, plain * scale, np.float32(0)).astype(np.float32)
        assert np.array_equal(Cm, want)
        sums = np.stack([want[r:r + 64].astype(np.float64).sum(0) for r in range(0, 64 * nparts, 64)])
        np.testing.assert_allclose(part, sums, rtol=1e-5, atol=2e-4)
        # ... and on two pieces (EMLOCO_GEMM_SPLIT2): the same epilogue around the two-piece product
        Cm2 = np.full((m, n), 7.0, np.float32)
        part_two = np.zeros_like(part)
        lib.emu_gemm_relu_bwd_ex(m, n, k, P(A), k, P(B), n if tb else k, tb, P(Cm2), P(y), C.c_float(scale), P(part_two), SPLIT | 4096)
        plain2 = _gemm(A[None], B[None], 0, tb, m, n, k, flags=SPLIT | 4096)[0]
        assert np.array_equal(Cm2, np.where(y > 0, plain2 * scale, np.float32(0)).astype(np.float32)) and not np.array_equal(Cm2, Cm)
        img = np.zeros(lib.emu_gemm_split_image_words(n, k), np.uint32)          # the weight as its piece image: same bits
        lib.emu_gemm_split_pack(P(B), n, k, n if tb else k, tb, C.c_void_p(img.ctypes.data))
        Cm2, part2 = np.full((m, n), 7.0, np.float32), np.zeros((nparts, n), np.float32)
        lib.emu_gemm_relu_bwd_ex(m, n, k, P(A), k, C.c_void_p(img.ctypes.data), 0, 0, P(Cm2), P(y), C.c_float(scale), P(part2), SPLIT | 2048)
        assert np.array_equal(Cm2, Cm) and np.array_equal(part2, part)


def test_mfma_gemm_kernel_bf16_operands():
    """EMLOCO_GEMM_BF16 (opt-in): operands rounded to bf16 on their way into the matrix cores, fp32 accumulation.  Against
    the product of the bf16-ROUNDED operands the kernel is exact to fp32 accumulation error; against the fp32 product the
    error is the operand rounding (2^-9 relative per factor).  All four layouts, both stage depths, ragged edges."""
    rng = np.random.default_rng(5)

    def bf16(x):
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
        return u.astype(np.uint32).view(np.float32)
    for m, n, k in ((148, 136, 44), (72, 132, 520)):
        A = rng.normal(size=(1, m, k)).astype(np.float32)
        B = rng.normal(size=(1, n, k)).astype(np.float32)
        exact = bf16(A[0]).astype(np.float64) @ bf16(B[0]).astype(np.float64).T
        full = A[0].astype(np.float64) @ B[0].astype(np.float64).T
        At, Bt = np.ascontiguousarray(A.transpose(0, 2, 1)), np.ascontiguousarray(B.transpose(0, 2, 1))
        for a_, b_, ta, tb in ((A, B, 0, 0), (At, Bt, 1, 1), (A, Bt, 0, 1), (At, B, 1, 0)):
            got = _gemm(a_, b_, ta, tb, m, n, k, flags=16)[0]
            np.testing.assert_allclose(got, exact, rtol=1e-5, atol=1e-4)
            assert 1e-4 < np.abs(got - full).max() < 0.02 * np.sqrt(k)          # really reduced precision, and only that
        bias = rng.normal(size=n).astype(np.float32)
        np.testing.assert_allclose(_gemm(A, B, 0, 0, m, n, k, bias=bias, flags=16 | 3)[0], np.maximum(exact + bias, 0), rtol=1e-5, atol=1e-4)
    # shapes without 16-byte alignment fall back to the fp32 operand path (flag ignored, full precision)
    A = rng.normal(size=(1, 37, 30)).astype(np.float32)
    B = rng.normal(size=(1, 41, 30)).astype(np.float32)
    At = np.ascontiguousarray(A.transpose(0, 2, 1))[:, :, :]
    np.testing.assert_allclose(_gemm(A[:, :, :29].copy(), B[:, :, :29].copy(), 0, 0, 37, 41, 29, flags=16)[0],
                               A[0, :, :29].astype(np.float64) @ B[0, :, :29].astype(np.float64).T, rtol=1e-5, atol=2e-5)


def test_softmax_and_layernorm_kernels():
    lib = emu.lib()
    rng = np.random.default_rng(1)
    n_seq, rps, cols = 3, 5, 77
    S = rng.normal(size=(n_seq * rps, cols)).astype(np.float32) * 3
    kp = np.zeros((n_seq, cols), np.float32)
    kp[0, ::3] = 1.0                  # float mask = additive bias (what the reference actually passes)
    kp[1, 50:] = -np.inf              # bool-style mask
    kp[2, :] = -np.inf                # fully padded sequence -> zeros, not NaN
    Pm = np.zeros_like(S)
    lib.emu_softmax_fwd(n_seq, rps, cols, C.c_float(0.25), P(S), P(kp), P(Pm))
    z = S.astype(np.float64) * 0.25 + np.repeat(kp, rps, 0)
    with np.errstate(invalid="ignore"):
        e = np.exp(z - np.nanmax(np.where(np.isinf(z), np.nan, z), axis=1, keepdims=True))
    e[np.isnan(e)] = 0
    ref = e / np.maximum(e.sum(1, keepdims=True), 1e-300)
    ref[2 * rps:] = 0
    np.testing.assert_allclose(Pm, ref, rtol=1e-5, atol=1e-7)
    dP = rng.normal(size=S.shape).astype(np.float32)
    dS = np.zeros_like(S)
    lib.emu_softmax_bwd(n_seq * rps, cols, C.c_float(0.25), P(Pm), P(dP), P(dS))
    refd = 0.25 * ref * (dP - (dP * ref).sum(1, keepdims=True))
    np.testing.assert_allclose(dS, refd, rtol=1e-4, atol=1e-6)

    # rows longer than 1024 take the streaming path
    Sl = rng.normal(size=(2, 1100)).astype(np.float32)
    Pl = np.zeros_like(Sl)
    lib.emu_softmax_fwd(2, 1, 1100, C.c_float(0.5), P(Sl), None, P(Pl))
    el = np.exp(0.5 * Sl.astype(np.float64) - (0.5 * Sl).max(1, keepdims=True))
    np.testing.assert_allclose(Pl, el / el.sum(1, keepdims=True), rtol=1e-4, atol=1e-7)
    dPl, dSl = rng.normal(size=Sl.shape).astype(np.float32), np.zeros_like(Sl)
    lib.emu_softmax_bwd(2, 1100, C.c_float(0.5), P(Pl), P(dPl), P(dSl))
    np.testing.assert_allclose(dSl, 0.5 * Pl * (dPl - (dPl * Pl).sum(1, keepdims=True)), rtol=1e-4, atol=1e-7)

    for rows, d in ((2130, 128), (2129, 128), (700, 200)):     # 34 row blocks: two fold levels; d = 128 has its own backward (two rows per wave: odd tail)
        x = rng.normal  (size=(rows, d)).astype(np.float32)
        res = rng.normal(size=(rows, d)).astype(np.float32)
        gam = rng.normal(size=d).astype(np.float32)
        bet = rng.normal(size=d).astype(np.float32)
        y, mean, rstd = np.zeros_like(x), np.zeros(rows, np.float32), np.zeros(rows, np.float32)
        lib.emu_layernorm_fwd(rows, d, C.c_float(1e-5), P(x), P(res), P(gam), P(bet), P(y), P(mean), P(rstd))
        xr = (x + res).astype(np.float64)
        mu, var = xr.mean(1, keepdims=True), xr.var(1, keepdims=True)
        xh = (xr - mu) / np.sqrt(var + 1e-5)
        np.testing.assert_allclose(y, xh * gam + bet, rtol=1e-4, atol=1e-5)
        dy = rng.normal(size=(rows, d)).astype(np.float32)
        dxr, dg, db = np.zeros_like(x), np.zeros(d, np.float32), np.zeros(d, np.float32)
        lib.emu_layernorm_bwd_workspace.restype = C.c_long
        ws = np.zeros(lib.emu_layernorm_bwd_workspace(rows, d), np.float32)
        xrf = (x + res).astype(np.float32)
        lib.emu_layernorm_bwd(rows, d, P(xrf), P(gam), P(mean), P(rstd), P(dy), P(dxr), P(dg), P(db), P(ws))
        gg = dy * gam
        ref_dx = (gg - gg.mean(1, keepdims=True) - xh * (gg * xh).mean(1, keepdims=True)) / np.sqrt(var + 1e-5)
        np.testing.assert_allclose(dxr, ref_dx, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(dg, (dy * xh).sum(0), rtol=1e-4, atol=5e-4)
        np.testing.assert_allclose(db, dy.sum(0), rtol=1e-4, atol=5e-4)
        # two incoming gradients added on load (emloco_layernorm_bwd2) = the one launch on their fp32 sum, bit for bit
        dya = rng.normal(size=(rows, d)).astype(np.float32)
        dyb = (dy - dya).astype(np.float32)
        one = [np.zeros_like(x), np.zeros(d, np.float32), np.zeros(d, np.float32)]
        two = [np.zeros_like(x), np.zeros(d, np.float32), np.zeros(d, np.float32)]
        lib.emu_layernorm_bwd(rows, d, P(xrf), P(gam), P(mean), P(rstd), P((dya + dyb).astype(np.float32)), P(one[0]), P(one[1]), P(one[2]), P(ws))
        lib.emu_layernorm_bwd2(rows, d, P(xrf), P(gam), P(mean), P(rstd), P(dya), P(dyb), P(two[0]), P(two[1]), P(two[2]), P(ws))
        for a_, b_ in zip(one, two):
            assert np.array_equal(a_, b_)


def test_column_sum_kernels():
    """Bias gradients: the 16-byte-row kernel (n % 4 == 0) and the scalar one, short (32-row partials) and tall (256-row partials) inputs
    with ragged tails, against a float64 sum."""
    lib = emu.lib()
    lib.emu_colsum_workspace.restype = C.c_long
    rng = np.random.default_rng(4)
    for m, n, scalar, x16 in ((1000, 384, 0, 0), (1000, 384, 1, 0), (33000, 128, 0, 0), (517, 100, 0, 0), (517, 69, 0, 0), (40, 260, 0, 0),
                              (1000, 384, 0, 1), (517, 384, 1, 1)):
        X = rng.normal(size=(m, n)).astype(np.float32)
        Xin = X
        if x16:                                                 # rows of bf16 in memory (the reduced-precision mode's q|k|v gradient)
            bits = (X.view(np.uint32) >> 16).astype(np.uint16)
            X = (bits.astype(np.uint32) << 16).view(np.float32)
            Xin = bits
        out = np.zeros(n, np.float32)
        ws = np.zeros(lib.emu_colsum_workspace(m, n) + 4, np.float32)
        off = (-ws.ctypes.data // 4) % 4                        # a 16-byte aligned workspace, as torch hands one out
        lib.emu_colsum(m, n, C.c_void_p(Xin.ctypes.data), P(out), C.c_void_p(ws.ctypes.data + 4 * off), scalar, x16)
        np.testing.assert_allclose(out, X.astype(np.float64).sum(0), rtol=1e-5, atol=2e-5 * np.sqrt(m), err_msg=f"{m}x{n} x16={x16}")


def test_flat_clip_adam_kernels_follow_torch():
    """emloco_adam_clip_flat's three kernels against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (CPU, single-tensor path) over
    three steps of a flat vector whose length is not a multiple of the block: parameters, moments, the clipped gradient, the norm."""
    import torch
    lib = emu.lib()
    rng = np.random.default_rng(8)
    n, lr, b1, b2, eps, wd, max_norm = 9001, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.7
    p0 = rng.normal(size=n).astype(np.float32)
    p, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    ws = np.zeros(n // 4096 + 6, np.float32)
    # the counted variant (emloco_adam_clip_flat_counted: the step count lives on the device, a captured step replays unchanged) beside it
    pc, mc, vc, wsc, count = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n // 4096 + 6, np.float32), np.zeros(1, np.float32)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([tp], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, foreach=False)
    for t in range(1, 4):
        g = (rng.normal(size=n) * (3.0 if t != 2 else 1e-3)).astype(np.float32)        # step 2: under the bound, coefficient 1
        tp.grad = torch.from_numpy(g.copy())
        norm = torch.nn.utils.clip_grad_norm_([tp], max_norm)
        opt.step()
        gk, gc = g.copy(), g.copy()
        lib.emu_adam_clip_flat(C.c_long(n), P(p), P(gk), P(m), P(v), C.c_float(lr), C.c_double(b1), C.c_double(b2), C.c_float(eps), C.c_float(wd),
                               C.c_float(1 - b1 ** t), C.c_float(np.sqrt(1 - b2 ** t)), C.c_float(max_norm), P(ws), None)
        lib.emu_adam_clip_flat(C.c_long(n), P(pc), P(gc), P(mc), P(vc), C.c_float(lr), C.c_double(b1), C.c_double(b2), C.c_float(eps), C.c_float(wd),
                               C.c_float(1.0), C.c_float(1.0), C.c_float(max_norm), P(wsc), P(count))
        np.testing.assert_allclose(ws[0], float(norm), rtol=2e-6)
        np.testing.assert_allclose(gk, tp.grad.numpy(), rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=1e-6, atol=2e-7)
        st = opt.state[tp]
        np.testing.assert_allclose(m, st["exp_avg"].numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(v, st["exp_avg_sq"].numpy(), rtol=1e-5, atol=1e-12)
        assert count[0] == t and np.array_equal(gc, gk) and np.array_equal(mc, m) and np.array_equal(vc, v)
        np.testing.assert_allclose(wsc[2:4], [1 - b1 ** t, np.sqrt(1 - b2 ** t)], rtol=1e-6)
        np.testing.assert_allclose(pc, p, rtol=0, atol=3e-7)
    assert np.abs(p - p0).max() > 1e-3


def test_locoval_kernels_match_reference_golden(golden):
    """Forward value, EmLoco loss gradient w.r.t. the predicted trajectory and all six parameter gradients."""
    g = golden("locoval")
    lib = emu.lib()
    B = 8
    traj, pose, vel = g["traj"].copy(), g["pose"].copy(), g["vel"].copy()
    w1, b1, w2, b2, w3, b3 = (g["_network_fc1_weight"], g["_network_fc1_bias"], g["_network_fc2_weight"],
                              g["_network_fc2_bias"], g["_network_fc3_weight"], g["_network_fc3_bias"])
    value, x100 = np.zeros(B, np.float32), np.zeros((B, 100), np.float32)
    h1, h2, ang = np.zeros((B, 49), np.float32), np.zeros((B, 24), np.float32), np.zeros(B, np.float32)
    lib.emu_locoval_fwd(B, P(traj), 3, P(pose), P(vel), P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), P(value), P(x100), P(h1), P(h2), P(ang))
    np.testing.assert_allclose(value, g["value"][:, 0], rtol=1e-5, atol=1e-6)
    # the reference mutates the caller's pose in place (rotation + hidden joints): our x100 carries that result
    np.testing.assert_allclose(x100[:, 26:98].reshape(B, 24, 3), g["pose_after_inplace"], rtol=1e-5, atol=1e-6)
    loss = np.mean((value - 1.0) ** 2)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=1e-5)
    dvalue = (2.0 * (value - 1.0) / B).astype(np.float32)     # d mean((v-1)^2) / dv
    dparams, dtraj = np.zeros(6174, np.float32), np.zeros_like(traj)
    ws = np.zeros(B * 6174, np.float32)
    lib.emu_locoval_bwd(B, P(traj), 3, P(pose), P(vel), P(w1), P(w2), P(w3), P(value), P(x100), P(h1), P(h2), P(ang),
                        P(dvalue), P(dparams), P(dtraj), P(ws))
    np.testing.assert_allclose(dtraj, g["grad_traj"], rtol=2e-4, atol=1e-7)
    o = 0
    for name, shape in (("fc1_weight", (49, 100)), ("fc1_bias", (49,)), ("fc2_weight", (24, 49)), ("fc2_bias", (24,)),
                        ("fc3_weight", (1, 24)), ("fc3_bias", (1,))):
        n = int(np.prod(shape))
        np.testing.assert_allclose(dparams[o:o + n].reshape(shape), g["grad__network_" + name], rtol=2e-4, atol=1e-7)
        o += n


def test_obs_normalize_kernel_matches_reference_golden(golden):
    """RunningMeanStd eval-mode normalisation with the self | task column split (policy input, row A19)."""
    g = golden("policy_net")
    lib = emu.lib()
    obs = g["obs"].astype(np.float32)
    rows, cols = obs.shape
    mean, var = g["running_mean"].astype(np.float32), g["running_var"].astype(np.float32)
    split, ld1 = 368, 1056
    out0 = np.zeros((rows, 400), np.float32)
    out1 = np.zeros((rows, ld1), np.float32)
    lib.emu_obs_normalize(rows, cols, P(obs), cols, P(mean), P(var), C.c_float(float(g["epsilon"])), C.c_float(5.0), split,
                          P(out0), 400, P(out1), ld1)
    ref = g["norm_obs"]
    np.testing.assert_allclose(out0[:, :split], ref[:, :split], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(out1[:, :cols - split], ref[:, split:], rtol=2e-6, atol=2e-6)
    assert np.all(out1[:, cols - split:] == 0) and ref.max() == 5.0 and ref.min() == -5.0


def test_rms_update_kernel_matches_the_reference_class(golden):
    """rms_update_kernel (RunningMeanStd's training-mode statistics update, one launch) against the moments the reference's own
    class reached over four batches (tests/golden/gen_golden_rms.py), freeze_partial(3) from the third batch on: float64 moments
    to 2e-6 (the reference takes the batch mean / variance in float32 before merging; the kernel accumulates in float64), the count
    exactly."""
    g = golden("running_mean_std")
    lib = emu.lib()
    cols = g["x0"].shape[1]
    mean, var = np.zeros(cols, np.float64), np.ones(cols, np.float64)
    cnt, cnt2 = np.ones(1, np.float64), np.zeros(1, np.float64)
    for i in range(int(g["n_batches"])):
        x = np.ascontiguousarray(g[f"x{i}"], np.float32)
        first = cols - 3 if i >= 2 else 0
        lib.emu_rms_update(x.shape[0], cols, P(x), cols, mean.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p),
                           cnt.ctypes.data_as(C.c_void_p), cnt2.ctypes.data_as(C.c_void_p), first)
        cnt[0] = cnt2[0]
        np.testing.assert_allclose(mean, g[f"mean{i}"], rtol=2e-6, atol=2e-6, err_msg=f"batch {i}")
        np.testing.assert_allclose(var, g[f"var{i}"], rtol=2e-6, atol=2e-6, err_msg=f"batch {i}")
        assert cnt[0] == float(g[f"count{i}"])


def test_split_mode_gemm_with_non_finite_and_huge_operands():
    """The documented behaviour of EMLOCO_GEMM_SPLIT for operands it cannot cut into bf16 pieces (include/emloco_predictor.h): an
    Inf, a NaN or a finite value above bf16's largest finite makes exactly the output elements whose reduction it enters NaN (the
    plain fp32 mode gives +-Inf for the lone Inf); every other element equals the clean product bit for bit."""
    rng = np.random.default_rng(8)
    SPLIT = 1024
    m, n, k = 72, 136, 48
    A = rng.normal(size=(1, m, k)).astype(np.float32)
    B = rng.normal(size=(1, n, k)).astype(np.float32)
    clean = _gemm(A, B, 0, 0, m, n, k, flags=SPLIT)[0]
    Ab, Bb = A.copy(), B.copy()
    Ab[0, 5, 7] = np.inf                      # row 5 of the output
    Ab[0, 40, 0] = np.float32(3.40e38)        # finite in fp32, overflows bf16: row 40
    Bb[0, 100, 3] = np.nan                    # column 100
    got = _gemm(Ab, Bb, 0, 0, m, n, k, flags=SPLIT)[0]
    bad = np.zeros((m, n), bool)
    bad[5, :] = bad[40, :] = bad[:, 100] = True
    assert np.isnan(got[bad]).all()
    assert np.array_equal(got[~bad], clean[~bad])
    plain = _gemm(Ab, Bb, 0, 0, m, n, k)[0]
    assert np.isinf(plain[5, :100]).all() and not np.isnan(plain[40, :100]).any() and np.isnan(plain[:, 100]).all()


def test_chunked_rms_update_equals_the_single_launch():
    """rms_partial_kernel + rms_merge_kernel (the learner's tall minibatches: per-chunk moments from registers, in-order fold) against
    rms_update_kernel and numpy float64 on a ragged tall batch (1 100 rows: four full 256-row chunks and a 76-row one whose last
    wave is empty; 70 columns: a partial column group), with freeze_partial columns: moments to 1e-12, count exact."""
    lib = emu.lib()
    rng = np.random.default_rng(4)
    rows, cols = 1100, 70
    x = (rng.normal(size=(rows, cols)) * rng.uniform(0.1, 30, size=cols) + rng.uniform(-50, 50, size=cols)).astype(np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    res = []
    for chunked in (False, True):
        mean, var = rng.normal(size=cols) * 0 + np.linspace(-1, 1, cols), np.linspace(0.5, 2, cols)
        cnt, cnt2 = np.array([37.0]), np.zeros(1)
        if chunked:
            ws = np.zeros(((rows + 255) // 256) * 3 * cols, np.float64)
            lib.emu_rms_update_chunked(rows, cols, P(x), cols, vp(mean), vp(var), vp(cnt), vp(cnt2), 5, vp(ws))
        else:
            lib.emu_rms_update(rows, cols, P(x), cols, vp(mean), vp(var), vp(cnt), vp(cnt2), 5)
        res.append((mean.copy(), var.copy(), cnt2[0]))
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(res[1][1], res[0][1], rtol=1e-12, atol=1e-12)
    assert res[0][2] == res[1][2] == 37.0 + rows
    x64 = x.astype(np.float64)
    m0, v0 = np.linspace(-1, 1, cols), np.linspace(0.5, 2, cols)
    bm, bv = x64.mean(0), x64.var(0, ddof=1)
    d, tot = bm - m0, 37.0 + rows
    em = m0 + d * rows / tot
    ev = (v0 * 37.0 + bv * rows + d * d * 37.0 * rows / tot) / tot
    np.testing.assert_allclose(res[1][0][5:], em[5:], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(res[1][1][5:], ev[5:], rtol=1e-11, atol=1e-11)
    np.testing.assert_array_equal(res[1][0][:5], m0[:5])                  # frozen columns keep their moments


@pytest.fixture(params=[0, 2], ids=["fp32", "split"])
def attn_precision(request):
    """the fused attention on the fp32 matrix instruction (0) and with fp32-class tile products from bf16 pieces (2, EMLOCO_ATTN_SPLIT:
    the product's default) -- the same tolerances against float64 in both"""
    lib = emu.lib()
    lib.emu_attention_set_precision(request.param)
    yield request.param
    lib.emu_attention_set_precision(0)


def test_fused_attention_kernels_forward_and_backward(attn_precision):
    """attn_fwd / attn_bwd_dq / attn_bwd_dkv (head dim 32) against a float64 numpy attention, ragged S (two query blocks'
    worth of tiles would be slow in the emulator: S = 70 covers a partial last tile), additive and -inf key biases."""
    lib = emu.lib()
    rng = np.random.default_rng(5)
    n_seq, S, H = 2, 70, 2
    d = H * 32
    qkv = (rng.normal(size=(n_seq, S, 3 * d)) * 0.7).astype(np.float32)
    kb = np.zeros((n_seq, S), np.float32)
    kb[0, 5::7] = 1.0                     # the reference's float padding mask: an additive +1
    kb[1, 50:] = -np.inf                  # bool-style mask
    scale = 1.0 / np.sqrt(32.0)
    out = np.zeros((n_seq, S, d), np.float32)
    lse = np.zeros((n_seq * H, S), np.float32)
    lib.emu_attention_fwd(n_seq, S, H, d, C.c_float(scale), P(qkv), P(kb), P(out), P(lse))
    q = qkv[..., :d].reshape(n_seq, S, H, 32).transpose(0, 2, 1, 3).astype(np.float64)
    k = qkv[..., d:2 * d].reshape(n_seq, S, H, 32).transpose(0, 2, 1, 3).astype(np.float64)
    v = qkv[..., 2 * d:].reshape(n_seq, S, H, 32).transpose(0, 2, 1, 3).astype(np.float64)
    s = np.einsum("bhqd,bhkd->bhqk", q, k) * scale + kb[:, None, None, :].astype(np.float64)
    mx = s.max(-1, keepdims=True)
    e = np.exp(s - mx)
    p = e / e.sum(-1, keepdims=True)
    ref = np.einsum("bhqk,bhkd->bhqd", p, v).transpose(0, 2, 1, 3).reshape(n_seq, S, d)
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(lse.reshape(n_seq, H, S), (mx[..., 0] + np.log(e.sum(-1))), rtol=1e-5, atol=1e-5)
    dout = rng.normal(size=out.shape).astype(np.float32)
    dqkv = np.zeros_like(qkv)
    dsum = np.zeros((n_seq * H, S), np.float32)
    lib.emu_attention_bwd(n_seq, S, H, d, C.c_float(scale), P(qkv), P(kb), P(out), P(lse), P(dout), P(dqkv), P(dsum))
    do = dout.reshape(n_seq, S, H, 32).transpose(0, 2, 1, 3).astype(np.float64)
    dv = np.einsum("bhqk,bhqd->bhkd", p, do)
    dp = np.einsum("bhqd,bhkd->bhqk", do, v)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) * scale
    dq = np.einsum("bhqk,bhkd->bhqd", ds, k)
    dk = np.einsum("bhqk,bhqd->bhkd", ds, q)
    back = lambda t: t.transpose(0, 2, 1, 3).reshape(n_seq, S, d)
    np.testing.assert_allclose(dqkv[..., :d], back(dq), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(dqkv[..., d:2 * d], back(dk), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(dqkv[..., 2 * d:], back(dv), rtol=1e-4, atol=5e-6)
    # a fully masked sequence gives zeros, not NaN ("safe softmax", torch >= 2.5)
    kb2 = np.full((1, S), -np.inf, np.float32)
    out2 = np.ones((1, S, d), np.float32)
    lib.emu_attention_fwd(1, S, H, d, C.c_float(scale), P(qkv[:1]), P(kb2), P(out2), P(lse[:H]))
    assert np.all(out2 == 0)


def test_fused_attention_kernels_with_dropout_on_the_probabilities(attn_precision):
    """nn.MultiheadAttention(dropout = p) in training mode: softmax -> dropout -> . V.  The fused kernels with the counter-based
    keep mask against a float64 attention that applies the SAME mask (the hash is evaluated on the host through the emulator
    library): forward and all three gradients; the keep rate is 1 - p; p = 0 reproduces the plain kernels."""
    lib = emu.lib()
    lib.emu_attention_set_dropout.argtypes = [C.c_float, C.c_uint]
    lib.emu_attn_keep.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_float]
    rng = np.random.default_rng(8)
    n_seq, S, H = 2, 70, 2
    d = H * 32
    pdrop, seed = 0.1, 0xABCDEF
    qkv = (rng.normal(size=(n_seq, S, 3 * d)) * 0.7).astype(np.float32)
    kb = np.zeros((n_seq, S), np.float32)
    kb[1, 60:] = -np.inf
    scale = 1.0 / np.sqrt(32.0)
    out, lse = np.zeros((n_seq, S, d), np.float32), np.zeros((n_seq * H, S), np.float32)
    dout = rng.normal(size=out.shape).astype(np.float32)
    dqkv, dsum = np.zeros_like(qkv), np.zeros((n_seq * H, S), np.float32)
    try:
        lib.emu_attention_set_dropout(pdrop, seed)
        lib.emu_attention_fwd(n_seq, S, H, d, C.c_float(scale), P(qkv), P(kb), P(out), P(lse))
        lib.emu_attention_bwd(n_seq, S, H, d, C.c_float(scale), P(qkv), P(kb), P(out), P(lse), P(dout), P(dqkv), P(dsum))
    finally:
        lib.emu_attention_set_dropout(0.0, 0)
    keep = np.array([[[lib.emu_attn_keep(seed, bh, i, j, pdrop) for j in range(S)] for i in range(S)] for bh in range(n_seq * H)],
                    np.float64).reshape(n_seq, H, S, S)
    assert abs(keep.mean() - (1 - ATTN_P8(pdrop))) < 0.01
    M = keep / (1 - ATTN_P8(pdrop))                        # (the mask realises p in 1/256ths: at_drop_thr8)
    q = qkv[..., :d].reshape(n_seq, S, H, 32).transpose(0, 2, 1, 3).astype(np.float64)
    k = qkv[..., d:2 * d].reshape(n_seq, S, H, 32).transpose(0, 2, 1, 3).astype(np.float64)
    v = qkv[..., 2 * d:].reshape(n_seq, S, H, 32).transpose(0, 2, 1, 3).astype(np.float64)
    s = np.einsum("bhqd,bhkd->bhqk", q, k) * scale + kb[:, None, None, :].astype(np.float64)
    e = np.exp(s - s.max(-1, keepdims=True))
    p = e / e.sum(-1, keepdims=True)
    pm = p * M
    back = lambda t: t.transpose(0, 2, 1, 3).reshape(n_seq, S, d)
    np.testing.assert_allclose(out, back(np.einsum("bhqk,bhkd->bhqd", pm, v)), rtol=1e-5, atol=3e-6)
    do = dout.reshape(n_seq, S, H, 32).transpose(0, 2, 1, 3).astype(np.float64)
    dv = np.einsum("bhqk,bhqd->bhkd", pm, do)
    dp = np.einsum("bhqd,bhkd->bhqk", do, v) * M                      # gradient w.r.t. the undropped probabilities
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) * scale
    np.testing.assert_allclose(dqkv[..., :d], back(np.einsum("bhqk,bhkd->bhqd", ds, k)), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(dqkv[..., d:2 * d], back(np.einsum("bhqk,bhqd->bhkd", ds, q)), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(dqkv[..., 2 * d:], back(dv), rtol=1e-4, atol=5e-6)
    assert np.abs(out - back(np.einsum("bhqk,bhkd->bhqd", p, v))).max() > 1e-2      # ... and it is not the undropped attention


def test_fused_attention_kernels_with_live_query_rows():
    """emloco_attention_{fwd,bwd}_queries: only the first n_query rows attend (the last layer of a former feeds 21 tokens on,
    model_jta.py:316).  Forward rows and log-sum-exp are BIT-equal to the first rows of the full launch (with and without
    dropout: the mask is indexed by (head-sequence, query, key)); the gradients equal the full backward fed with zeros on the
    dropped rows, and dQ of the rows that did not attend is exactly zero."""
    lib = emu.lib()
    lib.emu_attention_set_dropout.argtypes = [C.c_float, C.c_uint]
    rng = np.random.default_rng(9)
    n_seq, S, H = 2, 70, 2
    d = H * 32
    qkv = (rng.normal(size=(n_seq, S, 3 * d)) * 0.7).astype(np.float32)
    kb = np.zeros((n_seq, S), np.float32)
    kb[0, 3::5] = 1.0
    kb[1, 60:] = -np.inf
    scale = C.c_float(1.0 / np.sqrt(32.0))
    for Sq, pdrop in ((21, 0.0), (21, 0.1), (40, 0.1)):
        try:
            lib.emu_attention_set_dropout(pdrop, 77)
            out, lse = np.zeros((n_seq, S, d), np.float32), np.zeros((n_seq * H, S), np.float32)
            lib.emu_attention_fwd(n_seq, S, H, d, scale, P(qkv), P(kb), P(out), P(lse))
            outq, lseq = np.zeros((n_seq, Sq, d), np.float32), np.zeros((n_seq * H, Sq), np.float32)
            lib.emu_attention_fwd_queries(n_seq, S, Sq, H, d, scale, P(qkv), P(kb), P(outq), P(lseq))
            assert np.array_equal(outq, out[:, :Sq]) and np.array_equal(lseq, lse[:, :Sq])
            doutq = rng.normal(size=outq.shape).astype(np.float32)
            dout = np.zeros_like(out)
            dout[:, :Sq] = doutq
            dqkv, dsum = np.zeros_like(qkv), np.zeros((n_seq * H, S), np.float32)
            lib.emu_attention_bwd(n_seq, S, H, d, scale, P(qkv), P(kb), P(out), P(lse), P(dout), P(dqkv), P(dsum))
            dqkvq, dsumq = np.full_like(qkv, 7.0), np.zeros((n_seq * H, Sq), np.float32)
            lib.emu_attention_bwd_queries(n_seq, S, Sq, H, d, scale, P(qkv), P(kb), P(outq), P(lseq), P(doutq), P(dqkvq), P(dsumq))
        finally:
            lib.emu_attention_set_dropout(0.0, 0)
        assert np.array_equal(dqkvq[:, :Sq, :d], dqkv[:, :Sq, :d])
        assert np.all(dqkvq[:, Sq:, :d] == 0)
        # dK / dV sum over fewer query tiles: the same terms, zeros dropped -> equal up to the association of the tile sums
        np.testing.assert_allclose(dqkvq[..., d:], dqkv[..., d:], rtol=1e-5, atol=1e-6)


def test_fused_attention_kernels_bf16_operands():
    """EMLOCO_ATTN_BF16: the same three kernels with the tile products on v_mfma_f32_32x32x16_bf16 (operands rounded to bf16,
    fp32 accumulation, fp32 softmax statistics): within 2e-2 of the float64 attention (the stated bar of the reduced
    precision mode), visibly different from the fp32 kernels, masks / ragged tiles / safe softmax unchanged."""
    lib = emu.lib()
    rng = np.random.default_rng(6)
    n_seq, S, H = 2, 70, 2
    d = H * 32
    qkv = (rng.normal(size=(n_seq, S, 3 * d)) * 0.7).astype(np.float32)
    kb = np.zeros((n_seq, S), np.float32)
    kb[0, 5::7] = 1.0
    kb[1, 50:] = -np.inf
    scale = 1.0 / np.sqrt(32.0)
    dout = rng.normal(size=(n_seq, S, d)).astype(np.float32)
    res = {}
    try:
        for prec in (0, 1):
            lib.emu_attention_set_precision(prec)
            out = np.zeros((n_seq, S, d), np.float32)
            lse = np.zeros((n_seq * H, S), np.float32)
            lib.emu_attention_fwd(n_seq, S, H, d, C.c_float(scale), P(qkv), P(kb), P(out), P(lse))
            dqkv = np.zeros_like(qkv)
            dsum = np.zeros((n_seq * H, S), np.float32)
            lib.emu_attention_bwd(n_seq, S, H, d, C.c_float(scale), P(qkv), P(kb), P(out), P(lse), P(dout), P(dqkv), P(dsum))
            res[prec] = (out, lse, dqkv)
        kb2 = np.full((1, S), -np.inf, np.float32)
        out2 = np.ones((1, S, d), np.float32)
        lse2 = np.zeros((H, S), np.float32)
        lib.emu_attention_fwd(1, S, H, d, C.c_float(scale), P(qkv[:1]), P(kb2), P(out2), P(lse2))
        assert np.all(out2 == 0)
    finally:
        lib.emu_attention_set_precision(0)
    for a, b, what in zip(res[0], res[1], ("out", "lse", "dqkv")):
        err = np.abs(a - b).max()
        assert np.isfinite(b).all() and 1e-5 < err < 2e-2 * np.abs(a).max(), (what, err)
    assert np.all(res[1][2][1, 50:, d:] == 0)          # masked keys receive no gradient (dk, dv rows of the -inf keys)


def test_compact_flags_kernel_is_nonzero():
    """device-side `reset_buf.nonzero()`: ascending ids, -1 padding, count in the extra slot; ragged sizes around 1024"""
    lib = emu.lib()
    rng = np.random.default_rng(2)
    for n, prob in ((4096, 0.01), (1500, 0.5), (1024, 1.0), (77, 0.0), (2049, 0.3)):
        flags = (rng.random(n) < prob).astype(np.int64) * rng.integers(1, 5, n)
        ids = np.full(n + 1, 12345, np.int32)
        lib.emu_compact_flags(P(flags), n, P(ids))
        nz = np.nonzero(flags)[0]
        assert ids[n] == len(nz)
        assert np.array_equal(ids[:len(nz)], nz) and np.all(ids[len(nz):n] == -1)


def test_compact_order_kernel_compacts_and_sorts_in_one_launch():
    """compact_order_kernel (the fused chain's second launch): workgroup 0 = `reset_buf.nonzero()` + a snapshot of the flags, workgroup
    1 = the next rigid-body launch's dispatch order -- env ids by descending contact work (bucket = min(key, 127); the order inside a
    bucket is free).  Sizes around the 1024-thread chunk and beyond the 16 384 keys the sort keeps in LDS."""
    lib = emu.lib()
    rng = np.random.default_rng(5)
    for n, prob in ((4096, 0.04), (1500, 0.5), (77, 0.0), (20000, 0.01)):
        flags = (rng.random(n) < prob).astype(np.int64) * rng.integers(1, 5, n)
        ids = np.full(n + 1, 12345, np.int32)
        snap = np.full(n, -7, np.int64)
        ticks = rng.integers(0, 300, n).astype(np.uint32)
        order = np.full(n, -1, np.int32)
        ws = np.zeros(n, np.uint8)
        lib.emu_compact_order(P(flags), n, P(ids), P(snap), P(ticks), n, P(order), P(ws))
        nz = np.nonzero(flags)[0]
        assert ids[n] == len(nz) and np.array_equal(ids[:len(nz)], nz) and np.all(ids[len(nz):n] == -1)
        assert np.array_equal(snap, flags)
        assert np.array_equal(np.sort(order), np.arange(n))                       # a permutation of the envs
        b = np.minimum(ticks[order], 127)
        assert np.all(np.diff(b.astype(np.int64)) <= 0)                           # most work first
    # without keys (cost order off) the launch is the compaction alone
    flags = (rng.random(300) < 0.2).astype(np.int64)
    ids = np.zeros(301, np.int32)
    lib.emu_compact_order(P(flags), 300, P(ids), None, None, 0, None, None)
    assert ids[300] == flags.sum()


@pytest.mark.parametrize("name", ["traj_reset_plain", "traj_reset_heading", "traj_reset_real1", "traj_reset_real2", "traj_reset_real2_noadj"])
def test_traj_reset_kernel_matches_reference_golden(golden, name):
    """TrajGenerator.reset as the device computes it (reset_kernels.hip: reset_trajectory, incl. the real-path branch the
    headline bench runs) on the reference's own draws: vertices within 1e-4 m of the reference's, inversion mask bit-exact."""
    from helpers import TRAJ_CASES, traj_real_pick, traj_reset_bufs, traj_rnd_rows
    g = golden(name)
    E = 16
    verts = np.zeros((E, 101, 3), np.float32)
    inverted = np.full(E, 7, np.uint8)
    table = np.ascontiguousarray(g["real_table"].astype(np.float32)) if "real_table" in g else None
    pick = traj_real_pick(g) if table is not None else None
    b = traj_reset_bufs(TRAJ_CASES[name], g, verts.ctypes.data, inverted.ctypes.data, None if table is None else table.ctypes.data,
                        None if pick is None else pick.ctypes.data)
    ids = np.arange(E, dtype=np.int32)
    rnd = traj_rnd_rows(g)
    ip, rv = np.ascontiguousarray(g["init_pos"], np.float32), np.ascontiguousarray(g["root_vel"], np.float32)
    emu.lib().emu_task_traj_reset(C.byref(b), P(ids), E, P(rnd), P(ip), P(rv))
    assert np.abs(verts - g["verts"]).max() < 1e-4, np.abs(verts - g["verts"]).max()
    if "inverted" in g:
        np.testing.assert_array_equal(inverted, g["inverted"])


def test_traj_reset_kernel_real_rows_are_sampled_without_replacement(golden):
    """Without explicit rows the kernel takes P_key(list position): distinct within the call (random.sample semantics,
    traj_generator.py:132), equal to the host restatement, different per key."""
    from helpers import traj_reset_bufs, traj_rnd_rows
    from emloco_amd._lib import real_pick_perm
    g = dict(golden("traj_reset_real2_noadj"))
    E = 32
    n_real = 40
    table = np.zeros((n_real, 101, 3), np.float32)
    table[:, :, 0] = np.linspace(0, 3, 101)[None]
    table[:, :, 2] = np.arange(n_real)[:, None]                       # the row id rides in z, which the reset copies through
    g["real_table"] = table
    rnd = np.full((E, 512), 0.9, np.float32)                           # every env takes a real path
    ids = np.arange(E, dtype=np.int32)[::-1].copy()
    ip = np.zeros((E, 3), np.float32)
    rv = np.ones((E, 3), np.float32)
    seen = {}
    for key in (11, 12):
        verts = np.zeros((E, 101, 3), np.float32)
        inverted = np.zeros(E, np.uint8)
        b = traj_reset_bufs(dict(real_path=True), g, verts.ctypes.data, inverted.ctypes.data, table.ctypes.data, None, key=key)
        emu.lib().emu_task_traj_reset(C.byref(b), P(ids), E, P(rnd), P(ip), P(rv))
        rows = verts[ids, 0, 2].astype(int)                            # rows[i] = what list entry i took
        assert len(set(rows.tolist())) == E
        assert rows.tolist() == [real_pick_perm(i, n_real, key) for i in range(E)]
        seen[key] = rows
    assert not np.array_equal(seen[11], seen[12])


def test_get_heights_kernel_indices_match_reference_golden(golden):
    """The terrain probes as the kernels compute them (task_kernels.hip: grid_probe / center_probe / map_index, shared with the
    fused post-physics kernel): int64 map indices and heights array_equal to the reference's over 256 x (1024 + 9) probes."""
    from helpers import terrain_index_map
    g = golden("terrain_index")
    hf = terrain_index_map()
    E = 256
    fn = emu.lib().emu_task_get_heights
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    pose = np.ascontiguousarray(g["head_pose"], np.float32)
    h, px, py = np.zeros((E, 1024), np.float32), np.zeros((E, 1024), np.int64), np.zeros((E, 1024), np.int64)
    fn(hf.ctypes.data, 1080, 1080, 0.1, 0.005, pose.ctypes.data, E, 1, h.ctypes.data, px.ctypes.data, py.ctypes.data)
    np.testing.assert_array_equal(px, g["px"])
    np.testing.assert_array_equal(py, g["py"])
    np.testing.assert_array_equal(np.round(h / 0.005).astype(np.int16), g["heights_raw"])
    root7 = np.ascontiguousarray(g["root_states"][:, :7], np.float32)
    c, cx, cy = np.zeros((E, 9), np.float32), np.zeros((E, 9), np.int64), np.zeros((E, 9), np.int64)
    fn(hf.ctypes.data, 1080, 1080, 0.1, 0.005, root7.ctypes.data, E, 0, c.ctypes.data, cx.ctypes.data, cy.ctypes.data)
    np.testing.assert_array_equal(cx, g["cpx"])
    np.testing.assert_array_equal(cy, g["cpy"])
    np.testing.assert_array_equal(np.round(c / 0.005).astype(np.int16), g["center_raw"])


def test_fused_kernel_height_observations_equal_the_reference_golden(golden):
    """The fused post-physics kernel's 1024 height observations on the terrain fixture's own poses: array_equal to the
    observations torch produced (humanoid_pedestrain_terrain.py:427-437), no tolerated cell flips."""
    from emloco_amd import _lib as L
    gt = golden("terrain_heights")
    E = 16
    th = emu.TaskHost(E, gt["heightfield"])
    th.rb_state[:, 0, :7] = gt["root_states"][:, :7]
    th.rb_state[:, 13, :7] = gt["head_pose"]
    th.post_physics(L.POST_OBS)
    np.testing.assert_array_equal(th.obs[:, 398:], gt["height_obs"])


def test_sim_step_kernel_at_the_edge_of_the_heightfield_and_on_stairs_is_bit_exact_vs_oracle():
    """The terrain probes one radius out (four per sphere) where they leave the map (the border cell's plane extends: clamped cell
    indices) and where neighbouring triangles differ most (a staircase of one-cell risers): emulated kernel == oracle bytes."""
    E = 3
    models = varied_models(E, seed=17)
    root, dof, tgt = scene_state(E, seed=18)
    n = 200
    steps = ((np.arange(n)[:, None] // 3) % 4 * 30 + 0 * np.arange(n)[None, :]).astype(np.int16)      # 15 cm risers every 30 cm, 4 high, repeating
    hf = dict(samples=steps, horizontal_scale=0.1, vertical_scale=0.005)
    root = root.copy()
    root[:, 0] = [0.04, 9.93, 19.88]                      # over the first cell, mid map, over the last cell (the map is 19.9 m wide)
    root[:, 1] = [0.03, 10.0, 19.86]
    root[:, 2] = 1.05 + steps[np.clip((root[:, 0] / 0.1).astype(int), 0, n - 1), 0] * 0.005
    a = oracle_sim(models, root, dof, tgt, heightfield=hf, n_sub=4)
    b = oracle_sim(models, root, dof, tgt, heightfield=hf, n_sub=4)
    for _ in range(6):
        a.step(1)
        emu.sim_step(b, 1)
        for name in ("root_state", "dof_state", "rb_state", "contact_force", "dof_force", "lambda_ws"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert np.isfinite(a.rb_state).all() and np.abs(a.contact_force).max() > 50


def test_sim_step_kernel_on_the_slope_corrected_mesh_is_bit_exact_vs_oracle():
    """Round 5: collision with the slope-corrected terrain mesh (vertical risers: emloco_sim_set_ground_mesh_moves) -- the mesh
    triangle covering a point among the 3 x 3 cells' moved triangles, the probes on it, the closest-point test against the collapsed
    cells' vertical faces (faces looking along x AND along y: both tangent-frame branches).  Humanoids dropped across risers, into the
    trench and over the map's border: emulated kernel == oracle bytes, and the result differs from the uncorrected field's."""
    E = 4
    models = varied_models(E, seed=27)
    root, dof, tgt = scene_state(E, seed=28)
    from helpers import corrected_stairs
    steps, hf = corrected_stairs()
    root = root.copy()
    root[:, 0] = [0.04, 9.93, 10.21, 19.88]
    root[:, 1] = [0.03, 10.0, 9.62, 19.86]
    root[:, 2] = 0.98 + steps[np.clip((root[:, 0] / 0.1).astype(int), 0, 199), np.clip((root[:, 1] / 0.1).astype(int), 0, 199)] * 0.005
    root[:, 7] = [0.5, 1.5, -1.0, 0.0]
    root[:, 8] = [0.0, -1.0, 1.5, 0.0]
    a = oracle_sim(models, root, dof, tgt, heightfield=hf, n_sub=4)
    b = oracle_sim(models, root, dof, tgt, heightfield=hf, n_sub=4)
    plain = oracle_sim(models, root, dof, tgt, heightfield={k: v for k, v in hf.items() if not k.startswith("move")}, n_sub=4)
    for _ in range(10):
        a.step(1)
        emu.sim_step(b, 1)
        plain.step(1)
        for name in ("root_state", "dof_state", "rb_state", "contact_force", "dof_force", "lambda_ws"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert np.isfinite(a.rb_state).all() and np.abs(a.contact_force).max() > 50
    assert np.abs(a.contact_force[:, :, :2]).max() > 20                      # something pushed sideways: a riser or an edge
    assert not np.array_equal(a.rb_state, plain.rb_state)


def _bf16_bits(x):
    """fp32 -> bf16 bit patterns (uint16), round to nearest even"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) & 0xffff).astype(np.uint16)


def _bf16_val(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


def test_chained_feed_forward_kernels():
    """csrc/ffn_kernels.hip (round 5): the feed-forward block as two matrix products chained through registers -- the forward
    (hidden = dropout(relu(x W1^T + b1)) stored as bf16, out = dropout(hidden W2^T + b2)) and the input-gradient pass (dz1 = (dz2 W2) o
    [hidden > 0] / (1 - p) stored as bf16, dx = dz1 W1) -- emulated lane for lane against numpy on the bf16-rounded operands.  The first
    product is checked through the stored tile (one bf16 rounding apart at most, exact zeros where the mask says so), the second against
    the product of the tile the KERNEL stored (fp32 accumulation error only).  Ragged row count (the last workgroup is partial), both
    dropout settings; the hidden mask is the documented pair hash, the output mask the GEMM epilogue's counter hash."""
    lib = emu.lib()
    lib.emu_ffn_keep.restype = C.c_int
    lib.emu_drop_keep.restype = C.c_int
    rng = np.random.default_rng(11)
    M, F, D = 300, 128, 128
    x = rng.normal(size=(M, D)).astype(np.float32)
    W1 = (rng.normal(size=(F, D)) / np.sqrt(D)).astype(np.float32)
    W2 = (rng.normal(size=(D, F)) / np.sqrt(F)).astype(np.float32)
    b1 = (rng.normal(size=F) * 0.3).astype(np.float32)
    b2 = (rng.normal(size=D) * 0.3).astype(np.float32)
    W1b, W2b = _bf16_bits(W1), _bf16_bits(W2)
    xb = _bf16_val(_bf16_bits(x)).astype(np.float64)
    pre = xb @ _bf16_val(W1b).astype(np.float64).T + b1
    U16 = lambda a: a.ctypes.data_as(C.POINTER(C.c_ushort))
    for p, s1, s2 in ((0.0, 0, 0), (0.1, 12345, 777)):
        h = np.full((M, F), 0x7fc0, np.uint16)                    # NaN patterns: every element must be written
        out = np.full((M, D), np.nan, np.float32)
        mbits = np.full((M, F // 32), 0xdeadbeef, np.uint32)
        lib.emu_ffn_chain(0, M, F, P(x), U16(W1b), U16(W2b), P(b1), P(b2), U16(h), None, mbits.ctypes.data_as(C.POINTER(C.c_uint)), P(out),
                          C.c_float(p), C.c_uint(s1), C.c_uint(s2))
        keep = np.ones((M, F), bool)
        if p > 0:
            keep = np.array([[lib.emu_ffn_keep(C.c_uint(s1), r, f, C.c_float(p)) for f in range(F)] for r in range(M)], bool)
            assert 0.85 < keep.mean() < 0.95
        want = np.where(keep, np.maximum(pre, 0.0) / (1.0 - p), 0.0)
        got = _bf16_val(h).astype(np.float64)
        assert not np.isnan(got).any()
        # the mask bits the input-gradient pass reads: word [row][2 c + hi], bit 16 t + 4 q + e = unit 64 c + 32 t + 8 q + 4 hi + e
        unit = np.arange(F)
        cc, tt, qq, hh, ee = unit // 64, (unit % 64) // 32, (unit % 32) // 8, (unit % 8) // 4, unit % 4
        bits = (mbits[:, 2 * cc + hh] >> (16 * tt + 4 * qq + ee).astype(np.uint32)) & 1
        assert np.array_equal(bits.astype(bool), got > 0)
        assert ((got == 0) == (want <= 0)).mean() > 0.999            # (a pre-activation within rounding of zero may fall either side)
        np.testing.assert_allclose(got, want, rtol=2.0 ** -7, atol=2e-5)
        assert np.mean(got == _bf16_val(_bf16_bits(want.astype(np.float32)))) > 0.98      # the same bf16 value almost everywhere
        f = got @ _bf16_val(W2b).astype(np.float64).T + b2
        if p > 0:
            k2 = np.array([[lib.emu_drop_keep(C.c_uint(s2), C.c_ulonglong(r * D + c), C.c_float(p)) for c in range(D)] for r in range(M)], bool)
            f = np.where(k2, f / (1.0 - p), 0.0)
            assert 0.85 < k2.mean() < 0.95
        np.testing.assert_allclose(out, f, rtol=1e-5, atol=2e-5)
        # (round 6) the same launch with the post-norm layer's tail in its epilogue (emloco_ffn_fwd_norm): xr = f + res, y = LayerNorm(xr)
        res_ = rng.normal(size=(M, D)).astype(np.float32)
        gam, bet = (1.0 + 0.1 * rng.normal(size=D)).astype(np.float32), (0.1 * rng.normal(size=D)).astype(np.float32)
        y_n, xr_n = np.full((M, D), np.nan, np.float32), np.full((M, D), np.nan, np.float32)
        mean_n, rstd_n = np.full(M, np.nan, np.float32), np.full(M, np.nan, np.float32)
        h_n, mb_n = np.full((M, F), 0x7fc0, np.uint16), np.full((M, F // 32), 0xdeadbeef, np.uint32)
        lib.emu_ffn_set_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.emu_ffn_set_norm(P(res_), P(gam), P(bet), C.c_float(1e-5), P(xr_n), P(mean_n), P(rstd_n))
        lib.emu_ffn_chain(0, M, F, P(x), U16(W1b), U16(W2b), P(b1), P(b2), U16(h_n), None, mb_n.ctypes.data_as(C.POINTER(C.c_uint)), P(y_n),
                          C.c_float(p), C.c_uint(s1), C.c_uint(s2))
        lib.emu_ffn_set_norm(None, None, None, C.c_float(0.0), None, None, None)
        assert np.array_equal(h_n, h) and np.array_equal(mb_n, mbits)                       # the block itself is untouched
        xr_w = out.astype(np.float64) + res_                                                # `out`: the unfused launch's f
        np.testing.assert_allclose(xr_n, xr_w, rtol=1e-6, atol=1e-6)
        mu_w = xr_n.astype(np.float64).mean(1)
        rs_w = 1.0 / np.sqrt(xr_n.astype(np.float64).var(1) + 1e-5)
        np.testing.assert_allclose(mean_n, mu_w, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rstd_n, rs_w, rtol=1e-5)
        np.testing.assert_allclose(y_n, (xr_n - mu_w[:, None]) * rs_w[:, None] * gam + bet, rtol=1e-5, atol=1e-5)
        # input gradient: dz2 stands for the gradient w.r.t. linear2's output; P = W2^T [F][128], Q = W1^T [128][F]
        dz2 = rng.normal(size=(M, D)).astype(np.float32)
        W2T, W1T = np.ascontiguousarray(W2b.T), np.ascontiguousarray(W1b.T)
        dz1 = np.full((M, F), 0x7fc0, np.uint16)
        dx = np.full((M, D), np.nan, np.float32)
        # (round 6) the launch also leaves the column sums of dz1 per wave of 32 rows: linear1's bias gradient without a pass over dz1
        n_part = ((M + 255) // 256) * 8
        colpart = np.full((n_part, F), np.nan, np.float32)
        colpart[(M + 31) // 32:] = 0.0                           # (rows of waves wholly past the end: the launcher zeroes them)
        lib.emu_ffn_set_colpart(P(colpart))
        lib.emu_ffn_chain(1, M, F, P(dz2), U16(W2T), U16(W1T), None, None, None, U16(dz1), mbits.ctypes.data_as(C.POINTER(C.c_uint)), P(dx),
                          C.c_float(p), C.c_uint(0), C.c_uint(0))
        lib.emu_ffn_set_colpart(None)
        t = _bf16_val(_bf16_bits(dz2)).astype(np.float64) @ _bf16_val(W2b).astype(np.float64)
        want1 = np.where(got > 0, t / (1.0 - p), 0.0)
        got1 = _bf16_val(dz1).astype(np.float64)
        assert np.array_equal(got1 == 0, (got <= 0) | (want1 == 0))
        np.testing.assert_allclose(got1, want1, rtol=2.0 ** -7, atol=2e-5)
        np.testing.assert_allclose(dx, got1 @ _bf16_val(W1b).astype(np.float64), rtol=1e-5, atol=2e-5)
        assert np.isfinite(colpart).all()
        for w in range((M + 31) // 32):                          # every wave's partial = the sum of ITS rows of the stored gradient
            np.testing.assert_allclose(colpart[w], got1[32 * w:32 * w + 32].sum(0), rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(colpart.sum(0), got1.sum(0), rtol=1e-5, atol=1e-4)


def test_bf16_attention_kernels_two_blocks_per_wave():
    """csrc/attention16_kernels.hip (round 5): the reduced-precision mode's fused attention with q|k|v in memory as bf16 -- a wave owns two
    blocks of 32 rows, the walked tiles live in LDS as bf16 in the matrix instruction's operand layout, base-2 softmax.  Emulated against
    (a) float64 attention on the bf16-rounded q|k|v with the kernels' own dropout mask (at_keep_bit) -- forward output, log-sum-exp and
    all three gradients within the bf16 operand class -- and (b) round 4's kernels for the same mode (attn_*<1, DROP, 1>): the same
    mathematics and the same mask, so they agree to accumulation order.  S = 300 (two workgroups of 256 rows, ragged last tile, a wave
    whose second block is idle, a wave that is idle altogether), key bias with +1 entries and -inf keys, a fully masked sequence, and the
    live-query form (Sq < S)."""
    lib = emu.lib()
    lib.emu_attn_keep.restype = C.c_int
    rng = np.random.default_rng(9)
    U16 = lambda a: a.ctypes.data_as(C.POINTER(C.c_ushort))
    scale = 1.0 / np.sqrt(32.0)
    for (n_seq, S, Sq, H, p, seed) in ((2, 300, 300, 1, 0.0, 0), (1, 70, 70, 2, 0.1, 4242), (1, 300, 21, 1, 0.1, 77)):
        d = H * 32
        qkv32 = (rng.normal(size=(n_seq, S, 3 * d)) * 0.7).astype(np.float32)
        qkvb = _bf16_bits(qkv32)
        qkv = _bf16_val(qkvb).astype(np.float64)
        kb = np.zeros((n_seq, S), np.float32)
        kb[0, 5::7] = 1.0
        kb[-1, S - 20:] = -np.inf
        dout = rng.normal(size=(n_seq, Sq, d)).astype(np.float32)
        res = {}
        for which in (0, 1):
            out = np.full((n_seq, Sq, d), np.nan, np.float32)
            lse = np.full((n_seq * H, Sq), np.nan, np.float32)
            lib.emu_attention16(which, n_seq, S, Sq, H, d, C.c_float(scale), U16(qkvb), P(kb), P(out), P(lse), None, None, None, C.c_float(p), C.c_uint(seed))
            # rows that do not attend get dQ = 0: from the launcher's memset for round 4's kernels, from the dK / dV kernel itself in round 5's
            dqkv = np.zeros((n_seq, S, 3 * d), np.uint16) if which == 0 else np.full((n_seq, S, 3 * d), 0x7fc0, np.uint16)
            dsum = np.full((n_seq * H, Sq), np.nan, np.float32)
            lib.emu_attention16(which, n_seq, S, Sq, H, d, C.c_float(scale), U16(qkvb), P(kb), P(out), P(lse), P(dout), U16(dqkv), P(dsum), C.c_float(p), C.c_uint(seed))
            res[which] = (out, lse, _bf16_val(dqkv), dsum)
        # float64 reference with the kernels' mask
        ref_out = np.zeros((n_seq, Sq, d)); ref_lse = np.zeros((n_seq * H, Sq)); ref_dqkv = np.zeros((n_seq, S, 3 * d))
        for b in range(n_seq):
            for h in range(H):
                q, k, v = (qkv[b, :, i * d + h * 32:i * d + (h + 1) * 32] for i in range(3))
                s = q[:Sq] @ k.T * scale + kb[b].astype(np.float64)
                mx = s.max(1, keepdims=True)
                e = np.exp(s - mx)
                l = e.sum(1, keepdims=True)
                P_ = e / l
                keep = np.ones((Sq, S))
                if p > 0:
                    keep = np.array([[lib.emu_attn_keep(C.c_uint(seed), b * H + h, qi, ki, C.c_float(p)) for ki in range(S)] for qi in range(Sq)], np.float64) / (1 - ATTN_P8(p))      # (the mask realises p in 1/256ths: at_drop_thr8)
                Pd = P_ * keep
                ref_out[b, :, h * 32:(h + 1) * 32] = Pd @ v
                ref_lse[b * H + h] = (mx + np.log(l))[:, 0]
                do = dout[b, :, h * 32:(h + 1) * 32].astype(np.float64)
                dPd = do @ v.T
                dP = dPd * keep
                dS = P_ * (dP - (dP * P_).sum(1, keepdims=True))
                ref_dqkv[b, :Sq, h * 32:(h + 1) * 32] = dS @ k * scale
                ref_dqkv[b, :, d + h * 32:d + (h + 1) * 32] = dS.T @ q[:Sq] * scale
                ref_dqkv[b, :, 2 * d + h * 32:2 * d + (h + 1) * 32] = Pd.T @ do
        new, old = res[1], res[0]
        assert np.isfinite(new[0]).all() and np.isfinite(new[2]).all()
        for got, want, what, rel in ((new[0], ref_out, "out", 2e-2), (new[1], ref_lse, "lse", 2e-3), (new[2], ref_dqkv, "dqkv", 3e-2)):
            err = np.abs(got - want).max()
            assert err <= rel * np.abs(want).max() + 1e-6, (S, Sq, p, what, err, np.abs(want).max())
        # round 4's kernels of the same mode: same operands, same mask -- accumulation order (and the base of the exponential) apart
        # (a probability on a bf16 rounding boundary may fall either side: 2^-9 of one term of a row's sum)
        for a_, b_, what, tol in ((new[0], old[0], "out", 5e-3), (new[1], old[1], "lse", 2e-5), (new[3], old[3], "dsum", 5e-3), (new[2], old[2], "dqkv", 1.2e-2)):
            err = np.abs(a_ - b_).max()
            assert err <= tol * np.abs(b_).max() + 1e-7, (S, Sq, p, what, "vs round 4", err, np.abs(b_).max())
        assert np.all(new[2][-1, S - 20:, d:] == 0)                  # masked keys receive no gradient
    # a fully masked sequence: zeros, not NaN ("safe softmax"), and the sentinel log-sum-exp
    S, H, d = 40, 1, 32
    qkvb = _bf16_bits(rng.normal(size=(1, S, 3 * d)).astype(np.float32))
    kb2 = np.full((1, S), -np.inf, np.float32)
    out2 = np.ones((1, S, d), np.float32); lse2 = np.zeros((H, S), np.float32)
    lib.emu_attention16(1, 1, S, S, H, d, C.c_float(scale), U16(qkvb), P(kb2), P(out2), P(lse2), None, None, None, C.c_float(0.0), C.c_uint(0))
    assert np.all(out2 == 0) and np.all(lse2 > 1e38)


def test_split_mode_attention_on_piece_plane_tile_images():
    """csrc/attention16_kernels.hip with NP = 3 (round 5): the DEFAULT precision class of the fused attention -- fp32 q|k|v, every tile
    product rebuilt from bf16 pieces (six matrix instructions per 16 reduction entries) -- on the tile images of the bf16 kernels, one
    plane per piece: the walked tile's pieces are cut once, by the staging thread, not by every consuming wave.  Against float64 attention
    with the kernels' own dropout mask: fp32 class (1e-5 of the tensor scale; the bf16 kernels hold 2e-2), forward, log-sum-exp and all
    three gradients; and against round 4's split-mode kernels (attn_*<2, DROP, 0>): same pieces, same products, same mask -- summation
    order and the base of the exponential apart.  Ragged S, +1 and -inf key bias, dropout, the live-query form.  Round 6: the DEFAULT
    backward runs on two pieces per operand (three products): same forward bytes, gradients within 2^-16 of the tensor scale."""
    lib = emu.lib()
    lib.emu_attn_keep.restype = C.c_int
    rng = np.random.default_rng(10)
    scale = 1.0 / np.sqrt(32.0)
    VP = lambda a: a.ctypes.data_as(C.c_void_p)
    for (n_seq, S, Sq, H, p, seed) in ((2, 150, 150, 1, 0.0, 0), (1, 70, 70, 2, 0.1, 4242), (1, 150, 21, 1, 0.1, 77)):
        d = H * 32
        qkv = (rng.normal(size=(n_seq, S, 3 * d)) * 0.7).astype(np.float32)
        kb = np.zeros((n_seq, S), np.float32)
        kb[0, 5::7] = 1.0
        kb[-1, S - 20:] = -np.inf
        dout = rng.normal(size=(n_seq, Sq, d)).astype(np.float32)
        res = {}
        for which in (2, 3, 5):
            out = np.full((n_seq, Sq, d), np.nan, np.float32)
            lse = np.full((n_seq * H, Sq), np.nan, np.float32)
            lib.emu_attention16(which, n_seq, S, Sq, H, d, C.c_float(scale), VP(qkv), P(kb), P(out), P(lse), None, None, None, C.c_float(p), C.c_uint(seed))
            dqkv = np.zeros_like(qkv) if which == 2 else np.full_like(qkv, np.nan)
            dsum = np.full((n_seq * H, Sq), np.nan, np.float32)
            lib.emu_attention16(which, n_seq, S, Sq, H, d, C.c_float(scale), VP(qkv), P(kb), P(out), P(lse), P(dout), VP(dqkv), P(dsum), C.c_float(p), C.c_uint(seed))
            res[which] = (out, lse, dqkv, dsum)
        q64 = qkv.astype(np.float64)
        ref_out = np.zeros((n_seq, Sq, d)); ref_lse = np.zeros((n_seq * H, Sq)); ref_dqkv = np.zeros((n_seq, S, 3 * d))
        for b in range(n_seq):
            for h in range(H):
                q, k, v = (q64[b, :, i * d + h * 32:i * d + (h + 1) * 32] for i in range(3))
                s_ = q[:Sq] @ k.T * scale + kb[b].astype(np.float64)
                mx = s_.max(1, keepdims=True)
                e = np.exp(s_ - mx)
                l = e.sum(1, keepdims=True)
                P_ = e / l
                keep = np.ones((Sq, S))
                if p > 0:
                    keep = np.array([[lib.emu_attn_keep(C.c_uint(seed), b * H + h, qi, ki, C.c_float(p)) for ki in range(S)] for qi in range(Sq)], np.float64) / (1 - ATTN_P8(p))      # (the mask realises p in 1/256ths: at_drop_thr8)
                Pd = P_ * keep
                ref_out[b, :, h * 32:(h + 1) * 32] = Pd @ v
                ref_lse[b * H + h] = (mx + np.log(l))[:, 0]
                do = dout[b, :, h * 32:(h + 1) * 32].astype(np.float64)
                dP = (do @ v.T) * keep
                dS = P_ * (dP - (dP * P_).sum(1, keepdims=True))
                ref_dqkv[b, :Sq, h * 32:(h + 1) * 32] = dS @ k * scale
                ref_dqkv[b, :, d + h * 32:d + (h + 1) * 32] = dS.T @ q[:Sq] * scale
                ref_dqkv[b, :, 2 * d + h * 32:2 * d + (h + 1) * 32] = Pd.T @ do
        new, old, dflt = res[5], res[2], res[3]
        # the default launch (round 6): the same forward, the backward on TWO pieces per operand -- gradients to 2^-16 of the tensor scale
        assert np.array_equal(dflt[0], new[0]) and np.array_equal(dflt[1], new[1]) and np.array_equal(dflt[3], new[3])
        err2 = np.abs(dflt[2] - ref_dqkv).max()
        assert np.isfinite(dflt[2]).all() and err2 <= 1.6e-5 * np.abs(ref_dqkv).max() + 1e-7, (S, Sq, p, "two-piece backward", err2, np.abs(ref_dqkv).max())
        assert np.all(dflt[2][-1, S - 20:, d:] == 0)
        for got, want, what in ((new[0], ref_out, "out"), (new[1], ref_lse, "lse"), (new[2], ref_dqkv, "dqkv")):
            err = np.abs(got - want).max()
            assert np.isfinite(got).all() and err <= 1e-5 * np.abs(want).max() + 1e-7, (S, Sq, p, what, err, np.abs(want).max())
        for a_, b_, what in ((new[0], old[0], "out"), (new[1], old[1], "lse"), (new[3], old[3], "dsum"), (new[2], old[2], "dqkv")):
            err = np.abs(a_ - b_).max()
            assert err <= 1e-5 * np.abs(b_).max() + 1e-7, (S, Sq, p, what, "vs round 4", err)
        assert np.all(new[2][-1, S - 20:, d:] == 0)


def test_ppo_loss_heads_follow_the_torch_expressions():
    """csrc/ppo_kernels.hip (round 5): the PPO + AMP learner's loss heads -- actor (neglogp, clipped surrogate, entropy, bound loss, clip
    fraction, policy KL), critic (clipped value loss), discriminator (two binary cross entropies, accuracies) -- emulated against the
    torch expressions of learning/amp_agent.py (the restatement of amp_continuous.py:335-425 / rl_games 1.1.4), values AND gradients
    (torch autograd on the CPU): ratios inside and outside the clip range, both signs of the advantage, actions beyond the soft
    bound, value moves inside and outside the value clip."""
    import math
    import torch
    from emloco_amd.learning.amp_agent import neglogp, policy_kl
    lib = emu.lib()
    rng = np.random.default_rng(12)
    B, A, e = 203, 69, 0.2
    f = lambda *s: rng.normal(size=s).astype(np.float32)
    mu = (f(B, A) * 0.8).astype(np.float32)
    logstd = (f(B, A) * 0.05 - 1.0).astype(np.float32)
    sig = np.exp(logstd)
    actions = (mu + sig * f(B, A)).astype(np.float32)
    adv = f(B)
    old_mu, old_sigma = (mu + 0.05 * f(B, A)).astype(np.float32), (sig * np.exp(0.05 * f(B, A))).astype(np.float32)
    tmu, tls = torch.tensor(mu, requires_grad=True), torch.tensor(logstd, requires_grad=True)
    tsig = torch.exp(tls)
    nlp = neglogp(torch.tensor(actions), tmu, tsig, tls)
    old_nlp = (nlp.detach().numpy() + 0.25 * f(B)).astype(np.float32)                 # ratios on both sides of the clip range
    ratio = torch.exp(torch.tensor(old_nlp) - nlp)
    tadv = torch.tensor(adv)
    a_loss = torch.max(-tadv * ratio, -tadv * torch.clamp(ratio, 1.0 - e, 1.0 + e)).mean()
    ent = (0.5 + 0.5 * math.log(2 * math.pi) + tls).sum(-1).mean()
    bnd = (torch.clamp_max(tmu + 1.0, 0.0) ** 2 + torch.clamp_min(tmu - 1.0, 0.0) ** 2).sum(-1).mean()
    clipf = (torch.abs(ratio - 1.0) > e).float().mean()
    kl = policy_kl(tmu.detach(), tsig.detach(), torch.tensor(old_mu), torch.tensor(old_sigma))
    g3 = np.array([1.0, -0.003, 10.0], np.float32)
    (g3[0] * a_loss + g3[1] * ent + g3[2] * bnd).backward()
    rows, out5 = np.zeros((B, 5), np.float32), np.zeros(5, np.float32)
    dmu, dls = np.zeros((B, A), np.float32), np.zeros((B, A), np.float32)
    lib.emu_ppo_actor_head(B, A, P(mu), P(logstd), P(actions), P(old_nlp), P(adv), P(old_mu), P(old_sigma), C.c_float(e), P(rows), P(out5),
                           P(g3), P(dmu), P(dls))
    want = np.array([t.item() for t in (a_loss.detach(), ent.detach(), bnd.detach(), clipf, kl)])
    assert 0.2 < want[3] < 0.8 and want[2] > 0                                        # both clip states and the bound are exercised
    np.testing.assert_allclose(out5, want, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dmu, tmu.grad.numpy(), rtol=2e-4, atol=2e-7)
    np.testing.assert_allclose(dls, tls.grad.numpy(), rtol=2e-4, atol=2e-7)

    v, v_old, ret = f(B), f(B), f(B)
    v = (v_old + 0.3 * f(B)).astype(np.float32)
    for clip_value in (1, 0):
        tv = torch.tensor(v, requires_grad=True)
        tvo, tr = torch.tensor(v_old), torch.tensor(ret)
        if clip_value:
            cl = torch.max((tv - tr) ** 2, (tvo + (tv - tvo).clamp(-e, e) - tr) ** 2).mean()
        else:
            cl = ((tr - tv) ** 2).mean()
        (0.5 * cl).backward()
        rows1, out1, dv = np.zeros(B, np.float32), np.zeros(1, np.float32), np.zeros(B, np.float32)
        lib.emu_ppo_critic_head(B, P(v), P(v_old), P(ret), C.c_float(e), clip_value, P(rows1), P(out1), P(np.array([0.5], np.float32)), P(dv))
        np.testing.assert_allclose(out1[0], cl.item(), rtol=2e-6)
        np.testing.assert_allclose(dv, tv.grad.numpy(), rtol=2e-5, atol=1e-8)

    na, nd = 301, 150
    la, ld = (2.0 * f(na)).astype(np.float32), (2.0 * f(nd)).astype(np.float32)
    ta, td = torch.tensor(la, requires_grad=True), torch.tensor(ld, requires_grad=True)
    bce = torch.nn.BCEWithLogitsLoss()
    la_loss, ld_loss = bce(ta, torch.zeros_like(ta)), bce(td, torch.ones_like(td))
    (0.5 * la_loss + 0.25 * ld_loss).backward()
    rows2, out4 = np.zeros((na + nd, 2), np.float32), np.zeros(4, np.float32)
    da, dd = np.zeros(na, np.float32), np.zeros(nd, np.float32)
    lib.emu_ppo_disc_head(na, nd, P(la), P(ld), P(rows2), P(out4), P(np.array([0.5, 0.25], np.float32)), P(da), P(dd))
    np.testing.assert_allclose(out4, [la_loss.item(), (ta < 0).float().mean().item(), ld_loss.item(), (td > 0).float().mean().item()], rtol=3e-6)
    np.testing.assert_allclose(da, ta.grad.numpy(), rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(dd, td.grad.numpy(), rtol=2e-5, atol=1e-9)
