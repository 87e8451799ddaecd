"""A17 on the CPU: the LocoVal rollout bookkeeping (amp_continuous_value.py:63-145, common_agent.py:154-155,205-209) against
known answers, and its multi-rank path with rank-dependent episode ends (2 gloo ranks)."""
import os
import socket

import numpy as np
import pytest
import torch

GAMMA, STEP_TO_PRED = 0.99, 144


class ScriptedTask:
    def __init__(self, E, device="cpu"):
        self.device, self.num_envs, self.num_actions, self.step_to_pred = device, E, 69, STEP_TO_PRED
        self.obs_buf = torch.zeros(E, 8)
        self.inverted = torch.zeros(E, dtype=torch.bool)
        self.reset_buf = torch.ones(E, dtype=torch.long)


class ScriptedEnv:
    """VecEnv stand-in driven by tables: rewards[t, e], amp[t, e], done[t, e] (global env ids `ids` pick the columns), the
    LocoVal inputs of an episode are a function of (env, episode number) so that a wrong pairing of target and input shows."""

    def __init__(self, rewards, amp, done, inverted_by_episode, ids):
        self.ids = np.asarray(ids)
        E = len(ids)
        self.task = ScriptedTask(E)
        self.rewards, self.amp, self.done = rewards[:, ids], amp[:, ids], done[:, ids]
        self.inv = inverted_by_episode
        self.t = 0
        self.episode = np.zeros(E, np.int64) - 1
        self.resets = []

    def _begin_episode(self, env_ids):
        for e in env_ids.tolist():
            self.episode[e] += 1
            self.task.inverted[e] = bool(self.inv[self.ids[e], self.episode[e] % self.inv.shape[1]])
        self.task.reset_buf[env_ids] = 0
        self.resets.append((self.t, sorted(env_ids.tolist())))

    def reset(self, env_ids):
        self._begin_episode(env_ids)
        return self.task.obs_buf

    def reset_done(self):
        self._begin_episode(self.task.reset_buf.nonzero().flatten())

    def _feat(self, k):
        g = torch.Generator().manual_seed(0)
        base = torch.randn(256, 16, k, generator=g)
        return base[torch.from_numpy(self.ids % 256), torch.from_numpy(self.episode % 16)]

    def get_init_pose(self):
        return self._feat(72).view(-1, 24, 3)

    def get_waypoint_traj(self):
        f = self._feat(45).view(-1, 15, 3).clone()
        f[:, 1, 0] = f[:, 1, 0].abs() + 0.1
        return f

    def get_init_vel(self):
        return self._feat(2)

    def step(self, actions):
        r = torch.from_numpy(self.rewards[self.t]).float()
        d = torch.from_numpy(self.done[self.t]).long()
        info = {"amp_obs": torch.from_numpy(self.amp[self.t]).float()}
        self.task.reset_buf[:] = d
        self.t += 1
        return self.task.obs_buf, r, d, info


def script(E, T, seed=0):
    rng = np.random.default_rng(seed)
    rewards = rng.uniform(-0.2, 1.0, size=(T, E)).astype(np.float32)
    amp = rng.uniform(0.0, 2.0, size=(T, E)).astype(np.float32)
    done = np.zeros((T, E), np.int64)
    age = np.zeros(E, np.int64)
    plan = rng.integers(1, 168, size=(E, 8))
    plan[0, :] = 168                                  # env 0 always times out (> step_to_pred: emitted at step 144)
    plan[1, :] = 144                                  # env 1 ends exactly at the cut-off (done_early AND length == 144)
    plan[2, :] = [3, 145, 1, 143, 168, 7, 7, 7]       # env 2: very short, one past the cut-off, single-step episodes
    ep = np.zeros(E, np.int64)
    for t in range(T):
        age += 1
        for e in range(E):
            if age[e] >= plan[e, ep[e] % 8]:
                done[t, e] = 1
                age[e] = 0
                ep[e] += 1
    inverted = rng.random((E, 16)) < 0.4
    return rewards, amp, done, inverted


def expected_events(rewards, amp, done, inverted, pen=0.3):
    """Independent restatement: per env, walk the episodes; an episode's return sum_t gamma^t (r_t * (-pen if inverted) + amp_t)
    over its first min(length, 144) steps is emitted at the step it ends (length <= 144) or at its 144th step."""
    T, E = rewards.shape
    out = np.zeros((T, E), np.float64)
    for e in range(E):
        t0, ep = 0, 0
        while t0 < T:
            inv = inverted[e, ep % inverted.shape[1]]
            G, length = 0.0, 0
            t = t0
            while t < T:
                r = float(rewards[t, e]) * (-pen if inv else 1.0)
                if length < STEP_TO_PRED:
                    G += (GAMMA ** length) * (r + float(amp[t, e]))
                length += 1
                if done[t, e] and length <= STEP_TO_PRED:
                    out[t, e] = G
                elif length == STEP_TO_PRED and not done[t, e]:
                    out[t, e] = G
                t += 1
                if done[t - 1, e]:
                    break
            t0, ep = t, ep + 1
    return out


def test_return_accumulator_known_answer():
    """3 scripted envs (time-outs, exact cut-off, 1-step episodes) + 5 random ones over 400 steps: every emitted value, at
    the step and env where it is emitted, equals the closed form; nothing else is emitted."""
    from oracle.locoval_returns import ReturnAccumulator
    E, T = 8, 400
    rewards, amp, done, inverted = script(E, T)
    exp = expected_events(rewards, amp, done, inverted)
    acc = ReturnAccumulator(E, STEP_TO_PRED, GAMMA, "cpu")
    ep = np.zeros(E, np.int64)
    got = np.zeros((T, E))
    for t in range(T):
        inv = torch.from_numpy(inverted[np.arange(E), ep % 16])
        r = torch.from_numpy(rewards[t])
        r = torch.where(inv, r * (-0.3), r)
        got[t] = acc.update(r, torch.from_numpy(amp[t]), torch.from_numpy(done[t])).numpy()
        ep += done[t]
    assert (exp != 0).sum() > 30
    np.testing.assert_array_equal(got != 0, exp != 0)
    np.testing.assert_allclose(got, exp, rtol=2e-5, atol=1e-5)
    assert exp[143, 0] != 0 and exp[167, 0] == 0                     # env 0: emitted at its 144th step, not at the time-out
    assert exp[143, 1] != 0                                          # env 1: done exactly at 144 -> once


def _a17_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "locoval_returns.npz"))


def test_return_accumulator_matches_the_reference_play_steps():
    """The torch restatement (oracle/locoval_returns.py) against the fixture produced by the reference's own
    AMPValueAgent.play_steps (tests/golden/gen_golden_a17.py): which rows enter the fit at which step, their targets
    (G + 10) / 110, and the running state (discounted sum, gamma^t, episode length, undiscounted return) after every step."""
    from oracle.locoval_returns import ReturnAccumulator
    g = _a17_golden()
    T, E = g["rewards"].shape
    acc = ReturnAccumulator(E, int(g["step_to_pred"]), float(g["gamma"]), "cpu")
    pen, lo, hi = float(g["penalty"]), float(g["min_cum"]), float(g["max_cum"])
    for t in range(T):
        r = torch.from_numpy(g["rewards"][t])
        r = torch.where(torch.from_numpy(g["inverted"][t]), r * (-pen), r)
        em = acc.update(r, torch.from_numpy(g["amp"][t]), torch.from_numpy(g["dones"][t]))
        valid = (em != 0).numpy()
        np.testing.assert_array_equal(valid, g["valid"][t], err_msg=f"step {t}")
        np.testing.assert_array_equal(((em - lo) / (hi - lo)).numpy()[valid], g["target"][t][valid], err_msg=f"step {t}")
        np.testing.assert_array_equal(acc.current_combined_rewards.numpy(), g["cur_combined"][t], err_msg=f"step {t}")
        np.testing.assert_array_equal(acc.discount_coefs.numpy(), g["discount"][t], err_msg=f"step {t}")
        np.testing.assert_array_equal(acc.current_lengths.numpy(), g["lengths"][t], err_msg=f"step {t}")
        np.testing.assert_array_equal(acc.current_rewards.numpy(), g["cur_rewards"][t], err_msg=f"step {t}")
    assert g["valid"].sum() >= 30 and g["valid"][143].any() and not g["valid"][167, 1]


@pytest.mark.parametrize("staged", [False, True])
def test_fused_returns_kernel_matches_the_reference_play_steps(staged):
    """locoval_returns_kernel (the product's bookkeeping kernel, run through the CPU emulator) against the same fixture: the
    emitted targets / weights and the four state arrays, element for element over the 400 scripted steps.  `staged`: the two-phase
    form a rollout with an AMP discriminator uses -- the step is staged WITHOUT the style reward (which exists three GEMMs later,
    while the resets that follow overwrite what the task knows), state untouched, and locoval_returns_finish_kernel completes it
    with the reward: the reference's bytes again."""
    import ctypes as C
    import emu
    from emloco_amd.predictor.ops import LocoValStep
    g = _a17_golden()
    T, E = g["rewards"].shape
    f = lambda *s: np.zeros(s, np.float32)
    st = dict(cr=f(E), cl=f(E), cc=f(E), dc=np.ones(E, np.float32), traj13=f(E, 13, 3), pose=f(E, 24, 3), vel=f(E, 2), target=f(E), weight=f(E))
    wp, ip, iv = f(E, 15, 3), f(E, 24, 3), f(E, 2)
    p = lambda a: a.ctypes.data
    s = LocoValStep(E, int(g["step_to_pred"]), float(g["gamma"]), float(g["penalty"]), float(g["min_cum"]), float(g["max_cum"]), p(st["cr"]),
                    p(st["cl"]), p(st["cc"]), p(st["dc"]), p(wp), p(ip), p(iv), p(st["traj13"]), p(st["pose"]), p(st["vel"]), p(st["target"]), p(st["weight"]))
    fn = emu.lib().emu_locoval_returns
    fn.argtypes = [C.c_void_p] * 5
    fin = emu.lib().emu_locoval_returns_finish
    fin.argtypes = [C.c_void_p] * 2
    sr, sd = f(E), np.zeros(E, np.uint8)
    if staged:
        s.staged_reward, s.staged_done = p(sr), p(sd)
    for t in range(T):
        keep = (np.ascontiguousarray(g["rewards"][t]), np.ascontiguousarray(g["amp"][t]), np.ascontiguousarray(g["dones"][t]),
                g["inverted"][t].astype(np.uint8))
        if staged:
            before = {k: st[k].copy() for k in ("cr", "cl", "cc", "dc")}
            fn(C.addressof(s), p(keep[0]), None, p(keep[2]), p(keep[3]))
            assert all(np.array_equal(before[k], st[k]) for k in before)          # staging leaves the state alone
            np.testing.assert_array_equal(sd != 0, keep[2] != 0)
            keep[0][:] = np.nan; keep[2][:] = 0; keep[3][:] = 0                    # the resets that follow may overwrite the task's buffers
            fin(C.addressof(s), p(keep[1]))
        else:
            fn(C.addressof(s), *[p(k) for k in keep])
        valid = g["valid"][t]
        np.testing.assert_array_equal(st["weight"] != 0, valid, err_msg=f"step {t}")
        np.testing.assert_array_equal(st["target"][valid], g["target"][t][valid], err_msg=f"step {t}")
        np.testing.assert_array_equal(st["cc"], g["cur_combined"][t], err_msg=f"step {t}")
        np.testing.assert_array_equal(st["dc"], g["discount"][t], err_msg=f"step {t}")
        np.testing.assert_array_equal(st["cl"], g["lengths"][t], err_msg=f"step {t}")
        np.testing.assert_array_equal(st["cr"], g["cur_rewards"][t], err_msg=f"step {t}")


def test_fused_returns_kernel_equals_the_accumulator():
    """locoval_returns_kernel (the fused step's bookkeeping, run through the CPU emulator) against ReturnAccumulator and the
    closed form: emitted targets / weights element for element over 400 scripted steps, LocoVal inputs origin-relative."""
    import ctypes as C
    import emu
    from oracle.locoval_returns import ReturnAccumulator
    from emloco_amd.predictor.ops import LocoValStep
    E, T = 8, 400
    rewards, amp, done, inverted = script(E, T)
    exp = expected_events(rewards, amp, done, inverted)
    acc = ReturnAccumulator(E, STEP_TO_PRED, GAMMA, "cpu")
    f = lambda *s: np.zeros(s, np.float32)
    st = dict(cr=f(E), cl=f(E), cc=f(E), dc=np.ones(E, np.float32), traj13=f(E, 13, 3), pose=f(E, 24, 3), vel=f(E, 2), target=f(E), weight=f(E))
    rng = np.random.default_rng(1)
    wp, ip, iv = rng.normal(size=(E, 15, 3)).astype(np.float32), rng.normal(size=(E, 24, 3)).astype(np.float32), rng.normal(size=(E, 2)).astype(np.float32)
    p = lambda a: a.ctypes.data
    s = LocoValStep(E, STEP_TO_PRED, GAMMA, 0.3, -10.0, 100.0, p(st["cr"]), p(st["cl"]), p(st["cc"]), p(st["dc"]), p(wp), p(ip), p(iv),
                    p(st["traj13"]), p(st["pose"]), p(st["vel"]), p(st["target"]), p(st["weight"]))
    fn = emu.lib().emu_locoval_returns
    fn.argtypes = [C.c_void_p] * 5
    ep = np.zeros(E, np.int64)
    for t in range(T):
        inv = inverted[np.arange(E), ep % 16]
        keep = (np.ascontiguousarray(rewards[t]), np.ascontiguousarray(amp[t]), np.ascontiguousarray(done[t]), inv.astype(np.uint8))
        fn(C.addressof(s), *[p(k) for k in keep])
        r = torch.from_numpy(rewards[t])
        r = torch.where(torch.from_numpy(inv), r * (-0.3), r)
        em = acc.update(r, torch.from_numpy(amp[t]), torch.from_numpy(done[t])).numpy()
        np.testing.assert_array_equal(st["weight"], (em != 0).astype(np.float32))
        np.testing.assert_array_equal(st["target"], ((em - (-10.0)) / np.float32(110.0)).astype(np.float32))
        np.testing.assert_array_equal(st["cc"], acc.current_combined_rewards.numpy())
        np.testing.assert_array_equal(st["dc"], acc.discount_coefs.numpy())
        np.testing.assert_array_equal(st["weight"] != 0, exp[t] != 0)
        ep += done[t]
    np.testing.assert_array_equal(st["traj13"], wp[:, :13] - wp[:, :1])
    np.testing.assert_array_equal(st["pose"], ip - ip[:, :1])
    np.testing.assert_array_equal(st["vel"], iv)


def test_fused_fit_grad_and_gated_adamw_kernels():
    """locoval_fit_grad_kernel (sum-MSE gradient + [loss, count]) and adamw_gated_kernel against torch: 40 steps with the gate
    closed on some of them equal torch.optim.AdamW stepped only on the open ones; a closed gate leaves every buffer untouched."""
    import ctypes as C
    import emu
    rng = np.random.default_rng(0)
    n = 4096
    v, tg = rng.random(n).astype(np.float32), rng.random(n).astype(np.float32)
    w = (rng.random(n) < 0.1).astype(np.float32)
    dv, tail = np.zeros(n, np.float32), np.zeros(2, np.float32)
    fg = emu.lib().emu_locoval_fit_grad
    fg.argtypes = [C.c_int] + [C.c_void_p] * 6
    slot = np.zeros(n, np.int32)
    fg(n, v.ctypes.data, tg.ctypes.data, w.ctypes.data, dv.ctypes.data, tail.ctypes.data, slot.ctypes.data)
    np.testing.assert_array_equal(slot, np.where(w != 0, np.cumsum(w != 0) - 1, -1))        # rank among the valid rows, in row order
    np.testing.assert_allclose(dv, 2 * w * (v - tg), rtol=0, atol=0)
    assert tail[1] == w.sum() and abs(tail[0] - float((w * (v - tg) ** 2).sum())) < 1e-3
    N = 6174
    fa = emu.lib().emu_adamw_gated
    fa.argtypes = [C.c_int] + [C.c_void_p] * 7 + [C.c_float] * 5 + [C.c_void_p]
    p0 = rng.normal(size=N).astype(np.float32)
    p, m, vv = p0.copy(), np.zeros(N, np.float32), np.zeros(N, np.float32)
    steps = [np.zeros(1, np.float32), np.zeros(1, np.float32)]
    stats = np.zeros(5, np.float64)
    ref = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=1e-4)
    flip, n_open = 0, 0
    for it in range(40):
        g = rng.normal(size=N).astype(np.float32)
        gate = it % 4 != 2
        tl = np.array([0.5 * it, 3.0 if gate else 0.0], np.float32)
        before = (p.copy(), m.copy(), vv.copy())
        fa(N, p.ctypes.data, g.ctypes.data, m.ctypes.data, vv.ctypes.data, steps[flip].ctypes.data, steps[flip ^ 1].ctypes.data, tl.ctypes.data,
           1e-3, 0.9, 0.999, 1e-8, 1e-4, stats.ctypes.data)
        flip ^= 1
        if gate:
            n_open += 1
            ref.grad = torch.from_numpy(g.copy())
            opt.step()
        else:
            assert np.array_equal(p, before[0]) and np.array_equal(m, before[1]) and np.array_equal(vv, before[2])
        assert steps[flip][0] == n_open
    np.testing.assert_allclose(p, ref.detach().numpy(), rtol=2e-6, atol=2e-7)
    assert stats[4] == n_open and stats[3] == 3.0 * n_open and stats[1] == 3.0


def _stand_in_valuenet(seed):
    from oracle.predictor_torch import LocoValOracle
    torch.manual_seed(seed)
    return LocoValOracle()


def _run_rollout(env, horizon, epochs, seed=5, record=None):
    from oracle.locoval_returns import TorchLocoValRollout as LocoValRollout
    agent = LocoValRollout(env, horizon_length=horizon, valuenet=_stand_in_valuenet(seed), disc_reward=lambda a: a,
                           policy=lambda obs: torch.zeros(env.task.num_envs, 69))
    if record is not None:
        orig = agent._fit

        def fit():
            record.append((agent.game_combined_rewards.clone(), env.episode.copy()))
            orig()
        agent._fit = fit
    for _ in range(epochs):
        agent.play_steps()
    return agent


def test_locoval_rollout_targets_inputs_and_schedule():
    """The whole play_steps loop on a scripted env: the targets (G + 10) / 110 reach the fit paired with the inputs of the
    episode they belong to, the inversion penalty is 0.3 and the sum task + disc reward is unweighted, finished envs are
    reset at the start of the next step, the learning rate follows the reference's schedule."""
    E, H, EPOCHS = 8, 25, 12
    rewards, amp, done, inverted = script(E, H * EPOCHS)
    rec = []
    env = ScriptedEnv(rewards, amp, done, inverted, np.arange(E))
    agent = _run_rollout(env, H, EPOCHS, record=rec)
    exp = expected_events(rewards, amp, done, inverted, pen=0.3)
    got = np.stack([r[0].numpy() for r in rec])
    np.testing.assert_allclose(got, exp, rtol=2e-5, atol=1e-5)
    assert agent.fitted_episodes == int((exp != 0).sum()) and agent.frames == E * H * EPOCHS
    # resets: everything at t = 0, then exactly the envs that were done on the previous step
    assert env.resets[0] == (0, list(range(E)))
    for t, ids in env.resets[1:]:
        assert ids == np.nonzero(done[t - 1])[0].tolist()
    # one hand-checked number: env 1's first episode (144 steps, done at the cut-off)
    inv = inverted[1, 0]
    G = sum(GAMMA ** k * (float(rewards[k, 1]) * (-0.3 if inv else 1.0) + float(amp[k, 1])) for k in range(144))
    assert abs(got[143, 1] - G) < 1e-3 and abs((got[143, 1] + 10) / 110 - (G + 10) / 110) < 1e-5
    # replay the fit sequence with plain torch: same weights at the end
    net = _stand_in_valuenet(5)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-4)
    from emloco_amd.learning.scheduler import CosineAnnealingLR
    sch = CosineAnnealingLR(opt, warmup_epochs=20, max_epochs=20000)
    env2 = ScriptedEnv(rewards, amp, done, inverted, np.arange(E))
    env2.reset(torch.arange(E))
    for t in range(H * EPOCHS):
        if t > 0:
            env2.reset_done()
        env2.step(None)
        idx = torch.from_numpy(np.nonzero(exp[t])[0])
        if len(idx):
            pred = net(env2.get_waypoint_traj()[:, :13], env2.get_init_pose(), env2.get_init_vel()).squeeze(-1)
            target = (torch.from_numpy(got[t]).float()[idx] + 10.0) / 110.0
            opt.zero_grad()
            torch.nn.functional.mse_loss(pred[idx], target, reduction="sum").backward()
            opt.step()
        if (t + 1) % H == 0:
            sch.step()
    for a, b in zip(agent.valuenet.parameters(), net.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert abs(agent.vnet_optimizer.param_groups[0]["lr"] - opt.param_groups[0]["lr"]) < 1e-12


def test_cosine_schedule_matches_reference(golden):
    from emloco_amd.learning.scheduler import CosineAnnealingLR
    g = golden("locoval_lr_schedule")
    for tag in ("short", "locoval"):
        w, T, n = [int(v) for v in g["cfg_" + tag]]
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=1e-4)
        sch = CosineAnnealingLR(opt, warmup_epochs=w, max_epochs=T)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(n):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        np.testing.assert_allclose(np.array(lrs), g["lr_" + tag], rtol=1e-12, atol=0)


# ---------------------------------------------------------------------------------------------- two ranks
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_worker(rank, world, port, q, E_local, H, EPOCHS):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from emloco_amd.dist import init_from_env
    init_from_env("gloo")
    torch.set_num_threads(1)
    rewards, amp, done, inverted = script(E_local * world, H * EPOCHS)
    if rank == 1:
        done[:, E_local:] = 0
        done[7::31, E_local:] = 1                      # rank 1 finishes episodes rarely: many steps with no local episode
    ids = np.arange(rank * E_local, (rank + 1) * E_local)
    env = ScriptedEnv(rewards, amp, done, inverted, ids)
    agent = _run_rollout(env, H, EPOCHS, seed=5 + rank)      # different seeds: the broadcast must make the replicas equal
    q.put((rank, [p.detach().numpy().copy() for p in agent.valuenet.parameters()], agent.fitted_episodes, agent.vnet_fits))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_locoval_rollout_equals_single_rank_on_all_envs():
    """LocoValRollout on 2 gloo ranks whose envs finish at different steps (steps where only one rank -- or neither -- has an
    episode): no dead-lock (one collective per step on every rank), identical weights on both ranks, and the same weights as
    one rank running all the envs (sum-reduced gradients, global episode count)."""
    import torch.multiprocessing as mp
    world, E_local, H, EPOCHS = 2, 4, 20, 6
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, q, E_local, H, EPOCHS)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, n0, f0), (_, w1, n1, f1) = res
    assert n0 == n1 and f0 == f1 and n0 > 10
    for a, b in zip(w0, w1):
        assert np.array_equal(a, b)
    rewards, amp, done, inverted = script(E_local * world, H * EPOCHS)
    done[:, E_local:] = 0
    done[7::31, E_local:] = 1
    env = ScriptedEnv(rewards, amp, done, inverted, np.arange(E_local * world))
    single = _run_rollout(env, H, EPOCHS, seed=5)
    assert single.fitted_episodes == n0 and single.vnet_fits == f0
    for a, b in zip(w0, single.valuenet.parameters()):
        np.testing.assert_allclose(a, b.detach().numpy(), rtol=2e-5, atol=2e-6)
