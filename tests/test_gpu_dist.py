"""GPU: the data-parallel paths with the real trainers, two ranks sharing the one GPU of the test box (gloo; device tensors
staged through the host by emloco_amd.dist), and the NaN / zero-pose row handling of the EmLoco train step (B8)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, world, *args, timeout=600):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=timeout) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from emloco_amd.dist import init_from_env
    init_from_env("gloo")
    torch.cuda.set_device(0)


# ------------------------------------------------------------------------------------------------ predictor
def _jta_cfg(dev, multi=False):
    return {"DEVICE": dev, "MULTI_MODAL": multi, "TRAIN": {"input_track_size": 9, "output_track_size": 12, "lr": 1e-3,
                                                          "max_grad_norm": 1.0, "valuenet_weight": 1.0}}


def _jta_model(dev, seed, multi=False):
    from emloco_amd.predictor.model_jta import TransMotionJTA
    torch.manual_seed(seed)
    return TransMotionJTA(tok_dim=453, nhid=128, nhead=4, dim_feedfwd=256, nlayers_local=1, nlayers_global=1, nmode=3, output_scale=1,
                          obs_and_pred=21, num_tokens=49, device=dev, dropout=0.0, multi_modal=multi).to(dev)


def _jta_batch(B, N, seed, nan_rows=True):
    g = torch.Generator().manual_seed(seed)
    joints = torch.randn(B, N, 21, 49, 4, generator=g) * 0.3
    joints[:, :, :, 0, :2] = torch.cumsum(torch.rand(B, N, 21, 2, generator=g) * 0.4, dim=2)
    if nan_rows:
        joints[1, 0, 8, 5, 1] = float("nan")             # NaN in the primary pose -> row dropped by nan_handler
        joints[2, 0, 8, 3:27, :3] = 0.0                  # all-zero pose -> row dropped
        joints[5 % B, 0, 7, 0, 0] = float("nan")         # NaN trajectory sample at t = 7 -> NaN velocity -> row dropped
    masks = torch.ones(B, N, 21, 49)
    pad = torch.zeros(B, N, dtype=torch.bool)
    pad[0, N - 1] = True
    return joints, masks, pad


def _vnet(dev, seed=11):
    from emloco_amd.learning.value_pose_net import ValuePoseNet
    torch.manual_seed(seed)
    return ValuePoseNet(True, True).to(dev)


@pytest.mark.parametrize("multi", [False, True])
def test_emloco_loss_handles_nan_and_zero_pose_rows_like_the_reference(multi):
    """B8 (train_jta.py:143-165,288-308): rows with a NaN pose, an all-zero pose or a NaN velocity leave the EmLoco loss.  The
    masked, synchronisation-free form used by EmLocoTrainer.step equals the row-dropping restatement of the reference's
    nan_handler and the stock-torch LocoVal oracle on the surviving rows -- value and gradient; a batch with no surviving
    row adds nothing."""
    from oracle.predictor_torch import LocoValOracle
    from emloco_amd.predictor.train_jta import batch_process_coords, compute_loss, emloco_loss, emloco_loss_masked
    dev = "cuda:0"
    cfg = _jta_cfg(dev, multi)
    model = _jta_model(dev, 3, multi)
    vnet = _vnet(dev)
    for p in vnet.parameters():
        p.requires_grad_(False)
    joints, masks, pad = _jta_batch(8, 2, 0)
    in_j, in_m, out_j, out_m, pm = batch_process_coords(joints, masks, pad, cfg, training=True)
    pose = joints[:, 0, 8, 3:27, :3].clone().to(dev)
    pose[..., 2] *= -1
    vel = ((in_j[:, 8, 0, :2] - in_j[:, 7, 0, :2]) * 2.5).clone()
    assert torch.isnan(pose[1]).any() and (pose[2] == 0).all() and torch.isnan(vel[5]).any()
    model.eval()
    _, pred = compute_loss(model, cfg, in_j, out_j, in_m, out_m, pm.to(dev), mode='val')
    assert torch.isfinite(pred).all()                                   # NaN inputs were zeroed (train_jta.py:102-103)
    pred = pred.detach().requires_grad_(True)
    vsum, cnt = emloco_loss_masked(cfg, vnet, pred, pose.clone(), vel.clone(), 9)
    assert int(cnt.item()) == 5
    masked = vsum / cnt
    masked.backward()
    g_masked = pred.grad.clone()
    pred.grad = None
    dropped = emloco_loss(cfg, vnet, pred, pose.clone(), vel.clone(), 9)      # reference control flow: boolean-mask row drop
    dropped.backward()
    assert abs(masked.item() - dropped.item()) <= 1e-6 * max(1.0, abs(dropped.item()))
    assert torch.allclose(g_masked, pred.grad, rtol=1e-5, atol=1e-8)
    assert (g_masked[[1, 2, 5]] == 0).all() and (g_masked[[0, 3, 4, 6, 7]].abs().sum(dim=(1, 2, 3)) > 0).all()
    # stock-torch oracle on the survivors
    keep = torch.tensor([0, 3, 4, 6, 7], device=dev)
    orc = LocoValOracle().to(dev)
    orc.load_state_dict(vnet.state_dict())
    M = pred.shape[2] if multi else 1
    tot = 0.0
    po = pose[keep].clone()
    for i in range(M):
        tr = torch.cat([torch.zeros(5, 1, 2, device=dev), pred.detach()[keep, 9:, i]], dim=1)
        v = orc(tr, po, vel[keep])
        tot = tot + ((v - 1) ** 2).mean()
        if multi:     # the reference rotates the caller's pose in place, cumulatively over the modes (value_pose_net.py:97)
            a = torch.atan2(tr[:, 1, 1], torch.where(tr[:, 1, 0].abs() < 1e-10, torch.full_like(tr[:, 1, 0], 1e-10), tr[:, 1, 0]))
            R = torch.stack([torch.cos(a), -torch.sin(a), torch.sin(a), torch.cos(a)], -1).view(-1, 2, 2)
            po = torch.cat([torch.bmm(po[..., :2], R), po[..., 2:]], -1)
            po[:, [4, 8, 9, 10, 11]] = 0
    assert abs(tot.item() / M - masked.item()) <= 2e-5 * max(1.0, abs(masked.item()))
    # nothing survives -> contributes exactly 0, the step stays finite
    zsum, zcnt = emloco_loss_masked(cfg, vnet, pred.detach(), torch.zeros_like(pose), vel.clone(), 9)
    assert zcnt.item() == 0 and (zsum / zcnt.clamp(min=1)).item() == 0.0


def test_train_step_with_nan_rows_stays_finite_and_moves_the_weights():
    from emloco_amd.predictor.train_jta import EmLocoTrainer
    dev = "cuda:0"
    model = _jta_model(dev, 3)
    tr = EmLocoTrainer(model, _vnet(dev), _jta_cfg(dev))
    w0 = model.fc_out_traj.weight.detach().clone()
    joints, masks, pad = _jta_batch(8, 2, 0)
    for _ in range(3):
        loss, mse = tr.step(joints, masks, pad)
    assert torch.isfinite(loss) and torch.isfinite(mse)
    assert all(torch.isfinite(p).all() for p in model.parameters()) and not torch.equal(w0, model.fc_out_traj.weight)


def _trainer_worker(rank, world, port, q, steps):
    _init(rank, world, port)
    import torch.distributed as dist
    from emloco_amd.predictor.train_jta import EmLocoTrainer
    dev = "cuda:0"
    model = _jta_model(dev, 100 + rank)                    # different initial weights: the constructor's broadcast equalises them
    tr = EmLocoTrainer(model, _vnet(dev, 11 + rank), _jta_cfg(dev), data_parallel=True)
    joints, masks, pad = _jta_batch(8, 2, 0)
    sl = slice(rank * 4, (rank + 1) * 4)
    for _ in range(steps):
        tr.step(joints[sl], masks[sl], pad[sl], random_masking=False)
    torch.cuda.synchronize()
    q.put((rank, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_emloco_trainer_equals_single_rank_on_the_concatenated_batch():
    """EmLocoTrainer(data_parallel=True) on two ranks (4 + 4 scenes, NaN / zero-pose rows spread unevenly: rank 0 keeps 2 rows,
    rank 1 keeps 3): identical weights on both ranks after 3 Adam steps, equal to one process stepping on the 8 scenes."""
    from emloco_amd.predictor.train_jta import EmLocoTrainer
    res = _spawn(_trainer_worker, 2, 3)
    sd0, sd1 = res[0][1], res[1][1]
    for k in sd0:
        assert np.array_equal(sd0[k], sd1[k]), k
    dev = "cuda:0"
    model = _jta_model(dev, 100)
    tr = EmLocoTrainer(model, _vnet(dev, 11), _jta_cfg(dev))
    joints, masks, pad = _jta_batch(8, 2, 0)
    for _ in range(3):
        tr.step(joints, masks, pad, random_masking=False)
    for k, v in model.state_dict().items():
        a, b = sd0[k], v.detach().cpu().numpy()
        if k.endswith("self_attn.in_proj_bias"):
            # the KEY third of the bias has no gradient at all (a constant added to every key shifts a query's scores by one number, which
            # the softmax drops): what arrives is the rounding noise of the backward products, and Adam turns noise above its eps
            # (1e-8) into steps of +-lr whatever its size -- two ranks and one process see different noise (round 6's two-piece backward:
            # ~1e-7 where three pieces left ~1e-9).  Bound: 3 steps x lr; the query and value thirds compare like every other tensor.
            d = a.shape[0] // 3
            assert np.abs(a[d:2 * d] - b[d:2 * d]).max() <= 3 * _jta_cfg(dev)["TRAIN"]["lr"] * 1.01, (k, "key third")
            a, b = np.concatenate([a[:d], a[2 * d:]]), np.concatenate([b[:d], b[2 * d:]])
        assert np.abs(a - b).max() <= 2e-5 + 2e-4 * np.abs(b).max(), (k, np.abs(a - b).max())


# ------------------------------------------------------------------------------------------------ rollout
def _rollout_worker(rank, world, port, q, E, horizon, epochs):
    _init(rank, world, port)
    import torch.distributed as dist
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    from emloco_amd.run import create_rlgpu_env, fill_flags
    from emloco_amd.utils.config import get_args, load_cfg
    args = get_args(["--num_envs", str(E), "--seed", "3", "--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel",
                     "--input_init_pose", "--input_init_vel"])
    cfg, cfg_train, _ = load_cfg(args)
    fill_flags(args)
    env = create_rlgpu_env(args, cfg, cfg_train, rank=rank)           # seed + rank: different episodes per rank
    agent = LocoValRollout(env, horizon_length=horizon)
    w_start = [p.detach().cpu().numpy().copy() for p in agent.valuenet.parameters()]
    # large exploration noise: humanoids fall at different times on the two ranks
    agent.policy = lambda obs: torch.randn(E, 69, device=obs.device) * (0.6 if rank == 0 else 0.05)
    for _ in range(epochs):
        agent.play_steps()
    torch.cuda.synchronize()
    q.put((rank, w_start, [p.detach().cpu().numpy() for p in agent.valuenet.parameters()], agent.fitted_episodes, agent.vnet_fits,
           float(agent.vnet_loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_locoval_rollout_on_the_real_env_keeps_replicas_identical():
    """LocoValRollout over the real task on two ranks (64 envs each, seed + rank, very different termination rates): the
    replicas start from rank 0's LocoVal weights, issue one collective per step each, and hold bit-identical weights after
    48 steps in which the weights moved."""
    res = _spawn(_rollout_worker, 2, 64, 16, 3)
    (_, s0, w0, n0, f0, l0), (_, s1, w1, n1, f1, l1) = res
    for a, b in zip(s0, s1):
        assert np.array_equal(a, b)                                    # broadcast at construction
    for a, b in zip(w0, w1):
        assert np.array_equal(a, b)
    assert n0 == n1 > 0 and f0 == f1 > 0 and l0 == l1 and np.isfinite(l0)
    assert any(not np.array_equal(a, b) for a, b in zip(s0, w0))


def test_fused_locoval_step_equals_the_torch_formulation():
    """The fused rollout step (emloco_locoval_returns / _fit_grad / _adamw_gated around the LocoVal kernels) against the torch
    formulation of the same loop (oracle/locoval_returns.py: ReturnAccumulator + masked fit + GatedFlatAdamW), both on a scripted env on the GPU: after 300
    steps with ~17 fits the LocoVal weights agree to 1e-5, the counters exactly."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_locoval_rollout_cpu import ScriptedEnv, script
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    from emloco_amd.learning.value_pose_net import ValuePoseNet
    dev = torch.device("cuda", 0)
    E, H, EPOCHS = 8, 25, 12

    class GpuScripted(ScriptedEnv):
        def __init__(self, *a):
            super().__init__(*a)
            t = self.task
            t.device = dev
            t.obs_buf, t.inverted, t.reset_buf = t.obs_buf.to(dev), t.inverted.to(dev), t.reset_buf.to(dev)
            t.waypoint_traj = torch.zeros(E, 15, 3, device=dev)
            t.init_pose = torch.zeros(E, 24, 3, device=dev)
            t.init_vel = torch.zeros(E, 2, device=dev)

        def _begin_episode(self, env_ids):
            env_ids = env_ids.cpu()
            inv = self.task.inverted.cpu()
            rb = self.task.reset_buf.cpu()
            for e in env_ids.tolist():
                self.episode[e] += 1
                inv[e] = bool(self.inv[self.ids[e], self.episode[e] % self.inv.shape[1]])
            rb[env_ids] = 0
            self.task.inverted, self.task.reset_buf = inv.to(dev), rb.to(dev)
            # what the task captures at reset (humanoid_pedestrain_terrain.py:511-516): raw samples / body positions / velocity
            self.task.waypoint_traj.copy_(ScriptedEnv.get_waypoint_traj(self) + 3.0)
            self.task.init_pose.copy_(ScriptedEnv.get_init_pose(self) - 1.5)
            self.task.init_vel.copy_(ScriptedEnv.get_init_vel(self))

        def get_waypoint_traj(self):
            w = self.task.waypoint_traj.clone()
            return w - w[:, :1]

        def get_init_pose(self):
            p = self.task.init_pose.clone()
            return p - p[:, :1]

        def get_init_vel(self):
            return self.task.init_vel.clone()

        def step(self, actions):
            o, r, d, info = super().step(actions)
            self.task.reset_buf = self.task.reset_buf.to(dev)
            return o, r.to(dev), d.to(dev), {"amp_obs": info["amp_obs"].to(dev)}

    rewards, amp, done, inverted = script(E, H * EPOCHS)
    agents = []
    from oracle.locoval_returns import TorchLocoValRollout
    for cls in (LocoValRollout, TorchLocoValRollout):
        torch.manual_seed(21)
        env = GpuScripted(rewards, amp, done, inverted, np.arange(E))
        ag = cls(env, horizon_length=H, valuenet=ValuePoseNet(True, True, inplace_pose=False), disc_reward=lambda a: a,
                 policy=lambda obs: torch.zeros(E, 69, device=dev))
        for _ in range(EPOCHS):
            ag.play_steps()
        agents.append(ag)
    a, b = agents
    assert a.fused and not b.fused
    assert a.fitted_episodes == b.fitted_episodes > 15 and a.vnet_fits == b.vnet_fits, (a.fitted_episodes, b.fitted_episodes, a.vnet_fits, b.vnet_fits)
    assert abs(a.vnet_loss - b.vnet_loss) <= 1e-5 * max(1.0, abs(b.vnet_loss))
    for p, q in zip(a.valuenet.parameters(), b.valuenet.parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), (p - q).abs().max()
    w0 = ValuePoseNet(True, True)
    assert any(not torch.equal(p.cpu(), q) for p, q in zip(a.valuenet.parameters(), w0.parameters()))


def test_locoval_returns_kernel_matches_the_reference_play_steps():
    """A17 through the C ABI on the device: emloco_locoval_returns over the 400 scripted steps of the fixture that the reference's
    own AMPValueAgent.play_steps produced (tests/golden/gen_golden_a17.py) -- rows that enter the fit, their targets
    (G + 10) / 110, and the running sums / discounts / lengths after every step, element for element."""
    import ctypes as C
    from emloco_amd.predictor import ops
    from emloco_amd.sim import current_stream_handle
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "locoval_returns.npz"))
    dev = torch.device("cuda", 0)
    T, E = g["rewards"].shape
    f = lambda *s: torch.zeros(*s, device=dev)
    st = dict(cr=f(E), cl=f(E), cc=f(E), dc=torch.ones(E, device=dev), traj13=f(E, 13, 3), pose=f(E, 24, 3), vel=f(E, 2), target=f(E), weight=f(E))
    wp, ip, iv = f(E, 15, 3), f(E, 24, 3), f(E, 2)
    p = lambda t: t.data_ptr()
    s = ops.LocoValStep(E, int(g["step_to_pred"]), float(g["gamma"]), float(g["penalty"]), float(g["min_cum"]), float(g["max_cum"]), p(st["cr"]),
                        p(st["cl"]), p(st["cc"]), p(st["dc"]), p(wp), p(ip), p(iv), p(st["traj13"]), p(st["pose"]), p(st["vel"]), p(st["target"]), p(st["weight"]))
    lib = ops._lib()
    P = lambda t: C.c_void_p(t.data_ptr())
    for t in range(T):
        r, a = torch.from_numpy(g["rewards"][t]).to(dev), torch.from_numpy(g["amp"][t]).to(dev)
        d, inv = torch.from_numpy(g["dones"][t]).to(dev), torch.from_numpy(g["inverted"][t]).to(dev)
        ops._chk(lib.emloco_locoval_returns(C.byref(s), P(r), P(a), P(d), P(inv), current_stream_handle(dev)), "emloco_locoval_returns")
        valid = g["valid"][t]
        assert np.array_equal(st["weight"].cpu().numpy() != 0, valid), t
        assert np.array_equal(st["target"].cpu().numpy()[valid], g["target"][t][valid]), t
        for key, name in (("cc", "cur_combined"), ("dc", "discount"), ("cl", "lengths"), ("cr", "cur_rewards")):
            assert np.array_equal(st[key].cpu().numpy(), g[name][t]), (t, name)
    assert g["valid"].sum() >= 30


def test_locoval_forward_on_the_rows_that_carry_a_target():
    """emloco_locoval_fwd_rows (the fit of a rollout step): rows whose weight is non-zero get exactly what emloco_locoval_fwd gives
    them -- value and the activations the backward reads -- the other rows are left untouched."""
    import ctypes as C
    from emloco_amd.learning.value_pose_net import ValuePoseNet
    from emloco_amd.predictor import ops
    from emloco_amd.sim import current_stream_handle
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    B = 300
    net = ValuePoseNet(True, True).to(dev)._network
    w = [t.detach().contiguous() for t in (net.fc1.weight, net.fc1.bias, net.fc2.weight, net.fc2.bias, net.fc3.weight, net.fc3.bias)]
    traj, pose, vel = torch.randn(B, 13, 3, device=dev), torch.randn(B, 24, 3, device=dev), torch.randn(B, 2, device=dev)
    weight = (torch.rand(B, device=dev) < 0.1).float()
    lib = ops._lib()
    P = lambda t: C.c_void_p(t.data_ptr())
    st = current_stream_handle(dev)

    def bufs(fill):
        return [torch.full(s, fill, device=dev) for s in ((B,), (B, 100), (B, 49), (B, 24), (B,))]
    full, rows = bufs(0.0), bufs(-7.0)
    ops._chk(lib.emloco_locoval_fwd(B, P(traj), 3, P(pose), P(vel), *[P(t) for t in w], *[P(t) for t in full], st), "emloco_locoval_fwd")
    ops._chk(lib.emloco_locoval_fwd_rows(B, P(traj), 3, P(pose), P(vel), *[P(t) for t in w], *[P(t) for t in rows], P(weight), st), "emloco_locoval_fwd_rows")
    torch.cuda.synchronize()
    on = weight != 0
    assert 5 < int(on.sum()) < B
    for a, b in zip(full, rows):
        assert torch.equal(a[on], b[on])
        assert bool((b[~on] == -7.0).all())


def _eval_setup(dev):
    from emloco_amd.learning.value_pose_net import ValuePoseNet
    from emloco_amd.predictor.model_jrdb import TransMotionJRDB
    torch.manual_seed(0)
    cfg = {"DEVICE": dev, "MULTI_MODAL": True, "NOISY_TRAJ": 0, "TRAIN": {"input_track_size": 9, "output_track_size": 12},
           "MODEL": {"value_threshold": 0.8}, "DATA": {"train_datasets": ["jrdb_all_visual_cues"]}}
    model = TransMotionJRDB(tok_dim=246, nhid=128, nhead=4, dim_feedfwd=256, nlayers_local=1, nlayers_global=1, nmode=4, output_scale=1,
                            obs_and_pred=21, num_tokens=26, device=dev, multi_modal=True).to(dev)
    vnet = ValuePoseNet(True, True).to(dev)
    g = torch.Generator().manual_seed(5)
    data = []
    for _ in range(4):
        B, N = 8, 3
        joints = torch.randn(B, N, 21, 26, 4, generator=g) * 0.3
        joints[:, :, :, 0, :2] = torch.cumsum(torch.randn(B, N, 21, 2, generator=g) * 0.4, dim=2)
        pad = torch.zeros(B, N, dtype=torch.bool)
        pad[::3, 2] = True
        data.append((joints, torch.ones(B, N, 21, 26), pad))
    return cfg, model, vnet, data


def _eval_worker(rank, world, port, q):
    _init(rank, world, port)
    import torch.distributed as dist
    from emloco_amd.predictor.evaluate_jta import evaluate_ade_fde
    cfg, model, vnet, data = _eval_setup("cuda:0")
    res = evaluate_ade_fde(model, vnet, "test", "traj+all", data, 8, cfg, dataset="jrdb", random_ids=torch.arange(32) % 4)
    q.put((rank, {k: (np.asarray(v).tolist() if not isinstance(v, (int, float)) else v) for k, v in res.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_evaluation_equals_single_rank():
    """configs[4]: evaluate_ade_fde deals the batches round-robin to the ranks and all-reduces only its summary (sums, counts,
    histograms): both ranks return the numbers of the single-process evaluation of all four batches."""
    from emloco_amd.predictor.evaluate_jta import evaluate_ade_fde
    res = _spawn(_eval_worker, 2)
    r0, r1 = res[0][1], res[1][1]
    cfg, model, vnet, data = _eval_setup("cuda:0")
    single = evaluate_ade_fde(model, vnet, "test", "traj+all", data, 8, cfg, dataset="jrdb", random_ids=torch.arange(32) % 4)
    assert r0["samples"] == r1["samples"] == single["samples"] == 32
    for k, v in single.items():
        a, b, c = np.asarray(r0[k], np.float64), np.asarray(r1[k], np.float64), np.asarray(v, np.float64)
        assert np.array_equal(a, b, equal_nan=True), k
        np.testing.assert_allclose(a, c, rtol=1e-6, atol=1e-9, err_msg=k, equal_nan=True)


# ------------------------------------------------------------------------------------------------ launcher / RCCL smoke
def test_bench_launcher_four_ranks_sharing_the_gpu_prints_one_json_line():
    """`EMLOCO_BENCH_SHARE_GPU=1 python bench.py --gpus 4 --num_envs 256` -- the driver's multi-GPU contract on a one-GPU box: bench.py
    becomes the launcher (torch.distributed.run, 127.0.0.1), four ranks shard the envs, the timed loop is bracketed by barriers with
    MAX-over-ranks timing, the JTA / evaluation legs run data parallel, rank 0 prints ONE JSON line whose value counts all ranks'
    envs.  (gloo instead of RCCL: four processes on one device are refused by the collective library; no scaling number is
    claimed -- the line says TEST MODE.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EMLOCO_BENCH_SHARE_GPU="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--num_envs", "256", "--steps", "20", "--warmup", "5",
                        "--no_cpu_baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["steps"] == 20 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["num_envs_per_gpu"] == 256 and "TEST MODE" in d["config"]["parallelism"]
    assert abs(d["value"] - 4 * 256 * 20 / (d["ms_per_step"] * 20 / 1e3)) < 0.02 * d["value"]      # whole-job aggregate over the four shards
    assert d["jta"]["n_gpus"] == 4 and d["jta"]["value"] > 0 and d["jta"]["eval"]["n_gpus"] == 4
    assert d["config"]["locoval"]["exchange_floats_per_step"] == 6176


def test_bench_launcher_eight_ranks_sharing_the_gpu_runs_the_drivers_rank_count():
    """The driver's `--gpus 8` contract at its REAL rank count, once, before an 8-GPU node appears: eight processes (gloo, one device),
    128 envs each, the rollout legs only (the predictor legs are covered at four ranks above and would put eight models on one GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EMLOCO_BENCH_SHARE_GPU="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--num_envs", "128", "--steps", "12", "--warmup", "4",
                        "--no_cpu_baseline", "--no_jta"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 12 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["num_envs_per_gpu"] == 128 and "x8" in d["config"]["parallelism"] and "TEST MODE" in d["config"]["parallelism"]
    assert abs(d["value"] - 8 * 128 * 12 / (d["ms_per_step"] * 12 / 1e3)) < 0.02 * d["value"]      # whole-job aggregate over the eight shards
    assert d["summary"]["headline_env_steps_per_s"] == d["value"] and list(d)[-1] == "summary"      # the digest closes the line


def _rccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", EMLOCO_FORCE_COLLECTIVES="1")
    import torch.distributed as dist
    from emloco_amd import dist as D
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    from emloco_amd.run import create_rlgpu_env, fill_flags
    from emloco_amd.utils.config import get_args, load_cfg
    D.init_from_env("nccl")                                       # backend "nccl" IS RCCL on ROCm
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1 and D.is_distributed()
    t = torch.arange(8, dtype=torch.float32, device="cuda:0")
    D.all_reduce_(t)                                              # one collective on the caller's stream
    args = get_args(["--num_envs", "64", "--seed", "3", "--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel",
                     "--input_init_pose", "--input_init_vel"])
    cfg, cfg_train, _ = load_cfg(args)
    fill_flags(args)
    env = create_rlgpu_env(args, cfg, cfg_train)
    agent = LocoValRollout(env, horizon_length=16)
    agent.policy = lambda obs: torch.randn(64, 69, device=obs.device) * 0.6
    w0 = torch.cat([p.detach().reshape(-1) for p in agent.valuenet.parameters()]).clone()
    calls = {"n": 0, "side": 0}
    orig = dist.all_reduce

    def counted(tensor, *a, **k):
        calls["n"] += 1
        calls["side"] += int(torch.cuda.current_stream() != torch.cuda.default_stream())
        return orig(tensor, *a, **k)
    dist.all_reduce = counted
    for _ in range(3):
        agent.play_steps()
    n_fit = agent.fitted_episodes
    torch.cuda.synchronize()
    w1 = torch.cat([p.detach().reshape(-1) for p in agent.valuenet.parameters()])
    q.put((0, bool(torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))), calls["n"], calls["side"], n_fit, bool(torch.equal(w0, w1)),
           bool(torch.isfinite(w1).all())))
    dist.destroy_process_group()


def test_rccl_all_reduce_runs_stream_ordered_inside_the_rollout():
    """RCCL smoke (one rank: all a one-GPU box allows): `init_process_group("nccl")` with EMLOCO_FORCE_COLLECTIVES=1, so every
    collective of the product is issued -- the LocoVal rollout's per-step gradient all-reduce runs 48 times ON THE FIT'S SIDE STREAM
    between locoval_reduce and the gated AdamW (stream-ordered: no host wait), the fit still learns (episodes finish, weights move),
    results finite.  The multi-rank semantics are covered with gloo (tests above, tests/test_dist_cpu.py); what this adds is that
    the RCCL path itself has executed."""
    (_, ident, n, side, n_fit, same, finite), = _spawn(_rccl_worker, 1)
    assert ident and n >= 48 and side >= 48 and n_fit > 0 and not same and finite


def _ppo_graph_worker(rank, world, port, q, backend):
    """One AMPAgent epoch pair twice in one process -- eagerly, then with the graphed step -- under an initialised process group."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", EMLOCO_PPO_GRAPH="1")
    if world == 1:
        os.environ["EMLOCO_FORCE_COLLECTIVES"] = "1"
    import yaml
    import torch.distributed as dist
    from emloco_amd import dist as D
    from emloco_amd.learning.amp_agent import AMPAgent
    from emloco_amd.learning.amp_policy import DEFAULT_CFG
    from emloco_amd.run import create_rlgpu_env, fill_flags
    from emloco_amd.utils.config import get_args, load_cfg
    D.init_from_env(backend)
    torch.cuda.set_device(0)
    assert D.is_distributed() and dist.get_backend() == backend
    calls = {"n": 0}
    orig = dist.all_reduce

    def counted(tensor, *a, **k):
        calls["n"] += 1
        return orig(tensor, *a, **k)
    dist.all_reduce = counted
    outs = []
    for graph in (False, True):
        torch.manual_seed(21)
        np.random.seed(21)
        args = get_args(["--num_envs", "64", "--seed", "3", "--random_heading", "--init_heading", "--heading_inversion", "--adjust_root_vel"])
        cfg, _cfg_train, _ = load_cfg(args)
        fill_flags(args)
        env = create_rlgpu_env(args, cfg, _cfg_train, rank=0)       # the SAME shard on every rank: the mean gradient is each rank's own
        cfgt = yaml.safe_load(open(DEFAULT_CFG))
        cfgt["params"]["network"]["mlp"]["units"] = [256, 128]
        cfgt["params"]["network"]["disc"]["units"] = [128, 64]
        cfgt["params"]["config"].update(horizon_length=8, minibatch_size=128, amp_minibatch_size=128, amp_batch_size=64,
                                        amp_obs_demo_buffer_size=512, amp_replay_buffer_size=512, mini_epochs=2)
        agent = AMPAgent(env, cfgt, seed=4)
        agent.use_graph = graph
        torch.manual_seed(33)
        n0 = calls["n"]
        infos = [agent.train_epoch() for _ in range(2)]
        torch.cuda.synchronize()
        steps = 2 * agent.mini_epochs_num * (agent.batch_size // agent.minibatch_size)
        outs.append((torch.cat([p.detach().reshape(-1) for p in agent.a2c_network.parameters()]).cpu().numpy(), float(infos[-1]["actor_loss"]),
                     float(infos[-1]["disc_loss"]), calls["n"] - n0, steps, isinstance(agent._graph, tuple)))      # (numpy: a tensor in the queue is a file descriptor of a process that exits)
    q.put((rank, outs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend,world", [("nccl", 1), ("gloo", 2)])
def test_graphed_ppo_step_runs_data_parallel_around_the_gradient_exchange(backend, world):
    """Round 5: AMPAgent's captured optimiser step at world_size > 1 -- two HIP graphs (losses + backward | clip + Adam + the epoch's
    accumulators) replayed around `bucket.all_reduce` on the same stream (amp_continuous.py:440 `optimizer.synchronize()`).  RCCL with
    one rank (EMLOCO_FORCE_COLLECTIVES=1: every collective is issued; all a one-GPU box allows) and gloo with two ranks sharing the
    GPU on identical shards: the graphed step equals the eagerly issued data-parallel step (weights to float rounding, the epoch's
    losses), it really is the two-graph form, every optimiser step issues its gradient all-reduce, and the two gloo ranks end
    identical."""
    res = _spawn(_ppo_graph_worker, world, backend)
    for _, outs in res:
        (w0, a0, d0, n0, steps, g0), (w1, a1, d1, n1, _, g1) = outs
        assert not g0 and g1
        assert n0 >= steps and n1 >= steps                         # one gradient exchange per optimiser step, graphed or not
        assert np.isfinite(w1).all() and np.abs(w0 - w1).max() <= 2e-5 * np.abs(w0).max() + 1e-6
        assert abs(a0 - a1) <= 5e-3 * abs(a0) + 1e-5 and abs(d0 - d1) <= 5e-3 * abs(d0) + 1e-5
    if world > 1:
        assert np.array_equal(res[0][1][1][0], res[1][1][1][0])
