"""GPU parity of the policy / critic / discriminator network against golden vectors
produced by the reference's own AMPSeptBuilder.Network + RunningMeanStd (tests/golden/gen_golden_policy.py), and of
the frozen-policy fast path against a plain torch fp32 evaluation at the full network widths."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"


def _params(units, task_units, disc_units):
    import yaml
    from emloco_amd import __file__ as pkg
    path = os.path.join(os.path.dirname(pkg), "data", "cfg", "train", "rlg", "amp_humanoid_smpl_sept_task.yaml")
    p = yaml.safe_load(open(path))["params"]["network"]
    p["mlp"]["units"], p["task_mlp"]["units"], p["disc"]["units"] = list(units), list(task_units), list(disc_units)
    return p


def _build(g=None, units=(2048, 1024), task_units=(512, 256), disc_units=(1024, 512), amp=3090):
    from emloco_amd.learning.amp_network_sept_builder import AMPSeptBuilder
    from emloco_amd.utils.running_mean_std import RunningMeanStd
    rms = RunningMeanStd((1422,)).to(DEV)
    rms.eval()
    b = AMPSeptBuilder()
    b.load(_params(units, task_units, disc_units))
    net = b.build("amp", actions_num=69, input_shape=(1422,), num_seqs=1, value_size=1, amp_input_shape=(amp,),
                  self_obs_size=368, task_obs_size=1054, task_obs_size_detail={"traj": 30, "heightmap": 1024}, mean_std=rms).to(DEV)
    if g is not None:
        net.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
        rms.running_mean.copy_(torch.from_numpy(g["running_mean"]))
        rms.running_var.copy_(torch.from_numpy(g["running_var"]))
    return net, rms


def _close(a, b, rel=1e-4, abs_=1e-5, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    tol = abs_ + rel * np.abs(b).max()
    assert np.abs(a - b).max() <= tol, f"{what}: max err {np.abs(a - b).max():.3e} > {tol:.3e}"


def test_network_matches_reference_golden(golden):
    g = golden("policy_net")
    net, rms = _build(g, g["units"], g["task_units"], g["disc_units"], amp=int(g["sizes"][2]))
    assert [k for k in net.state_dict()] == [k[3:] for k in g if k.startswith("sd.")]      # same keys, same order
    obs = torch.from_numpy(g["obs"]).to(DEV)
    amp_obs = torch.from_numpy(g["amp_obs"]).to(DEV)
    nobs = rms(obs)
    _close(nobs.cpu(), g["norm_obs"], rel=2e-6, abs_=2e-6, what="normalised obs")
    mu, sigma = net.eval_actor(nobs)
    _close(mu.detach().cpu(), g["mu"], what="mu")
    assert np.array_equal(sigma.detach().cpu().numpy(), g["sigma"])
    _close(net.eval_task(nobs[:, 368:]).detach().cpu(), g["task_out"], what="task embedding")
    _close(net.eval_critic(nobs).detach().cpu(), g["value"], what="value")
    _close(net.eval_disc(amp_obs).detach().cpu(), g["disc_logits"], what="disc logits")
    out = net({"obs": nobs})
    assert len(out) == 4 and out[3] is None
    # backward through actor + critic + discriminator (what config 2 trains)
    loss = (net.eval_actor(nobs)[0] ** 2).mean() + net.eval_critic(nobs).mean() + (net.eval_disc(amp_obs) ** 2).mean()
    net.zero_grad()
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    for k in g:
        if k.startswith("grad."):
            p = dict(net.named_parameters())[k[5:]]
            _close(p.grad.cpu(), g[k], rel=2e-4, abs_=1e-6, what=k)


def test_frozen_policy_fast_path(golden):
    from emloco_amd.learning.policy_runner import FrozenPolicy
    g = golden("policy_net")
    net, rms = _build(g, g["units"], g["task_units"], g["disc_units"], amp=int(g["sizes"][2]))
    obs = torch.from_numpy(g["obs"]).to(DEV)
    pol = FrozenPolicy(net, rms, obs.shape[0], DEV)
    _close(pol.act_mean(obs).cpu(), g["mu"], what="frozen-policy mu (reduced widths, reference golden)")
    a = pol.act(obs, deterministic=True)
    assert float(a.abs().max()) <= 1.0
    gen = torch.Generator(device=DEV).manual_seed(3)
    s = pol.act(obs, deterministic=False, generator=gen)
    assert s.shape == a.shape and float(s.abs().max()) <= 1.0 and not torch.equal(s, a)


def test_frozen_policy_full_width_vs_torch_fp32():
    """4096 envs, the shipped widths (1054->512->256, 624->2048->1024->69): fp32 MFMA path vs torch fp32 (1e-4 rel)."""
    from emloco_amd.learning.policy_runner import FrozenPolicy
    torch.manual_seed(0)
    net, rms = _build()
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1 and p.shape[0] != 69:
                p.copy_(torch.randn_like(p) * 0.1)
        rms.running_mean.copy_(torch.randn(1422, dtype=torch.float64) * 0.2)
        rms.running_var.copy_(torch.rand(1422, dtype=torch.float64) + 0.1)
    E = 4096
    obs = torch.randn(E, 1422, device=DEV)
    pol = FrozenPolicy(net, rms, E, DEV)
    mu = pol.act_mean(obs)
    F = torch.nn.functional
    with torch.no_grad():
        x = torch.clamp((obs - rms.running_mean.float()) / torch.sqrt(rms.running_var.float() + rms.epsilon), -5, 5)
        sd = net.state_dict()
        t = F.relu(F.linear(x[:, 368:], sd["_task_mlp.0.weight"], sd["_task_mlp.0.bias"]))
        t = F.relu(F.linear(t, sd["_task_mlp.2.weight"], sd["_task_mlp.2.bias"]))
        h = F.relu(F.linear(torch.cat([x[:, :368], t], 1), sd["actor_mlp.0.weight"], sd["actor_mlp.0.bias"]))
        h = F.relu(F.linear(h, sd["actor_mlp.2.weight"], sd["actor_mlp.2.bias"]))
        ref = F.linear(h, sd["mu.weight"], sd["mu.bias"])
    _close(mu.cpu(), ref.cpu(), what="full-width mu")
    assert pol.flops_per_env == 2 * (1056 * 512 + 512 * 256 + 624 * 2048 + 2048 * 1024 + 1024 * 69)
    # round 5: the frozen weights are read as piece images (cut once at first use: policy_runner._frozen_image) -- the same bits as the matrices
    from emloco_amd.predictor import ops
    if ops._matmul_precision[0] == "fp32_split":
        assert len(pol._images) == 5
        with_images = mu.clone()
        ops._IMAGE_FROZEN = False
        try:
            assert torch.equal(FrozenPolicy(net, rms, E, DEV).act_mean(obs), with_images)
        finally:
            ops._IMAGE_FROZEN = True


class _FakeTask:
    """just enough of the task interface for AMPAgent's constructor and losses (no simulator)"""
    def __init__(self, E=8, dev=DEV):
        self.num_envs, self.device, self.num_actions = E, dev, 69
        self._num_amp_obs_steps = 2
        self.motion_sym_loss = True
        self.cfg = {"env": {}}
        self.left_to_right_index_action = [4, 5, 6, 7, 0, 1, 2, 3, 8, 9, 10, 11, 12, 18, 19, 20, 21, 22, 13, 14, 15, 16, 17]
        self.task = self

    def get_obs_size(self): return 1422
    def get_self_obs_size(self): return 368
    def get_num_amp_obs(self): return 412
    def get_task_obs_size_detail(self): return {"traj": 30, "heightmap": 1024}
    def fetch_amp_obs_demo(self, n): return torch.randn(n, 412, device=self.device)


def _agent(E=8):
    import yaml
    from emloco_amd.learning.amp_agent import AMPAgent
    from emloco_amd.learning.amp_policy import DEFAULT_CFG
    cfg = yaml.safe_load(open(DEFAULT_CFG))
    cfg["params"]["network"]["mlp"]["units"] = [96, 48]
    cfg["params"]["network"]["task_mlp"]["units"] = [64, 32]
    cfg["params"]["network"]["disc"]["units"] = [80, 40]
    c = cfg["params"]["config"]
    c.update(horizon_length=4, minibatch_size=16, amp_minibatch_size=16, amp_batch_size=32, amp_obs_demo_buffer_size=64,
             amp_replay_buffer_size=64, mini_epochs=1)
    return AMPAgent(_FakeTask(E), cfg)


def test_amp_ppo_losses_match_plain_torch_double_backward():
    """calc_gradients' scalar and every parameter gradient against a plain-torch restatement that takes the discriminator
    gradient penalty by true double backward (torch.autograd.grad(create_graph=True), amp_continuous.py:560-583)."""
    import math
    from emloco_amd.learning.amp_agent import amp_dropout_mask
    agent = _agent()
    net = agent.a2c_network
    torch.manual_seed(1)
    with torch.no_grad():
        for p in net.parameters():
            if p.requires_grad:
                p.add_(torch.randn_like(p) * 0.05)
    agent.set_eval()                                   # keep the running statistics fixed for the comparison
    B = 16
    d = {"obs": torch.randn(B, 1422, device=DEV), "flip_obs": torch.randn(B, 1422, device=DEV), "next_obses": torch.randn(B, 1422, device=DEV),
         "actions": torch.randn(B, 69, device=DEV) * 0.3, "old_logp_actions": torch.randn(B, device=DEV) * 0.1 + 30,
         "advantages": torch.randn(B, device=DEV), "old_values": torch.randn(B, 1, device=DEV), "returns": torch.randn(B, 1, device=DEV),
         "mu": torch.randn(B, 69, device=DEV) * 0.1, "sigma": torch.full((B, 69), math.exp(-2.9), device=DEV),
         "amp_obs": torch.randn(B, 412, device=DEV), "amp_obs_replay": torch.randn(B, 412, device=DEV), "amp_obs_demo": torch.randn(B, 412, device=DEV)}
    masks = amp_dropout_mask(B, 2, 206, device=DEV)
    loss, info, mu, sigma = agent.compute_loss(d, dropout_masks=masks)
    agent.bucket.zero()
    loss.backward()
    got = {k: p.grad.clone() for k, p in net.named_parameters() if p.requires_grad}

    # ---- plain torch restatement (F.linear, autograd double backward)
    F = torch.nn.functional
    P = {k: p.detach().clone().requires_grad_(p.requires_grad) for k, p in net.named_parameters()}

    def mlp(x, name, n):
        for i in range(n):
            x = F.relu(F.linear(x, P[f"{name}.{2 * i}.weight"], P[f"{name}.{2 * i}.bias"]))
        return x

    def actor(x):
        t = mlp(x[:, 368:], "_task_mlp", 2)
        return F.linear(mlp(torch.cat([x[:, :368], t], 1), "actor_mlp", 2), P["mu.weight"], P["mu.bias"])

    def critic(x):
        t = mlp(x[:, 368:], "_task_mlp", 2)
        return F.linear(mlp(torch.cat([x[:, :368], t], 1), "critic_mlp", 2), P["value.weight"], P["value.bias"])

    def disc(x):
        return F.linear(mlp(x, "_disc_mlp", 2), P["_disc_logits.weight"], P["_disc_logits.bias"])

    obs = d["obs"]                                      # statistics are at their initial (0, 1) values: normalisation = clamp
    nrm = lambda x: torch.clamp(x / math.sqrt(1 + 1e-5), -5, 5)
    mu_r = actor(nrm(obs))
    logstd = P["sigma"]
    sig = torch.exp(logstd)
    nlp = 0.5 * (((d["actions"] - mu_r) / sig) ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * 69 + logstd.sum(-1) + mu_r.sum(-1) * 0
    ratio = torch.exp(d["old_logp_actions"] - nlp)
    a_loss = torch.max(-d["advantages"] * ratio, -d["advantages"] * torch.clamp(ratio, 0.8, 1.2)).mean()
    c_loss = ((d["returns"] - critic(nrm(obs))) ** 2).mean()
    b_loss = (torch.clamp_max(mu_r + 1, 0) ** 2 + torch.clamp_min(mu_r - 1, 0) ** 2).sum(-1).mean()
    demo = nrm(d["amp_obs_demo"]).requires_grad_(True)
    agent_logit = torch.cat([disc(nrm(d["amp_obs"]) * masks[..., 0]), disc(nrm(d["amp_obs_replay"]) * masks[..., 1])], 0)
    demo_logit = disc(demo * masks[..., 2])
    bce = torch.nn.BCEWithLogitsLoss()
    dl = 0.5 * (bce(agent_logit, torch.zeros_like(agent_logit)) + bce(demo_logit, torch.ones_like(demo_logit)))
    dl = dl + 0.01 * (P["_disc_logits.weight"] ** 2).sum()
    (gx,) = torch.autograd.grad(demo_logit, demo, grad_outputs=torch.ones_like(demo_logit), create_graph=True, retain_graph=True)
    dl = dl + 5 * gx.pow(2).sum(-1).mean()
    dl = dl + 0.0001 * sum((P[k] ** 2).sum() for k in ("_disc_mlp.0.weight", "_disc_mlp.2.weight", "_disc_logits.weight"))
    flip_a = actor(nrm(d["flip_obs"]))
    orig_a = (actor(nrm(d["next_obses"])).view(B, -1, 3) * torch.tensor([-1.0, 1.0, -1.0], device=DEV))[:, agent.task.left_to_right_index_action]
    s_loss = ((orig_a.reshape(B, -1) - flip_a) ** 2).mean(-1).mean() * 50
    ref = a_loss + 5 * c_loss + 10 * b_loss + 5 * dl + s_loss
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item()), (loss.item(), ref.item())
    for k, g in got.items():
        _close(g.cpu(), P[k].grad.cpu(), rel=3e-4, abs_=1e-6, what=f"grad {k}")
    assert abs(info["disc_grad_penalty"].item() - gx.pow(2).sum(-1).mean().item()) <= 1e-4 * gx.pow(2).sum(-1).mean().item()


def test_round6_step_variants_agree(monkeypatch):
    """Round 6's forms of the optimiser step's loss -- one stacked actor evaluation, weight gradients written straight into the flat bucket,
    the discriminator's operands born padded (an AMP width that is NOT a multiple of four, like the shipped 3 090), the one-launch
    ReLU-mask op of the gradient-penalty network -- give the loss and every gradient of the reference-shaped step (three actor evaluations,
    autograd's accumulation, F.pad per call) on the same weights and minibatch."""
    import math
    from emloco_amd.learning import amp_agent as A
    from emloco_amd.predictor import ops

    class Task(_FakeTask):
        def get_num_amp_obs(self): return 2 * 207                           # 414: not a multiple of four (two steps of a 207-wide row)
        def fetch_amp_obs_demo(self, n): return torch.randn(n, 414, device=self.device)

    def make(stack, direct, padded):
        import yaml
        from emloco_amd.learning.amp_policy import DEFAULT_CFG
        monkeypatch.setenv("EMLOCO_PPO_STACK_ACTOR", stack); monkeypatch.setenv("EMLOCO_PPO_DIRECT_GRAD", direct); monkeypatch.setenv("EMLOCO_PPO_DISC_PADDED", padded)
        cfg = yaml.safe_load(open(DEFAULT_CFG))
        cfg["params"]["network"]["mlp"]["units"] = [96, 48]
        cfg["params"]["network"]["task_mlp"]["units"] = [64, 32]
        cfg["params"]["network"]["disc"]["units"] = [80, 40]
        cfg["params"]["config"].update(horizon_length=4, minibatch_size=16, amp_minibatch_size=16, amp_batch_size=32, amp_obs_demo_buffer_size=64,
                                       amp_replay_buffer_size=64, mini_epochs=1)
        return A.AMPAgent(Task(8), cfg)

    ref_agent = make("0", "0", "0")
    new_agent = make("1", "1", "1")
    new_agent.a2c_network.load_state_dict(ref_agent.a2c_network.state_dict())
    assert any(getattr(p_, "_emloco_direct_grad", False) for p_ in new_agent.a2c_network.parameters())
    assert not any(getattr(p_, "_emloco_direct_grad", False) for p_ in ref_agent.a2c_network.parameters())
    torch.manual_seed(3)
    B = 16
    d = {"obs": torch.randn(B, 1422, device=DEV), "flip_obs": torch.randn(B, 1422, device=DEV), "next_obses": torch.randn(B, 1422, device=DEV),
         "actions": torch.randn(B, 69, device=DEV) * 0.3, "old_logp_actions": torch.randn(B, device=DEV) * 0.1 + 30,
         "advantages": torch.randn(B, device=DEV), "old_values": torch.randn(B, 1, device=DEV), "returns": torch.randn(B, 1, device=DEV),
         "mu": torch.randn(B, 69, device=DEV) * 0.1, "sigma": torch.full((B, 69), math.exp(-2.9), device=DEV),
         "amp_obs": torch.randn(B, 414, device=DEV), "amp_obs_replay": torch.randn(B, 414, device=DEV), "amp_obs_demo": torch.randn(B, 414, device=DEV)}
    masks = (torch.rand(B, 414, 3, device=DEV) > 0.3).float()
    out = []
    for ag in (ref_agent, new_agent):
        ag.set_eval()
        loss, info, mu, sigma = ag.compute_loss(d, dropout_masks=masks)
        ag.bucket.zero()
        loss.backward()
        out.append((loss.item(), {k: v.item() for k, v in info.items()}, {k: p_.grad.clone() for k, p_ in ag.a2c_network.named_parameters() if p_.requires_grad}))
    (l0, i0, g0), (l1, i1, g1) = out
    assert abs(l0 - l1) <= 2e-5 * abs(l0), (l0, l1)
    for k in i0:
        assert abs(i0[k] - i1[k]) <= 1e-4 * max(abs(i0[k]), 1e-3), (k, i0[k], i1[k])
    for k in g0:
        _close(g1[k].cpu(), g0[k].cpu(), rel=2e-4, abs_=1e-6, what=f"grad {k}")
    # the ReLU-mask op against its torch expression, value and gradient
    t = torch.randn(33, 40, device=DEV, requires_grad=True)
    hh = torch.randn(33, 40, device=DEV)
    y = ops.relu_mask(t, hh)
    assert torch.equal(y, (hh > 0).float() * t)
    (gt,) = torch.autograd.grad(y.sum() * 2.0, t)
    assert torch.equal(gt, (hh > 0).float() * 2.0)


def test_running_mean_std_training_update_matches_the_reference_class():
    """RunningMeanStd in training mode on the device (normalise with the old moments, then emloco_rms_update) against the
    reference's own class over four batches incl. freeze_partial (tests/golden/gen_golden_rms.py): outputs and moments to 2e-6 (the reference takes
    the batch mean / variance in float32 before merging, the kernel accumulates in float64), count exact; eval mode leaves the moments alone; a CPU batch in training mode is refused (no CPU path)."""
    import os
    from emloco_amd import _lib as L
    from emloco_amd.utils.running_mean_std import RunningMeanStd
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "running_mean_std.npz"))
    cols = g["x0"].shape[1]
    rms = RunningMeanStd((cols,)).to(DEV)
    rms.train()
    for i in range(int(g["n_batches"])):
        if i == 2:
            rms.freeze_partial(3)
        y = rms(torch.from_numpy(g[f"x{i}"]).to(DEV))
        np.testing.assert_allclose(y.cpu().numpy(), g[f"y{i}"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(rms.running_mean.cpu().numpy(), g[f"mean{i}"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(rms.running_var.cpu().numpy(), g[f"var{i}"], rtol=2e-6, atol=2e-6)
        assert float(rms.count) == float(g[f"count{i}"])
    rms.eval()
    before = rms.running_mean.clone()
    rms(torch.from_numpy(g["x0"]).to(DEV))
    assert torch.equal(before, rms.running_mean)
    cpu = RunningMeanStd((cols,))
    cpu.train()
    with pytest.raises(L.EmlocoError):
        cpu(torch.from_numpy(g["x0"]))


def test_minibatch_gather_of_all_tables_in_one_launch():
    """`ppo_heads.gather_rows` (emloco_ppo_gather_rows): the learner's minibatch gather -- every fp32 tensor of the dataset indexed with
    the same shuffled row ids (amp_datasets.py:16-33) -- as one launch: equal to torch.index_select per tensor, for row widths with and
    without 16-byte alignment, 1-D tables, more than 16 tables (two launches) and repeated ids."""
    from emloco_amd.learning import ppo_heads
    torch.manual_seed(9)
    N, n = 5000, 777
    widths = [1422, 69, 1, 3090, 7, 412] + [5] * 13
    srcs = [torch.randn(N, w, device=DEV) if w != 1 else torch.randn(N, device=DEV) for w in widths]
    idx = torch.randint(0, N, (n,), device=DEV)
    idx[10] = idx[11]
    dsts = [torch.full((n,) + tuple(s.shape[1:]), float("nan"), device=DEV) for s in srcs]
    ppo_heads.gather_rows(idx, srcs, dsts)
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert torch.equal(d, torch.index_select(s, 0, idx))
