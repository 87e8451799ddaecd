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
