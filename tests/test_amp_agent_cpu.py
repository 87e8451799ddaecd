"""Host-side pieces of the PPO + AMP learner (emloco_amd/learning/amp_agent.py) that need no GPU: replay buffer, GAE,
Gaussian policy helpers, the per-joint AMP dropout mask.  (Losses / gradients run on the GPU: tests/test_gpu_policy.py.)"""
import math

import numpy as np
import torch

from emloco_amd.learning.amp_agent import AMPAgent, ReplayBuffer, amp_dropout_mask, neglogp, policy_kl


def test_replay_buffer_wraps_and_samples_only_stored_rows():
    torch.manual_seed(0)
    rb = ReplayBuffer(10, "cpu")
    rb.store({"amp_obs": torch.arange(6, dtype=torch.float32).view(6, 1)})
    s = rb.sample(8)["amp_obs"]
    assert s.shape == (8, 1) and set(s.flatten().tolist()) <= set(range(6))          # only what was stored so far
    rb.store({"amp_obs": torch.arange(6, 13, dtype=torch.float32).view(7, 1)})       # wraps: rows 0..2 overwritten by 10..12
    assert rb.get_total_count() == 13 and rb._head == 3
    assert rb._data_buf["amp_obs"].flatten().tolist() == [10, 11, 12, 3, 4, 5, 6, 7, 8, 9]
    seen = set()
    for _ in range(5):
        seen |= set(rb.sample(4)["amp_obs"].flatten().tolist())                     # sample head passes the end -> reshuffle
    assert seen <= set(range(3, 13))


def test_gae_matches_the_recursion_definition():
    agent = AMPAgent.__new__(AMPAgent)
    agent.horizon_length, agent.gamma, agent.tau = 5, 0.99, 0.95
    g = torch.Generator().manual_seed(1)
    H, E = 5, 7
    rewards, values, next_values = torch.randn(H, E, 1, generator=g), torch.randn(H, E, 1, generator=g), torch.randn(H, E, 1, generator=g)
    dones = (torch.rand(H, E, generator=g) < 0.3).float()
    adv = agent.discount_values(dones, values, rewards, next_values)
    ref = torch.zeros(H, E, 1)
    for e in range(E):
        last = 0.0
        for t in reversed(range(H)):
            delta = rewards[t, e, 0] + 0.99 * next_values[t, e, 0] - values[t, e, 0]
            last = delta + 0.99 * 0.95 * (1 - dones[t, e]) * last
            ref[t, e, 0] = last
    torch.testing.assert_close(adv, ref)


def test_gaussian_policy_helpers_match_torch_distributions():
    g = torch.Generator().manual_seed(2)
    mu, x = torch.randn(6, 69, generator=g), torch.randn(6, 69, generator=g)
    logstd = torch.full((69,), -2.9)
    sigma = torch.exp(logstd)
    nlp = neglogp(x, mu, sigma, logstd)
    ref = -torch.distributions.Normal(mu, sigma).log_prob(x).sum(-1)
    torch.testing.assert_close(nlp, ref, rtol=1e-5, atol=1e-3)
    mu2, sig2 = mu + 0.01 * torch.randn(6, 69, generator=g), sigma * 1.1
    kl = policy_kl(mu, sigma.expand_as(mu), mu2, sig2.expand_as(mu))
    ref_kl = torch.distributions.kl_divergence(torch.distributions.Normal(mu, sigma), torch.distributions.Normal(mu2, sig2)).sum(-1).mean()
    assert abs(kl.item() - ref_kl.item()) < 0.05 * abs(ref_kl.item()) + 5e-2       # rl_games adds 1e-5 regularisers


def test_amp_dropout_mask_drops_whole_joints_consistently_over_steps():
    torch.manual_seed(3)
    m = amp_dropout_mask(64, 15, 206)
    assert m.shape == (64, 15 * 206, 3) and set(m.unique().tolist()) <= {0.0, 1.0}
    row = m.view(64, 15, 206, 3)
    assert torch.equal(row[:, 0], row[:, 7])                                          # the same mask on every history step
    assert (row[:, 0, :12] == 1).all() and (row[:, 0, 12 + 19 * 6 + 19 * 3:] == 1).all()   # root terms, key bodies, betas are never dropped
    rot = row[:, 0, 12:12 + 19 * 6].reshape(64, 19, 6, 3)
    vel = row[:, 0, 12 + 19 * 6:12 + 19 * 9].reshape(64, 19, 3, 3)
    assert (rot == rot[:, :, :1]).all() and (vel == rot[:, :, :3]).all()              # a joint's 6 rotation + 3 velocity values share one draw
    assert 0.6 < rot.mean().item() < 0.8                                              # keep probability 0.7


def test_amp_dropout_mask_draws_the_reference_stream():
    """The mask is assembled on the device from ONE host draw; that draw must be the stream of the reference's 19 successive
    `torch.rand(B, num_masks)` calls (amp_models.py:85-89), whatever the batch size."""
    for B in (2048, 2560, 64, 37):
        torch.manual_seed(5)
        m = amp_dropout_mask(B, 3, 206)
        torch.manual_seed(5)
        ref = torch.ones(B, 206, 3)
        for j in range(19):
            keep = (torch.rand(B, 3) > 0.3).float()
            ref[:, 12 + j * 6:12 + j * 6 + 6, :] = keep[:, None]
            ref[:, 126 + j * 3:126 + j * 3 + 3, :] = keep[:, None]
        assert torch.equal(m, ref.repeat(1, 3, 1)), B
