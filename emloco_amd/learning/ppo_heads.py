"""The PPO + AMP learner's loss heads as fused HIP launches (include/emloco_predictor.h: emloco_ppo_*_head_*).

`calc_gradients` (amp_continuous.py:335-425) ends in ~140 small torch launches per optimiser step between the networks' outputs and the
scalar loss -- neglogp, the clipped surrogate, the entropy, the bound loss, the policy KL; the clipped value loss; the discriminator's two
binary cross entropies and its accuracies -- and as many again in their backward.  Each head here is an autograd.Function over one
forward launch + a fixed-order mean, and one backward launch.  CUDA tensors only: the product has no CPU arithmetic; `learning/amp_agent.py`
keeps the torch expressions (the restatement the CPU tests pin to rl_games' formulas) for CPU tensors and `EMLOCO_PPO_HEADS=0`.
"""
import ctypes as C

import torch

from ..predictor import ops

_BOUND = False


def _lib():
    global _BOUND
    lib = ops._lib()
    if not _BOUND:
        vp, cf, ci = C.c_void_p, C.c_float, C.c_int
        lib.emloco_ppo_actor_head_fwd.argtypes = [ci, ci] + [vp] * 7 + [cf, vp, vp, vp]
        lib.emloco_ppo_actor_head_bwd.argtypes = [ci, ci] + [vp] * 5 + [cf, vp, vp, vp, vp]
        lib.emloco_ppo_critic_head_fwd.argtypes = [ci, vp, vp, vp, cf, ci, vp, vp, vp]
        lib.emloco_ppo_critic_head_bwd.argtypes = [ci, vp, vp, vp, cf, ci, vp, vp, vp]
        lib.emloco_ppo_disc_head_fwd.argtypes = [ci, ci, vp, vp, vp, vp, vp]
        lib.emloco_ppo_disc_head_bwd.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp]
        lib.emloco_ppo_gather_rows.argtypes = [ci, ci, vp, vp, vp, vp, vp]
        _BOUND = True
    return lib


def gather_rows(idx, srcs, dsts):
    """dst[r] = src[idx[r]] for every (src, dst) pair of 2-D-viewable contiguous fp32 CUDA tensors, in one launch (<= 16 pairs per launch)."""
    assert idx.dtype == torch.int64 and idx.is_cuda and idx.is_contiguous()
    n = idx.shape[0]
    pairs = list(zip(srcs, dsts))
    for o in range(0, len(pairs), 16):
        part = pairs[o:o + 16]
        for s_, d_ in part:
            assert s_.dtype == torch.float32 and d_.dtype == torch.float32 and s_.is_contiguous() and d_.is_contiguous() and d_.shape[0] == n
        k = len(part)
        src = (C.c_void_p * k)(*[t[0].data_ptr() for t in part])
        dst = (C.c_void_p * k)(*[t[1].data_ptr() for t in part])
        cols = (C.c_int * k)(*[max(t[0].numel() // max(t[0].shape[0], 1), 1) for t in part])
        ops._chk(_lib().emloco_ppo_gather_rows(k, n, _p(idx), src, dst, cols, ops._st(idx)), "emloco_ppo_gather_rows")


def _c(t):
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _ActorHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logstd, actions, old_neglogp, advantages, old_mu, old_sigma, e_clip):
        mu_, ls_, ac_, on_, ad_ = _c(mu), _c(logstd), _c(actions), _c(old_neglogp).reshape(-1), _c(advantages).reshape(-1)
        om_, os_ = (None, None) if old_mu is None else (_c(old_mu), _c(old_sigma))
        B, A = mu_.shape
        rows = torch.empty((B, 5), dtype=torch.float32, device=mu.device)
        out = torch.empty(5, dtype=torch.float32, device=mu.device)
        ops._chk(_lib().emloco_ppo_actor_head_fwd(B, A, _p(mu_), _p(ls_), _p(ac_), _p(on_), _p(ad_), _p(om_), _p(os_), float(e_clip),
                                                 _p(rows), _p(out), ops._st(mu_)), "emloco_ppo_actor_head_fwd")
        ctx.save_for_backward(mu_, ls_, ac_, on_, ad_)
        ctx.e_clip = float(e_clip)
        ctx.need_ls = logstd.requires_grad
        o = out.unbind(0)
        ctx.mark_non_differentiable(o[3], o[4])
        return o

    @staticmethod
    def backward(ctx, ga, ge, gb, _gc, _gk):
        mu_, ls_, ac_, on_, ad_ = ctx.saved_tensors
        B, A = mu_.shape
        z = torch.zeros((), dtype=torch.float32, device=mu_.device)
        g = torch.stack([z if ga is None else ga.float().reshape(()), z if ge is None else ge.float().reshape(()),
                         z if gb is None else gb.float().reshape(())])
        dmu = torch.empty_like(mu_)
        dls = torch.empty_like(ls_) if ctx.need_ls else None
        ops._chk(_lib().emloco_ppo_actor_head_bwd(B, A, _p(mu_), _p(ls_), _p(ac_), _p(on_), _p(ad_), ctx.e_clip, _p(g), _p(dmu), _p(dls),
                                                 ops._st(mu_)), "emloco_ppo_actor_head_bwd")
        return dmu, dls, None, None, None, None, None, None


def actor_head(mu, logstd, actions, old_neglogp, advantages, e_clip, old_mu=None, old_sigma=None):
    """(mean surrogate, mean entropy, mean bound loss, clipped fraction, KL(new || old)) -- the last two carry no gradient."""
    return _ActorHead.apply(mu, logstd, actions, old_neglogp, advantages, old_mu, old_sigma, e_clip)


class _CriticHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values, old_values, returns, e_clip, clip_value):
        v_, vo_, r_ = _c(values).reshape(-1), _c(old_values).reshape(-1), _c(returns).reshape(-1)
        B = v_.shape[0]
        rows = torch.empty(B, dtype=torch.float32, device=values.device)
        out = torch.empty(1, dtype=torch.float32, device=values.device)
        ops._chk(_lib().emloco_ppo_critic_head_fwd(B, _p(v_), _p(vo_), _p(r_), float(e_clip), int(bool(clip_value)), _p(rows), _p(out),
                                                  ops._st(v_)), "emloco_ppo_critic_head_fwd")
        ctx.save_for_backward(v_, vo_, r_)
        ctx.cfg = (float(e_clip), int(bool(clip_value)), tuple(values.shape))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        v_, vo_, r_ = ctx.saved_tensors
        e_clip, clip_value, shape = ctx.cfg
        dv = torch.empty_like(v_)
        ops._chk(_lib().emloco_ppo_critic_head_bwd(v_.shape[0], _p(v_), _p(vo_), _p(r_), e_clip, clip_value, _p(_c(g).reshape(1)), _p(dv),
                                                  ops._st(v_)), "emloco_ppo_critic_head_bwd")
        return dv.view(shape), None, None, None, None


def critic_head(values, old_values, returns, e_clip, clip_value):
    return _CriticHead.apply(values, old_values, returns, e_clip, clip_value)


class _DiscHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, agent_logits, demo_logits):
        a_, d_ = _c(agent_logits).reshape(-1), _c(demo_logits).reshape(-1)
        na, nd = a_.shape[0], d_.shape[0]
        rows = torch.empty((na + nd, 2), dtype=torch.float32, device=a_.device)
        out = torch.empty(4, dtype=torch.float32, device=a_.device)
        ops._chk(_lib().emloco_ppo_disc_head_fwd(na, nd, _p(a_), _p(d_), _p(rows), _p(out), ops._st(a_)), "emloco_ppo_disc_head_fwd")
        ctx.save_for_backward(a_, d_)
        ctx.shapes = (tuple(agent_logits.shape), tuple(demo_logits.shape))
        o = out.unbind(0)
        ctx.mark_non_differentiable(o[1], o[3])
        return o

    @staticmethod
    def backward(ctx, ga, _a, gd, _d):
        a_, d_ = ctx.saved_tensors
        z = torch.zeros((), dtype=torch.float32, device=a_.device)
        g = torch.stack([z if ga is None else ga.float().reshape(()), z if gd is None else gd.float().reshape(())])
        da, dd = torch.empty_like(a_), torch.empty_like(d_)
        ops._chk(_lib().emloco_ppo_disc_head_bwd(a_.shape[0], d_.shape[0], _p(a_), _p(d_), _p(g), _p(da), _p(dd), ops._st(a_)),
                 "emloco_ppo_disc_head_bwd")
        return da.view(ctx.shapes[0]), dd.view(ctx.shapes[1])


def disc_head(agent_logits, demo_logits):
    """(mean BCE(agent, 0), agent accuracy, mean BCE(demo, 1), demo accuracy) -- the accuracies carry no gradient."""
    return _DiscHead.apply(agent_logits, demo_logits)
