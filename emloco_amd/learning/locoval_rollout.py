"""LocoVal training rollout: the bookkeeping of AMPValueAgent.play_steps without rl_games.

Mirror of pacer/pacer/learning/amp_continuous_value.py:34-178 (play_steps) and common_agent.py:89-97,154-155,205-209
(LocoVal optimiser, cosine schedule, target normalisation): per step reset finished envs, act, step, apply the inversion
penalty, accumulate the discounted reward `shaped task reward + discriminator reward` per env up to `step_to_pred` control
steps, and when episodes finish fit LocoVal by sum-MSE on (waypoints, initial pose, initial velocity) -> normalised return
with AdamW(1e-3, wd 1e-4).  The sum is UNWEIGHTED (:96): task_reward_w / disc_reward_w belong to the PPO path
(`_combine_rewards`), not to the LocoVal target; `reward_shaper.scale_value` is 1 (amp_humanoid_smpl_sept_task.yaml:83-84).
The policy is pluggable (`policy(obs) -> actions`; default: the frozen policy's exploration noise N(0, e^-2.9)), so is the
AMP discriminator reward (`disc_reward(amp_obs) -> (E,)`, default 0).

Host synchronisation: the reference reads `dones.nonzero()` and `len(valid_rewards_idx[0])` on the host every step
(:98,123-124).  Here finished envs are reset by the device-side compaction (`reset_done`), the bookkeeping is mask
arithmetic, the fit runs as a masked sum over all envs, and AdamW commits its update through a device-side gate on the
(all-reduced) number of finished episodes (`flat_adamw.GatedFlatAdamW`): the host reads nothing during the rollout; the loss /
episode counters are device tensors read when someone asks for them (`vnet_loss`, `fitted_episodes`, ...).

Multi-GPU: every rank issues exactly one all-reduce per rollout step, whatever its own episodes did -- the flat gradient
bucket (6 174 floats) with the loss sum and the episode count in its tail; gradients are sum-reduced and the loss is
divided by the global episode count (MSELoss(reduction='sum') semantics, common_agent.py:96).  All ranks step together
when the global count is positive, so the replicas (broadcast from rank 0 at construction) stay identical.
"""
import math

import os
import torch

from ..dist import FlatGradBucket, broadcast_parameters
from .flat_adamw import GatedFlatAdamW
from .scheduler import CosineAnnealingLR
from .value_pose_net import ValuePoseNet


class _ReturnState:
    """Device buffers of the per-env discounted return (amp_continuous_value.py:93-118): updated in place by
    `locoval_returns_kernel` (csrc/predictor_kernels.hip).  The torch restatement of that arithmetic is test infrastructure
    (the test oracle, module locoval_returns), pinned by a fixture generated from the reference's own play_steps."""

    def __init__(self, num_envs, step_to_pred, gamma, device):
        z = lambda: torch.zeros(num_envs, device=device)
        self.step_to_pred, self.gamma = step_to_pred, gamma
        self.current_rewards, self.current_lengths, self.current_combined_rewards = z(), z(), z()
        self.discount_coefs = torch.ones(num_envs, device=device)


class LocoValRollout:
    def __init__(self, vec_env, use_pose=True, use_vel=True, horizon_length=32, gamma=0.99, inversion_penalty_scale=0.3,
                 policy=None, disc_reward=None, min_cum_rewards=-10.0, max_cum_rewards=100.0, lr=1e-3, weight_decay=1e-4,
                 valuenet=None, warmup_epochs=20, max_epochs=20000, fused=None, overlap_fit=True, overlap_reset=None):
        self.vec_env = vec_env
        env = vec_env.env if hasattr(vec_env, "env") else vec_env
        self.env = env
        self.task = env.task
        self.device = torch.device(self.task.device)
        self.num_actors = self.task.num_envs
        self.horizon_length, self.gamma = horizon_length, gamma
        self.inversion_penalty_scale = inversion_penalty_scale         # amp_humanoid_smpl_sept_task.yaml:128
        self.step_to_pred = self.task.step_to_pred
        self.policy = policy or (lambda obs: torch.randn(self.num_actors, self.task.num_actions, device=self.device) * math.exp(-2.9))
        self._no_disc = disc_reward is None
        self.disc_reward = disc_reward or (lambda amp_obs: torch.zeros(self.num_actors, device=self.device))
        self.min_cum_rewards, self.max_cum_rewards = min_cum_rewards, max_cum_rewards     # common_agent.py:154-155
        self.valuenet = (valuenet if valuenet is not None else ValuePoseNet(use_pose=use_pose, use_vel=use_vel)).to(self.device)
        broadcast_parameters(self.valuenet)                                                # hvd.setup_algo, common_agent.py:165-166
        self.bucket = FlatGradBucket(self.valuenet.parameters(), extra=2)                  # tail: [loss sum, episode count]
        # the fused step: three small HIP launches around the LocoVal kernels (include/emloco_predictor.h)
        self.overlap_fit = bool(overlap_fit)
        if self.overlap_fit and hasattr(self.task, "overlap_obs") and getattr(self.task, "_fused_reset", False):
            self.task.overlap_obs = True           # this loop calls task.wait_obs() before the policy reads the observations
            # the chain between two rigid-body steps in three launches on ONE stream (flags | compaction + dispatch order | reset
            # chain + every observation): this loop calls reset_done() before the policy reads the observations, which is all that
            # mode asks for.  With a discriminator the AMP observations of a step are scored right after it, before the resets:
            # the AMP rows of every env then stay in the flags launch (task.fused_amp_early), only the observation rows are deferred.
            if hasattr(self.task, "fused_chain") and os.environ.get("EMLOCO_FUSED_CHAIN", "1") != "0":
                self.task.fused_chain = True
                self.task.fused_amp_early = not self._no_disc      # the discriminator scores a step's AMP observations before the resets
            if overlap_reset is None:
                overlap_reset = os.environ.get("EMLOCO_OVERLAP_RESET", "0") == "1"
            if hasattr(self.task, "overlap_reset") and overlap_reset:
                # opt-in: the reset chain and the reset envs' step on a second stream beside the step of the live envs.  This
                # loop calls task.wait_reset() before a policy that reads the reset envs' fresh observations (a policy object
                # may declare `reads_obs = False`).  Measured on MI355X: +2.6 % on the 4096-env rollout
                # without a policy in the loop; the resident rigid-body launch leaves no wave slot free for most of its run,
                # so the gain hangs on launch timing and the mode stays off unless asked for.
                self.task.overlap_reset = True
                self.task.fused_chain = False          # the two-chain schedule has its own arrangement of these launches
                # with the reset chain off the caller's stream the observation launch has nothing to hide behind: on a side
                # stream it costs two cross-stream hand-overs (~13 us each, measured) around a 50 us launch the loop waits for
                self.task.overlap_obs = os.environ.get("EMLOCO_OVERLAP_OBS", "0") == "1"
        # Nobody in this loop reads the AMP observations when there is no discriminator: the task keeps their history as a ring
        # (HumanoidAMP.enable_amp_ring: no 23 KB / env shift per step; `task.amp_obs_logical()` still gives the reference's tensor;
        # detach() hands the task back in the reference's layout).  EMLOCO_AMP_RING=0: the shifted layout.
        self._amp_ring = (self._no_disc and getattr(self.task, "fused_chain", False) and hasattr(self.task, "enable_amp_ring")
                          and self.device.type == "cuda" and os.environ.get("EMLOCO_AMP_RING", "1") != "0")
        if self._amp_ring:
            self.task.enable_amp_ring(True)
        elif getattr(self.task, "amp_ring", False):
            self.task.enable_amp_ring(False)               # a task an earlier loop left in the ring layout: this loop reads the plain tensor
        self.fused = (isinstance(self.valuenet, ValuePoseNet) and self.device.type == "cuda") if fused is None else bool(fused)
        if not self.fused and type(self)._bookkeeping is LocoValRollout._bookkeeping:
            raise RuntimeError("LocoValRollout runs its bookkeeping and fit as HIP kernels: it needs libemloco_hip.so, a gfx950 device "
                               "and the HIP ValuePoseNet (there is no CPU path in the product)")
        if self.fused:
            self._flat_params = torch.cat([p.detach().reshape(-1) for p in self.valuenet.parameters()]).contiguous()
            o = 0
            for p in self.valuenet.parameters():               # the parameters become views of one flat buffer (AdamW in one launch)
                p.data = self._flat_params[o:o + p.numel()].view_as(p)
                o += p.numel()
        self.vnet_optimizer = GatedFlatAdamW(self.valuenet.parameters(), self.bucket.grads, lr=lr, weight_decay=weight_decay)
        self.vnet_scheduler = CosineAnnealingLR(self.vnet_optimizer, warmup_epochs=warmup_epochs, max_epochs=max_epochs)
        E = self.num_actors
        self.acc = self._make_return_state(E, self.step_to_pred, gamma, self.device)
        self.game_combined_rewards = torch.zeros(E, device=self.device)
        self.started = False
        self.frames = 0
        # [last fit's loss sum, last fit's episodes, total loss sum, total episodes, number of fits] -- on the device
        self._stats = torch.zeros(5, device=self.device, dtype=torch.float64)
        if hasattr(self.task, "attach_returns"):
            self.task.attach_returns(None)
        self._disc_halves = None
        if self.fused:
            self._init_fused()

    def _init_fused(self):
        import ctypes as C
        from ..predictor import ops
        E, dev, task, a = self.num_actors, self.device, self.task, self.acc
        f = lambda *s: torch.zeros(*s, device=dev)
        self._fz = dict(traj13=f(E, 13, 3), pose=f(E, 24, 3), vel=f(E, 2), target=f(E), weight=f(E), value=f(E), dvalue=f(E),
                        x100=f(E, 100), h1=f(E, 49), h2=f(E, 24), ang=f(E), dtraj=f(E, 13, 3), ws=f(E * 6174), zeros=f(E),
                        steps=[f(1), f(1)], m=f(6174), v=f(6174), slot=torch.zeros(E, dtype=torch.int32, device=dev))
        z = self._fz
        self._check_fused_inputs()
        p = lambda t: t.data_ptr()
        self._fstep = ops.LocoValStep(E, int(self.step_to_pred), float(self.gamma), float(self.inversion_penalty_scale),
                                      float(self.min_cum_rewards), float(self.max_cum_rewards), p(a.current_rewards), p(a.current_lengths),
                                      p(a.current_combined_rewards), p(a.discount_coefs), p(task.waypoint_traj), p(task.init_pose),
                                      p(task.init_vel), p(z["traj13"]), p(z["pose"]), p(z["vel"]), p(z["target"]), p(z["weight"]))
        self._flip = 0
        # The fit (forward, loss gradient, backward, all-reduce, AdamW: ~75 us of small launches) only reads what the returns
        # kernel staged (traj13 / pose / vel / target / weight), so it runs on a side stream while the main stream goes on to
        # reset the finished envs and to launch the next physics step; the next returns kernel waits for it.
        # the return bookkeeping rides in the task's flags launch when there is no AMP reward to wait for (one launch less on the chain)
        self._returns_in_flags = (self._no_disc and hasattr(task, "attach_returns") and torch.device(dev).type == "cuda"
                                  and os.environ.get("EMLOCO_RETURNS_IN_FLAGS", "1") != "0")
        if hasattr(task, "attach_returns"):
            task.attach_returns(self._fstep if self._returns_in_flags else None, before=self._before_flags)      # (None: a hook an earlier loop left behind goes)
        self._side = torch.cuda.Stream(device=dev, priority=int(os.environ.get("EMLOCO_FIT_PRIORITY", "0"))) if (self.overlap_fit and torch.device(dev).type == "cuda") else None
        self._ev_staged = torch.cuda.Event() if self._side is not None else None
        # The staging buffers the returns launch writes and the fit reads are a RING of `EMLOCO_FIT_BUFFERS` sets (default 4): step t
        # writes set t mod 4, which the fit of step t - 4 has long read.  That a set is free is then checked by the HOST (a wait on the
        # event of a fit issued four steps ago: it has completed, the call returns at once; it only ever blocks a host that has run four
        # steps ahead of the GPU) instead of by a wait packet on the main stream behind the rigid-body launch: the gap between that
        # launch and the flags launch shrinks from 11 to 6 us (profiles/r04_env_step_trace.txt), 0.4580 -> 0.4545 ms per step, same bytes.
        # 1 = one set and the stream-side wait.  (The other packet of the hand-shake -- the event the fit's stream waits for -- stays:
        # replacing it by a counter the flags launch publishes + hipStreamWaitValue64 on the fit's stream works and is 80 us SLOWER.)
        self._nbuf = max(1, int(os.environ.get("EMLOCO_FIT_BUFFERS", "4"))) if self._side is not None else 1
        # The fit in GROUPS of `EMLOCO_FIT_EVERY` steps (round 5; default 4, 1 = every step as before).  What is left of the hand-shake
        # after the ring above is the event the fit's stream waits for: a record packet on the main stream behind every flags launch and
        # five small launches on the side stream beside every chain (~15 us of a 0.458 ms step, profiles/r04_env_step_trace.txt).  The
        # return bookkeeping already stages a step's rows in a set of its own, so the fits of k steps can be issued together, in step
        # order, behind ONE event after the k-th step: the same launches on the same staged values with the same learning rate (a group
        # never straddles an epoch: end_epoch flushes), so the network after every group -- and after every epoch -- is the per-step
        # loop's bit for bit (tests/test_gpu_env.py).  The ring then holds two groups (one being written, one being fitted).  Not with a
        # deferred discriminator (its GEMMs want to start under the NEXT rigid-body launch) nor with EMLOCO_FIT_UNDER_PHYSICS.
        self._fit_every = max(1, int(os.environ.get("EMLOCO_FIT_EVERY", "4"))) if self._side is not None else 1
        if os.environ.get("EMLOCO_FIT_UNDER_PHYSICS", "0") == "1" or not self._no_disc:
            self._fit_every = 1
        if self._fit_every > 1:
            self._nbuf = max(self._nbuf, 2 * self._fit_every)
        self._pending = []                                 # staging sets whose fits have not been issued yet, in step order
        # Host-side readers of the LocoVal network that do not go through this loop (a checkpoint save mid-epoch, `state_dict()` of the
        # value net) would see a network up to `_fit_every` - 1 fits behind: its state_dict waits for the pending fits first.
        if self._fit_every > 1 and hasattr(self.valuenet, "register_state_dict_pre_hook"):
            import weakref
            me = weakref.ref(self)

            def _flush_before_state_dict(*_a, **_k):
                loop = me()
                if loop is not None and getattr(loop, "_pending", None):
                    loop._sync_fit()
            self.valuenet.register_state_dict_pre_hook(_flush_before_state_dict)
        stage_keys = ("traj13", "pose", "vel", "target", "weight")
        self._stage = [{k: (z[k] if i == 0 else torch.zeros_like(z[k])) for k in stage_keys} for i in range(self._nbuf)]
        self._ev_fits = [torch.cuda.Event() for _ in range(self._nbuf)] if self._side is not None else []
        # WHEN the fit's five small launches are issued: at once (beside the resets' launches, which are on the chain between two
        # rigid-body launches) or right ahead of the next rigid-body launch (beside a kernel nothing waits for).  Measured on 16 hardware
        # queues: at once 0.4478 ms per step, ahead of the launch 0.4510 -- the fit's workgroups then sit in the rigid-body launch's wave
        # slots for its first ~300 us (its kernel time goes 0.364 -> 0.373 ms).  Off unless EMLOCO_FIT_UNDER_PHYSICS=1.
        self._fit_under_physics = (self._side is not None and os.environ.get("EMLOCO_FIT_UNDER_PHYSICS", "0") == "1")
        self._fit_waiting = False
        self._buf_busy = [False] * self._nbuf             # a fit that reads the set has been issued (its event recorded)
        self._buf_event = list(range(self._nbuf))          # which set's event covers that fit (the last set of its group)
        self._buf = 0                                      # the set the next returns launch writes
        # The discriminator off the chain between two rigid-body steps.  Its style reward (amp_continuous_value.py:90-96) feeds the
        # return bookkeeping only -- not the action, not the resets -- but in the reference's order it sits between env.step and the
        # next env_reset: 3 GEMMs (30 GFLOP at 4096 envs, ~0.26 ms) every step.  Deferred mode: the task's flags launch STAGES the
        # step (LocoVal inputs, penalised reward, done flag: emloco_task_post_physics_returns with staging arrays), one launch on the
        # main stream takes the normalised GEMM operand out of the AMP observations (`disc_stage`), and the side stream runs the
        # GEMMs, the scalar transform, emloco_locoval_returns_finish and the fit while the main stream goes on to the resets, the
        # policy and the next rigid-body launch.  Same kernels on the same values: returns, targets and the fitted network are the
        # sequential loop's bit for bit (tests/test_gpu_env.py).  Needs a discriminator that comes in two halves (`disc_stage` /
        # `disc_reward_staged` on the object `disc_reward` is bound to: AMPPolicyBundle) and the task's returns hook.
        owner = getattr(self.disc_reward, "__self__", None)
        self._disc_halves = None
        self._disc_after_launch = False
        if (not self._no_disc and self._side is not None and hasattr(task, "attach_returns") and hasattr(owner, "disc_stage")
                and hasattr(owner, "disc_reward_staged") and getattr(task, "fused_chain", False)
                and os.environ.get("EMLOCO_DEFER_DISC", "1") != "0"):
            self._disc_halves = (owner.disc_stage, owner.disc_reward_staged)
            # the staged GEMM operand travels with the ring of staging sets: stage(t) on the main stream must not overwrite what the
            # discriminator of an earlier step still reads on the side stream (the host-side hand-back of a set, `_acquire_stage`, runs
            # ahead of every flags launch and so ahead of every stage; it waits for the fit BEHIND that discriminator)
            if hasattr(owner, "disc_stage_ring"):
                owner.disc_stage_ring(self._nbuf)
            elif self._nbuf > 1:
                raise RuntimeError("LocoValRollout: a two-half discriminator must provide disc_stage_ring(n) to run beside a ring of "
                                   "staging sets (set EMLOCO_FIT_BUFFERS=1 for a single set and the stream-side wait)")
            # (measured on MI355X, 4096 envs: issued at once 0.93 ms / step, issued ahead of the rigid-body launch 1.06 ms, the
            # sequential order 1.01 ms -- the rigid-body launch holds 3 waves x 168 registers per SIMD and 12 x 12.4 KB of LDS per CU:
            # a GEMM workgroup beside it takes residency away from it, the pipes do not overlap for free; off by default)
            # ... on the runtime's default of 4 hardware queues.  With 16 (what the package sets at import, emloco_amd/__init__.py) the
            # discriminator's stream has a queue of its own and issuing it ahead of the rigid-body launch WINS: 0.961 -> 0.915 ms per step,
            # 4.26 -> 4.48 M env-steps/s on one box (profiles/r04_ab_disc_schedule.txt); default: on from 16 queues up.
            dup = os.environ.get("EMLOCO_DISC_UNDER_PHYSICS", "auto")
            from .. import hw_queues                       # what the runtime was initialised with (None: unknown -> the 4-queue order)
            self._disc_under_physics = dup == "1" or (dup == "auto" and (hw_queues() or 4) >= 16)
            self._disc_pending = False
            # Round 6 experiment: WHERE in the host's sequence the side stream's ~15 launches are issued.  Ahead of the rigid-body launch
            # (round 4; default) there is a 64 us hole on the main stream between the policy's last launch and the PD-target launch
            # (profiles/r06_env_step_trace_disc.txt); issued right BEHIND the rigid-body launch (EMLOCO_DISC_AFTER_LAUNCH=1: the task's
            # `after_physics_launch` hook; the event they wait for is still recorded ahead of it) the step is no faster: 0.873-0.878
            # against 0.865-0.876 ms, three interleaved runs each (tools/exp/c2_after_launch.py) -- the hole is not the host's issue time.
            self._disc_after_launch = (self._disc_under_physics and os.environ.get("EMLOCO_DISC_AFTER_LAUNCH", "0") == "1")
            if self._disc_after_launch:
                task.after_physics_launch = self._issue_deferred_disc
            for i, st_ in enumerate(self._stage):          # the staged reward / done flag travel with the set
                st_["staged_reward"] = f(E)
                st_["staged_done"] = torch.zeros(E, dtype=torch.uint8, device=dev)
            z["staged_reward"], z["staged_done"] = self._stage[0]["staged_reward"], self._stage[0]["staged_done"]
            self._fstep.staged_reward, self._fstep.staged_done = p(z["staged_reward"]), p(z["staged_done"])
            task.attach_returns(self._fstep, before=self._before_flags)
            self._returns_in_flags = True

    def _check_fused_inputs(self):
        """The fused step keeps raw device pointers of the task's LocoVal inputs (the kernel hard-codes their strides: 15 x 3
        waypoints, 24 x 3 joints, 2 velocity components per env): shape, dtype, device and -- from the second call on -- the
        address must be what they were when the pointers were taken (a task that re-allocates them must rebuild the rollout)."""
        task, E = self.task, self.num_actors
        want = {"waypoint_traj": (E, 15, 3), "init_pose": (E, 24, 3), "init_vel": (E, 2)}
        ptrs = []
        for name, shape in want.items():
            t = getattr(task, name)
            if tuple(t.shape) != shape or t.dtype != torch.float32 or t.device.type != self.device.type or not t.is_contiguous():
                raise RuntimeError(f"LocoValRollout: task.{name} must be a contiguous float32 {shape} tensor on {self.device}, got "
                                   f"{tuple(t.shape)} {t.dtype} {t.device}")
            ptrs.append(t.data_ptr())
        if getattr(self, "_fused_ptrs", None) is None:
            self._fused_ptrs = ptrs
        elif ptrs != self._fused_ptrs:
            raise RuntimeError("LocoValRollout: the task re-allocated waypoint_traj / init_pose / init_vel after the fused step took their addresses")

    def _acquire_stage(self):
        """Make the staging set of the coming returns launch current (struct pointers + the arrays the fit launches read) once the
        fit that last read it is done: the host waits on that fit's event (ring of sets), or the main stream does (one set)."""
        if self._side is None:
            return
        k = self._buf
        if k in self._pending:                              # (a ring shorter than the fits in flight: cannot happen with 2 groups of sets)
            self._flush_fits()
        if self._buf_busy[k]:
            ev = self._ev_fits[self._buf_event[k]]          # the event behind the LAST fit of the group this set belonged to
            if self._nbuf > 1:
                ev.synchronize()
            else:
                ev.wait(torch.cuda.current_stream(self.device))
        if self._nbuf > 1:
            st_, s, z = self._stage[k], self._fstep, self._fz
            for name, t in st_.items():
                z[name] = t
                setattr(s, name, t.data_ptr())

    def _fit_issued(self):
        """The fit that reads the current staging set has been issued on the side stream: record its event, move to the next set."""
        k = self._buf
        self._ev_fits[k].record(self._side)
        self._buf_busy[k] = True
        self._buf_event[k] = k
        self._buf = (k + 1) % self._nbuf

    def _sync_fit(self):
        """Host-side readers of what the fit writes (statistics, LocoVal weights) wait for the side stream."""
        if getattr(self, "_disc_halves", None) is not None:
            self._issue_deferred_disc()                     # a step whose discriminator half was still waiting for the next step
        if getattr(self, "_fit_waiting", False):
            self._issue_fit()
        if getattr(self, "_pending", None):
            self._flush_fits()
        if getattr(self, "_side", None) is not None:
            self._side.synchronize()

    def _fused_step(self, rewards, amp_rewards, dones, inverted):
        """Bookkeeping + fit of one rollout step in 6 launches (returns, LocoVal fwd, fit grad, LocoVal bwd x2, gated AdamW)."""
        import ctypes as C
        from ..predictor import ops
        from ..sim import current_stream_handle
        lib = ops._lib()
        z, E = self._fz, self.num_actors
        st = current_stream_handle(self.device)
        P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        assert rewards.dtype == torch.float32 and dones.dtype == torch.int64 and inverted.dtype == torch.bool
        self._check_fused_inputs()
        s = self._fstep
        s.inversion_penalty = float(self.inversion_penalty_scale)
        main = torch.cuda.current_stream(self.device) if self._side is not None else None
        if getattr(self.task, "_returns_in_flags", False) and amp_rewards is None:
            self.task._returns_in_flags = False             # this step's flags launch has advanced the returns (_before_step waited for the fit)
        else:
            self._acquire_stage()                           # the fit that last read this staging set is done with it
            ops._chk(lib.emloco_locoval_returns(C.byref(s), P(rewards.contiguous()), P(amp_rewards), P(dones.contiguous()), P(inverted.contiguous()), st),
                     "emloco_locoval_returns")
            if s.staged_reward:                             # a step that carries staging arrays was only staged by that call
                ops._chk(lib.emloco_locoval_returns_finish(C.byref(s), P(amp_rewards), st), "emloco_locoval_returns_finish")
        if self._side is None:
            self._fit_launches(st)
            return
        if self._fit_under_physics:                         # issued from _before_step, right ahead of the next rigid-body launch
            self._fit_waiting = True
            return
        self._issue_fit()

    def _issue_fit(self):
        """The step whose returns are staged in the current set joins the pending group; the group's fits are issued once it is full."""
        self._fit_waiting = False
        self._pending.append(self._buf)
        self._buf = (self._buf + 1) % self._nbuf
        if len(self._pending) >= self._fit_every:
            self._flush_fits()

    def _flush_fits(self):
        """The fits of the pending steps, in step order, on the side stream behind ONE event of the main stream; one event behind the
        last of them hands all their staging sets back."""
        from ..sim import current_stream_handle
        if not self._pending:
            return
        self._ev_staged.record(torch.cuda.current_stream(self.device))
        self._side.wait_event(self._ev_staged)
        last = self._pending[-1]
        with torch.cuda.stream(self._side):
            st = current_stream_handle(self.device)
            for k in self._pending:
                self._fit_launches(st, self._stage[k] if self._nbuf > 1 else None)
            self._ev_fits[last].record(self._side)
        for k in self._pending:
            self._buf_busy[k] = True
            self._buf_event[k] = last
        self._pending = []

    def _fit_launches(self, st, stage=None):
        """The five launches of one step's fit on stream `st`; `stage`: the staging set that holds the step's rows (default: the
        current one, whose arrays `self._fz` points at)."""
        import ctypes as C
        from ..predictor import ops
        lib = ops._lib()
        z, E = self._fz, self.num_actors
        if stage is not None:
            z = dict(z)
            z.update(stage)
        P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        n = self.valuenet._network
        w = [n.fc1.weight, n.fc1.bias, n.fc2.weight, n.fc2.bias, n.fc3.weight, n.fc3.bias]
        ops._chk(lib.emloco_locoval_fwd_rows(E, P(z["traj13"]), 3, P(z["pose"]), P(z["vel"]), *[P(t) for t in w], P(z["value"]), P(z["x100"]),
                                             P(z["h1"]), P(z["h2"]), P(z["ang"]), P(z["weight"]), st), "emloco_locoval_fwd_rows")
        ops._chk(lib.emloco_locoval_fit_grad(E, P(z["value"]), P(z["target"]), P(z["weight"]), P(z["dvalue"]), P(self.bucket.tail), P(z["slot"]), st),
                 "emloco_locoval_fit_grad")
        ops._chk(lib.emloco_locoval_bwd_rows(E, P(z["traj13"]), 3, P(z["pose"]), P(z["vel"]), P(w[0]), P(w[2]), P(w[4]), P(z["value"]), P(z["x100"]),
                                             P(z["h1"]), P(z["h2"]), P(z["ang"]), P(z["dvalue"]), P(z["slot"]), P(self.bucket.tail[1:]),
                                             P(self.bucket.grads), P(z["dtraj"]), P(z["ws"]), st), "emloco_locoval_bwd_rows")
        self.bucket.all_reduce(average=False)                                               # unconditional: one collective per step
        g = self.vnet_optimizer.param_groups[0]
        a, b = z["steps"][self._flip], z["steps"][self._flip ^ 1]
        self._flip ^= 1
        ops._chk(lib.emloco_adamw_gated(6174, P(self._flat_params), P(self.bucket.grads), P(z["m"]), P(z["v"]), P(a), P(b), P(self.bucket.tail),
                                        float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                                        P(self._stats), st), "emloco_adamw_gated")

    # counters of the fit, read from the device on demand (a host synchronisation each)
    @property
    def vnet_loss(self):
        """sum-MSE of the most recent fit divided by its (global) episode count"""
        self._sync_fit()
        a = self._stats.tolist()
        return a[0] / a[1] if a[1] > 0 else 0.0

    @property
    def fitted_episodes(self):
        self._sync_fit()
        return int(round(self._stats[3].item()))

    @property
    def vnet_fits(self):
        self._sync_fit()
        return int(round(self._stats[4].item()))

    def _reset_finished(self):
        """env_reset(done_indices) of :46 -- every env on the first call, afterwards the envs whose reset flag is set."""
        if not self.started:
            self.env.reset(torch.arange(self.num_actors, device=self.device))
            self.started = True
        elif hasattr(self.env, "reset_done"):
            self.env.reset_done()
        else:
            self.env.reset(self.task.reset_buf.nonzero(as_tuple=False).flatten())

    def step_once(self):
        """One iteration of the play_steps loop (:45-145): reset finished envs, act, step, bookkeeping, LocoVal fit."""
        task = self.task
        with torch.no_grad():
            self._reset_finished()
            if hasattr(task, "wait_obs"):
                task.wait_obs()                                   # the observations of the last step (built on a side stream)
            if hasattr(task, "wait_reset") and getattr(self.policy, "reads_obs", True):
                task.wait_reset()                                 # the observations of the envs that were just reset
            actions = self.policy(task.obs_buf)
            self._before_step()
            obs, rewards, dones, infos = self.vec_env.step(actions)
            inverted = task.inverted
            self.frames += self.num_actors
            if self._disc_halves is not None and getattr(task, "_returns_in_flags", False):
                self._deferred_disc_step(infos["amp_obs"])
                return
            if not self._no_disc and hasattr(task, "wait_obs") and not (getattr(task, "fused_chain", False) and getattr(task, "fused_amp_early", False)):
                task.wait_obs()                                   # the discriminator reads this step's AMP observations
            amp_rewards = None if self._no_disc else self.disc_reward(infos["amp_obs"]).contiguous()
        self._bookkeeping(rewards, amp_rewards, dones, inverted)

    def _deferred_disc_step(self, amp_obs):
        """The step's flags launch has staged the return bookkeeping (after waiting for the previous fit: `_before_flags`).  Main stream:
        one launch that reads the AMP observations; side stream: discriminator GEMMs, reward transform, the bookkeeping's second half,
        the fit.  Nothing the side stream reads is written by the main stream before the next flags launch, whose staging set is free (`_acquire_stage`).
        WHEN the side stream's work is issued decides what it runs beside: issued here it competes with the next step's policy GEMMs for
        the matrix pipes (the resets' small launches aside); issued right ahead of the next rigid-body launch (`_disc_under_physics`,
        from `_before_step`) it runs beside a kernel that is bound by the vector pipe."""
        stage, _finish = self._disc_halves
        self.task._returns_in_flags = False
        stage(amp_obs)
        self._disc_pending = True
        if not self._disc_under_physics:
            self._issue_deferred_disc()

    def _issue_deferred_disc(self):
        import ctypes as C
        from ..predictor import ops
        from ..sim import current_stream_handle
        if not getattr(self, "_disc_pending", False):
            return
        self._disc_pending = False
        _stage, finish = self._disc_halves
        if not getattr(self, "_disc_event_recorded", False):   # (issued behind the rigid-body launch: `_before_step` recorded it AHEAD of the launch)
            self._ev_staged.record(torch.cuda.current_stream(self.device))
        self._disc_event_recorded = False
        self._side.wait_event(self._ev_staged)
        with torch.cuda.stream(self._side):
            amp_rewards = finish().contiguous()
            st = current_stream_handle(self.device)
            ops._chk(ops._lib().emloco_locoval_returns_finish(C.byref(self._fstep), C.c_void_p(amp_rewards.data_ptr()), st),
                     "emloco_locoval_returns_finish")
            self._fit_launches(st)
            self._fit_issued()

    def detach(self):
        """Take this loop's return bookkeeping out of the task's flags launch (a caller that steps the env on its own in between)."""
        if getattr(self, "_disc_halves", None) is not None:
            self._issue_deferred_disc()
        if getattr(self, "_fit_waiting", False):
            self._issue_fit()
        if getattr(self, "_pending", None):
            self._flush_fits()
        if getattr(self, "_returns_in_flags", False) and hasattr(self.task, "attach_returns"):
            self.task.attach_returns(None)
        if getattr(self, "_disc_after_launch", False):
            self.task.after_physics_launch = None
        if getattr(self, "_amp_ring", False):
            self.task.enable_amp_ring(False)

    def attach(self):
        if getattr(self, "_returns_in_flags", False) and hasattr(self.task, "attach_returns"):
            self.task.attach_returns(self._fstep, before=self._before_flags)
        if getattr(self, "_disc_after_launch", False):
            self.task.after_physics_launch = self._issue_deferred_disc
        if getattr(self, "_amp_ring", False):
            self.task.enable_amp_ring(True)

    def _before_flags(self):
        """Runs right ahead of the task's flags launch when that launch carries the return bookkeeping: the fit that last read the
        staging set the launch is about to overwrite must be done (`_acquire_stage`)."""
        self._acquire_stage()

    def _before_step(self):
        """With the return bookkeeping inside the task's flags launch: what _fused_step does ahead of its own returns launch -- the
        staging buffers must be free (the previous fit has read them), the penalty scale current."""
        if getattr(self, "_fit_waiting", False):
            self._issue_fit()                               # the previous step's fit, beside this step's rigid-body launch
        if not getattr(self, "_returns_in_flags", False) or not self.fused:
            return
        if self._disc_halves is not None:
            if not self._disc_after_launch:
                self._issue_deferred_disc()                 # the previous step's discriminator half, beside this step's rigid-body launch
            elif getattr(self, "_disc_pending", False):
                # its launches follow BEHIND the rigid-body launch (`after_physics_launch`); what they wait for -- the staged operand -- is
                # marked here, ahead of the launch: an event behind it would make the side stream wait for the rigid-body kernel
                self._ev_staged.record(torch.cuda.current_stream(self.device))
                self._disc_event_recorded = True
        self._check_fused_inputs()
        self._fstep.inversion_penalty = float(self.inversion_penalty_scale)

    def _make_return_state(self, E, step_to_pred, gamma, device):
        return _ReturnState(E, step_to_pred, gamma, device)

    def _bookkeeping(self, rewards, amp_rewards, dones, inverted):
        """The step's return accumulation and LocoVal fit: 6 HIP launches (the torch formulation of the same arithmetic lives
        with the tests, in their oracle package: TorchLocoValRollout overrides this hook)."""
        self._fused_step(rewards, amp_rewards, dones, inverted)

    def end_epoch(self):
        """common_agent.py:205-209: the cosine schedule advances once per epoch, once episodes have finished."""
        if getattr(self, "_disc_halves", None) is not None:
            self._issue_deferred_disc()                          # the epoch's last fit is issued with the epoch's learning rate
        if getattr(self, "_fit_waiting", False):
            self._issue_fit()
        if getattr(self, "_pending", None):
            self._flush_fits()                                   # a group never straddles an epoch: its fits run at this epoch's rate
        if not getattr(self, "_sched_live", False):
            self._sched_live = self.fitted_episodes > 0          # one read per epoch until the first episode has finished
        if self._sched_live:
            # (the fit's AdamW is a HIP kernel, `emloco_adamw_gated`: torch never sees an optimizer.step() and would warn on every
            # epoch that the schedule is stepped first -- the counter it looks at is bumped here instead; the schedule itself is pinned
            # to the reference's class, tests/golden/locoval_lr_schedule.npz)
            self.vnet_optimizer._opt_called = True
            self.vnet_scheduler.step()

    def play_steps(self):
        for n in range(self.horizon_length):
            self.step_once()
        self.end_epoch()
        return self.vnet_loss

    # ------------------------------------------------------------------ checkpoints (common_agent.py:248-264, finetune branch)
    def save(self, model_output_file, epoch_num=None):
        """`<file>_valuenet.pth`, or `<file>_valuenet_<epoch:08d>.pth` for the intermediate checkpoints: a plain state_dict
        with the reference's keys (`_network.fc{1,2,3}.{weight,bias}`), loadable by train_jta.py / evaluate_jta.py."""
        path = model_output_file + ("_valuenet.pth" if epoch_num is None else "_valuenet_" + str(epoch_num).zfill(8) + ".pth")
        self._sync_fit()
        torch.save({k: v.detach().cpu() for k, v in self.valuenet.state_dict().items()}, path)
        return path

    def restore(self, path):
        self._sync_fit()
        self.valuenet.load_state_dict(torch.load(path, map_location=self.device))

    def train(self, max_epochs, model_output_file=None, save_freq=200, save_intermediate=True):
        """The finetune loop of CommonAgent.train: one epoch = one play_steps horizon."""
        for epoch_num in range(1, max_epochs + 1):
            self.play_steps()
            if model_output_file and save_freq > 0 and epoch_num % save_freq == 0:
                self.save(model_output_file)
                if save_intermediate and epoch_num % (save_freq * 5) == 0:
                    self.save(model_output_file, epoch_num)
        if model_output_file:
            self.save(model_output_file)
        return self.vnet_loss

