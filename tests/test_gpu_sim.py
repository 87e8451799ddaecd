"""GPU parity: the HIP rollout step (through the C ABI) against the CPU oracle on the same inputs.

The north star asks fp32 body states within 1e-4 rel.  The step does better: it is BIT-EXACT against the
oracle, because both sides run the same fp32 operation sequence (library built with -ffp-contract=off,
correctly rounded divide/sqrt, wave reductions in a fixed butterfly order that the oracle mirrors, and
sin/cos/atan built from + - * / sqrt only).  That matters: a falling humanoid is chaotic, so any
last-ulp difference grows to centimetres within ~20 control steps.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _close(a, b, rel=1e-4, abs_=2e-5, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b)
    tol = abs_ + rel * np.maximum(np.abs(a), np.abs(b))
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} beyond tol, max err {err.max():.3e} (scale {np.abs(b).max():.3e})"


def _mk(E, seed=0, n_sub=2, n_calls=2):
    from emloco_amd import _lib as L
    from emloco_amd.sim import NativeSim
    from helpers import oracle_sim, scene_state, varied_models
    models = varied_models(E, seed)
    root, dof, tgt = scene_state(E, seed + 1)
    # the HIP step fuses the n_calls x n_sub substeps of one env.step into one launch (joint quaternions stay
    # in registers in between), so the oracle runs them as one call of n_calls * n_sub substeps too
    osim = oracle_sim(models, root, dof, tgt, n_sub=n_sub * n_calls)
    gsim = NativeSim(models, L.default_sim_params(n_sub=n_sub))
    gsim.root_state.copy_(torch.from_numpy(root))
    gsim.dof_state.view(E, 69, 2).copy_(torch.from_numpy(dof))
    gsim.pd_target.copy_(torch.from_numpy(tgt))
    return osim, gsim


def _compare(osim, gsim, E, rel=1e-4, what=""):
    torch.cuda.synchronize()
    pairs = [("root_state", gsim.root_state.cpu().numpy(), osim.root_state),
             ("dof_state", gsim.dof_state.view(E, 69, 2).cpu().numpy(), osim.dof_state),
             ("rb_state", gsim.rigid_body_state.view(E, 24, 13).cpu().numpy(), osim.rb_state),
             ("contact_force", gsim.contact_force.view(E, 24, 3).cpu().numpy(), osim.contact_force),
             ("dof_force", gsim.dof_force.view(E, 69).cpu().numpy(), osim.dof_force)]
    for name, a, b in pairs:
        _close(a, b, rel, abs_=1e-4, what=f"{what} {name}")          # the stated bar (1e-4 rel)
        assert np.array_equal(a, b), f"{what} {name}: not bit-exact, max abs diff {np.abs(a - b).max():.3e}"


def test_one_env_step_matches_oracle():
    E = 8
    osim, gsim = _mk(E)
    osim.step(1)           # = controlFrequencyInv (2) calls of gym.simulate x 2 substeps, fused
    gsim.step(2)
    _compare(osim, gsim, E, what="1 step")
    assert np.abs(osim.contact_force).max() > 50.0   # the scene really is in contact


def test_pd_stand_episode_matches_oracle():
    """168 control steps (one episode): env 0 stands, the perturbed ones stumble and fall (contact-rich, chaotic)."""
    E = 4
    osim, gsim = _mk(E, seed=3)
    for k in range(168):
        osim.step(1)
        gsim.step(2)
        if k in (0, 9, 49, 167):
            _compare(osim, gsim, E, what=f"step {k}")
    w = np.array([m for m in osim.arr["mass"].sum(1)]) * 9.81
    fz = gsim.contact_force.view(E, 24, 3)[:, :, 2].sum(1).cpu().numpy()
    assert abs(fz[0] - w[0]) / w[0] < 0.02, (fz, w)


def test_step_is_deterministic():
    E = 16
    _, a = _mk(E, seed=5)
    _, b = _mk(E, seed=5)
    for _ in range(10):
        a.step(2)
        b.step(2)
    torch.cuda.synchronize()
    assert torch.equal(a.rigid_body_state, b.rigid_body_state)
    assert torch.equal(a.contact_force, b.contact_force)


def test_fk_after_indexed_set_matches_oracle():
    import oracle
    E = 6
    osim, gsim = _mk(E, seed=7)
    ids = torch.tensor([1, 4], dtype=torch.int32, device=gsim.device)
    gsim.set_root_state_indexed(gsim.root_state, ids)
    gsim.set_dof_state_indexed(gsim.dof_state, ids)
    osim.fk()
    torch.cuda.synchronize()
    rb = gsim.rigid_body_state.view(E, 24, 13).cpu().numpy()
    _close(rb[[1, 4]], osim.rb_state[[1, 4]], 1e-5, what="fk")
    assert np.all(rb[0] == 0) or np.allclose(rb[0, :, 6], 0)  # untouched envs keep their (zero-initialised) body state


def test_4096_envs_stand_and_carry_their_weight():
    from emloco_amd import _lib as L
    from emloco_amd.sim import NativeSim
    from helpers import varied_models
    E = 4096
    models = varied_models(64, seed=11)
    models = [models[i % 64] for i in range(E)]
    sim = NativeSim(models, L.default_sim_params())
    sim.root_state[:, 2] = 0.93
    for _ in range(60):
        sim.step(2)
    torch.cuda.synchronize()
    rb = sim.rigid_body_state.view(E, 24, 13)
    assert torch.isfinite(rb).all()
    w = torch.tensor([m.total_mass() * 9.81 for m in models], device=sim.device)
    fz = sim.contact_force.view(E, 24, 3)[:, :, 2].sum(1)
    assert ((fz - w).abs() / w).max().item() < 0.05
    assert (rb[:, 0, 2] > 0.7).all()   # nobody fell


def test_self_collision_step_is_bit_exact_and_changes_the_motion():
    """has_self_collision: limb-limb penalty contacts (kernel phase 1b) over a full 168-step episode of folding, falling
    humanoids with weak drives: the HIP step stays on the oracle's bytes, and differs from the run without it."""
    from emloco_amd import _lib as L
    from emloco_amd.model import pack_self_collision
    from emloco_amd.sim import NativeSim
    from helpers import oracle_sim, scene_state, varied_models
    E = 8
    models = varied_models(E, 21)
    for m in models:
        m.kp = m.kp * 0.05
    root, dof, tgt = scene_state(E, 22, perturbed_from=0)
    dof[:, :, 0] *= 3.0
    sc = pack_self_collision(models)
    osim = oracle_sim(models, root, dof, tgt, self_collision=sc, n_sub=4)
    gsim = NativeSim(models, L.default_sim_params(n_sub=2), self_collision=sc)
    gref = NativeSim(models, L.default_sim_params(n_sub=2))
    for g in (gsim, gref):
        g.root_state.copy_(torch.from_numpy(root))
        g.dof_state.view(E, 69, 2).copy_(torch.from_numpy(dof))
        g.pd_target.copy_(torch.from_numpy(tgt))
    for k in range(168):
        osim.step(1)
        gsim.step(2)
        gref.step(2)
        if k in (0, 10, 60, 167):
            _compare(osim, gsim, E, what=f"self-collision step {k}")
    torch.cuda.synchronize()
    assert not torch.equal(gsim.dof_state, gref.dof_state)
    assert torch.isfinite(gsim.rigid_body_state).all()


def test_heightfield_ground_is_bit_exact_and_tilts_the_contact_forces():
    """Height-field terrain (emloco_sim_set_ground_heightfield): humanoids falling and tumbling on a sloped, bumpy field with
    self-collision on stay on the oracle's bytes over an episode; a flat field reproduces the plane bit for bit."""
    from emloco_amd import _lib as L
    from emloco_amd.model import pack_self_collision
    from emloco_amd.sim import NativeSim
    from helpers import bumpy_heightfield, oracle_sim, scene_state, varied_models
    E = 8
    models = varied_models(E, 31)
    root, dof, tgt = scene_state(E, 32, perturbed_from=0)
    hf = bumpy_heightfield(seed=5, amp=0.12, slope=0.2)
    root[:, 2] += 0.2 * (root[:, 0] - 50.0) + 0.1
    sc = pack_self_collision(models)
    osim = oracle_sim(models, root, dof, tgt, self_collision=sc, heightfield=hf, n_sub=4)
    gsim = NativeSim(models, L.default_sim_params(n_sub=2), self_collision=sc, heightfield=hf)
    flat = dict(samples=np.zeros((1100, 1100), np.int16), horizontal_scale=0.1, vertical_scale=0.005)
    gflat = NativeSim(models, L.default_sim_params(n_sub=2), self_collision=sc, heightfield=flat)
    gplane = NativeSim(models, L.default_sim_params(n_sub=2), self_collision=sc)
    for g in (gsim, gflat, gplane):
        g.root_state.copy_(torch.from_numpy(root))
        g.dof_state.view(E, 69, 2).copy_(torch.from_numpy(dof))
        g.pd_target.copy_(torch.from_numpy(tgt))
    tilted = 0.0
    for k in range(168):
        osim.step(1)
        gsim.step(2)
        if k < 40:
            gflat.step(2)
            gplane.step(2)
        if k in (0, 5, 20, 60, 167):
            _compare(osim, gsim, E, what=f"height-field step {k}")
        tilted = max(tilted, float(gsim.contact_force.view(E, 24, 3)[..., :2].abs().max()))
    torch.cuda.synchronize()
    assert tilted > 20.0                                       # contact forces have horizontal parts on the slope
    assert torch.isfinite(gsim.rigid_body_state).all()
    assert torch.equal(gflat.rigid_body_state, gplane.rigid_body_state) and torch.equal(gflat.contact_force, gplane.contact_force)


def test_large_random_targets_stay_finite_and_match_the_oracle():
    """Exploration-sized and absurd target noise (new targets every control step): the saturating drives' second pass and the
    link-speed limiter run in many envs; the first 8 envs stay on the oracle's bytes, all 1024 stay finite and bounded."""
    from emloco_amd import _lib as L
    from emloco_amd.sim import NativeSim
    from helpers import oracle_sim, scene_state, varied_models
    E, EO = 1024, 8
    models = varied_models(E, 41)
    root, dof, tgt = scene_state(E, 42)
    osim = oracle_sim(models[:EO], root[:EO], dof[:EO], tgt[:EO], n_sub=4)
    gsim = NativeSim(models, L.default_sim_params(n_sub=2))
    gsim.root_state.copy_(torch.from_numpy(root))
    gsim.dof_state.view(E, 69, 2).copy_(torch.from_numpy(dof))
    rng = np.random.default_rng(43)
    sig = np.where(np.arange(E) % 2 == 0, 0.3, 1.2)[:, None].astype(np.float32)      # even envs: exploration, odd: absurd
    vmax = 0.0
    for k in range(90):
        t = (rng.normal(size=(E, 69)).astype(np.float32) * sig)
        gsim.pd_target.copy_(torch.from_numpy(t))
        osim.pd_target[:] = t[:EO]
        osim.step(1)
        gsim.step(2)
        if k in (0, 15, 89):
            torch.cuda.synchronize()
            for name, a, b in (("root_state", gsim.root_state[:EO].cpu().numpy(), osim.root_state),
                               ("dof_state", gsim.dof_state.view(E, 69, 2)[:EO].cpu().numpy(), osim.dof_state),
                               ("dof_force", gsim.dof_force.view(E, 69)[:EO].cpu().numpy(), osim.dof_force)):
                assert np.array_equal(a, b), f"step {k} {name}: max abs diff {np.abs(a - b).max():.3e}"
        vmax = max(vmax, float(gsim.root_state[:, 7:10].norm(dim=1).max()))
    torch.cuda.synchronize()
    assert torch.isfinite(gsim.rigid_body_state).all() and torch.isfinite(gsim.dof_state).all()
    assert vmax < 60.0 and float(gsim.root_state[0::2, 7:10].norm(dim=1).max()) < 15.0
    eff = torch.from_numpy(np.stack([m.effort for m in models]).astype(np.float32)).cuda()
    assert (gsim.dof_force.view(E, 69).abs() <= eff * (1 + 1e-6)).all()


def test_cost_ordered_split_and_subset_launches_are_the_same_step():
    """emloco_sim_set_split (the substeps of a step as two dependent workgroups per env in one launch), emloco_sim_set_cost_order (longest-first dispatch from the previous launch's per-env durations) and
    emloco_sim_step_subset (skip flags + compacted id list = two launches that together step every env once) only change
    which workgroup steps which env: every state tensor stays on the oracle's bytes over an episode with falls."""
    E = 300
    osim, gsim = _mk(E, seed=21)
    gsim.set_cost_order(True)
    gsim.set_split(2)                                          # each env's 4 substeps as two dependent workgroups of one launch
    dev = gsim.device
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    for k in range(12):
        osim.step(1)
        if k % 3 == 2:                                         # split launch: a random third of the envs through the id list
            skip = (torch.rand(E, device=dev, generator=g) < 0.33).to(torch.int64)
            ids = torch.full((E,), -1, dtype=torch.int32, device=dev)
            nz = skip.nonzero().flatten().to(torch.int32)
            ids[:nz.numel()] = nz
            gsim.step_subset(2, skip=skip)
            gsim.step_subset(2, ids=ids)
        else:
            gsim.step(2)
        _compare(osim, gsim, E, what=f"step {k}")
    assert np.abs(osim.contact_force).max() > 50.0


def test_split_launch_lost_handover_is_an_error_not_silent_corruption():
    """A later part of a split launch whose predecessor never publishes its flag gives up after a bounded wait, raises the
    device error word and abandons that env's step; emloco_sim_sync then returns EMLOCO_E_HIP (reported once) instead of the
    step going on from stale hand-over state.  The other envs are stepped on the oracle's bytes; launches with other part
    counts in between (unsplit, list launches) cannot make a stale flag match: the tags are unique per launch and part."""
    from emloco_amd import _lib as L
    E = 64
    osim, gsim = _mk(E, seed=33)
    gsim.set_split(4)
    for _ in range(2):
        osim.step(1)
        gsim.step(2)
    gsim.sync()
    _compare(osim, gsim, E, what="before the poisoned launch")
    before = gsim.rigid_body_state.view(E, 24, 13)[5].clone()
    gsim.lib.emloco_sim_debug_poison_part.argtypes = [L.C.c_void_p, L.C.c_int, L.C.c_int]
    L.check(gsim.lib.emloco_sim_debug_poison_part(gsim._h, 5, 1 << 14), "poison")
    osim.step(1)
    gsim.step(2)
    with pytest.raises(L.EmlocoError, match="hand-over"):
        gsim.sync()
    gsim.sync()                                                # reported once, then cleared
    rb = gsim.rigid_body_state.view(E, 24, 13).cpu().numpy()
    keep = np.arange(E) != 5
    assert np.array_equal(rb[keep], osim.rb_state[keep]), "the other envs must be on the oracle's bytes"
    assert torch.equal(gsim.rigid_body_state.view(E, 24, 13)[5], before), "the poisoned env's step is abandoned, not corrupted"
    L.check(gsim.lib.emloco_sim_debug_poison_part(gsim._h, -1, 0), "unpoison")
    # a launch with another part count, then split launches again: stale flags never match a tag
    gsim.set_split(2); gsim.step(2); gsim.set_split(1); gsim.step(2); gsim.set_split(4); gsim.step(2)
    gsim.sync()


def test_cost_order_above_the_lds_sort_capacity():
    """emloco_sim_set_cost_order beyond 16384 envs (the sort keeps its buckets in a global workspace there): still every env
    stepped exactly once -- the cost-ordered launch equals the plain one."""
    from emloco_amd import _lib as L
    from emloco_amd.sim import NativeSim
    from helpers import scene_state, varied_models
    E = 16384 + 512
    base = varied_models(64, 3)
    models = [base[i % 64] for i in range(E)]
    root, dof, tgt = scene_state(E, 4)
    sims = []
    for order in (False, True):
        s = NativeSim(models, L.default_sim_params(n_sub=2))
        s.root_state.copy_(torch.from_numpy(root)); s.dof_state.view(E, 69, 2).copy_(torch.from_numpy(dof)); s.pd_target.copy_(torch.from_numpy(tgt))
        s.set_cost_order(order)
        for _ in range(3):
            s.step(2)
        s.sync()
        sims.append(s)
    assert torch.equal(sims[0].rigid_body_state, sims[1].rigid_body_state)
    assert torch.equal(sims[0].dof_state, sims[1].dof_state)


def test_effort_drives_through_the_c_abi_are_bit_exact_and_switch_back():
    """emloco_sim_set_dof_actuation_force (gym.set_dof_actuation_force_tensor, humanoid.py:1203-1207, `pdControl: False`): joint
    torques instead of position targets -- bit-exact against the oracle's drive_mode = 1, the reported dof forces are the
    commands clipped to the effort limits; emloco_sim_set_pd_targets switches the sim back to position drives."""
    from helpers import oracle_sim, scene_state, varied_models
    E = 48
    osim, gsim = _mk(E, seed=51)
    models = varied_models(E, 51)
    rng = np.random.default_rng(52)
    torque = (rng.normal(size=(E, 69)) * 120.0).astype(np.float32)
    torque[:, ::7] *= 8.0
    osim.params.drive_mode = 1
    osim.pd_target[:] = torque
    tq = torch.from_numpy(torque).to(gsim.device)
    for k in range(4):
        osim.step(1)
        gsim.set_dof_actuation_force(tq)
        gsim.step(2)
        _compare(osim, gsim, E, what=f"effort step {k}")
    lim = np.stack([m.effort for m in models]).astype(np.float32)
    assert np.array_equal(gsim.dof_force.view(E, 69).cpu().numpy(), np.clip(torque, -lim, lim))
    osim.params.drive_mode = 0
    osim.pd_target[:] = 0.0
    gsim.set_pd_targets(torch.zeros(E, 69, device=gsim.device))
    for k in range(2):
        osim.step(1)
        gsim.step(2)
        _compare(osim, gsim, E, what=f"position step {k}")


def test_stairs_and_map_border_stay_on_the_oracles_bytes():
    """The terrain probes one radius out (four per sphere) on a staircase of one-cell risers and where they leave the map (clamped border
    cells): humanoids standing over the first cell, mid map and the last cell, falling and tumbling over the steps for an episode."""
    from emloco_amd import _lib as L
    from emloco_amd.sim import NativeSim
    from helpers import oracle_sim, scene_state, varied_models
    E = 6
    models = varied_models(E, 41)
    root, dof, tgt = scene_state(E, 42, perturbed_from=0)
    n = 200
    steps = ((np.arange(n)[:, None] // 3) % 4 * 30 + 0 * np.arange(n)[None, :]).astype(np.int16)      # 15 cm risers every 30 cm
    hf = dict(samples=steps, horizontal_scale=0.1, vertical_scale=0.005)
    root[:, 0] = [0.04, 9.93, 19.88, 5.0, 5.15, 12.31]
    root[:, 1] = [0.03, 10.0, 19.86, 0.02, 8.0, 19.89]
    root[:, 2] = 1.05 + steps[np.clip((root[:, 0] / 0.1).astype(int), 0, n - 1), 0] * 0.005
    osim = oracle_sim(models, root, dof, tgt, heightfield=hf, n_sub=4)
    gsim = NativeSim(models, L.default_sim_params(n_sub=2), heightfield=hf)
    gsim.root_state.copy_(torch.from_numpy(root))
    gsim.dof_state.view(E, 69, 2).copy_(torch.from_numpy(dof))
    gsim.pd_target.copy_(torch.from_numpy(tgt))
    for k in range(120):
        osim.step(1)
        gsim.step(2)
        if k in (0, 3, 15, 50, 119):
            _compare(osim, gsim, E, what=f"stairs step {k}")
    assert torch.isfinite(gsim.rigid_body_state).all() and float(gsim.contact_force.abs().max()) > 50.0


def test_slope_corrected_mesh_stays_on_the_oracles_bytes():
    """Round 5: collision with the slope-corrected terrain mesh (emloco_sim_set_ground_mesh_moves: vertical risers along x, a trench
    with faces looking along y) -- humanoids dropped across risers, into the trench, over the map's border and walking speed into a
    face, for an episode: bytes of the oracle at every compared step, and a different trajectory from the raw height field's."""
    from emloco_amd import _lib as L
    from emloco_amd.sim import NativeSim
    from helpers import corrected_stairs, oracle_sim, scene_state, varied_models
    E = 6
    models = varied_models(E, 51)
    root, dof, tgt = scene_state(E, 52, perturbed_from=0)
    steps, hf = corrected_stairs()
    root[:, 0] = [0.04, 9.93, 10.21, 19.88, 5.15, 12.31]
    root[:, 1] = [0.03, 10.0, 9.62, 19.86, 8.0, 19.89]
    root[:, 2] = 1.0 + steps[np.clip((root[:, 0] / 0.1).astype(int), 0, 199), np.clip((root[:, 1] / 0.1).astype(int), 0, 199)] * 0.005
    root[:, 7] = [0.5, 1.5, -1.0, 0.0, 1.2, -0.8]
    root[:, 8] = [0.0, -1.0, 1.5, 0.0, 0.0, 0.3]
    osim = oracle_sim(models, root, dof, tgt, heightfield=hf, n_sub=4)
    gsim = NativeSim(models, L.default_sim_params(n_sub=2), heightfield=hf)
    raw = NativeSim(models, L.default_sim_params(n_sub=2), heightfield={k: v for k, v in hf.items() if not k.startswith("move")})
    for g in (gsim, raw):
        g.root_state.copy_(torch.from_numpy(root))
        g.dof_state.view(E, 69, 2).copy_(torch.from_numpy(dof))
        g.pd_target.copy_(torch.from_numpy(tgt))
    side = 0.0
    for k in range(120):
        osim.step(1)
        gsim.step(2)
        raw.step(2)
        side = max(side, float(gsim.contact_force[:, :2].abs().max()))
        if k in (0, 3, 15, 50, 119):
            _compare(osim, gsim, E, what=f"corrected mesh step {k}")
    assert torch.isfinite(gsim.rigid_body_state).all() and float(gsim.contact_force.abs().max()) > 50.0 and side > 20.0
    assert not torch.equal(gsim.rigid_body_state, raw.rigid_body_state)
