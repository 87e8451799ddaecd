"""Known-answer tests of the CPU physics oracle (oracle/oracle_sim.c).

The reference's engine (PhysX 5 behind gym.simulate) is absent, so the oracle cannot be pinned to reference
outputs ("parity unpinned").  These analytic cases pin OUR documented scheme instead; the HIP step
is then held bit-exact to this oracle (tests/test_gpu_sim.py, tests/test_emu_kernels.py).
"""
import numpy as np
import pytest

import oracle
from emloco_amd.model import pack_models, smpl_humanoid


def _sim(models=None, **p):
    return oracle.Sim(pack_models(models or [smpl_humanoid()]), oracle.default_params(**p))


def test_articulated_body_solve_equals_dense_solve():
    """ABA factorisation vs an independent dense M^-1 (body Jacobians, float64) at a random state."""
    s = _sim()
    rng = np.random.default_rng(0)
    s.root_state[0, :3] = [0, 0, 5.0]
    q = rng.normal(size=4)
    s.root_state[0, 3:7] = q / np.linalg.norm(q)
    s.root_state[0, 7:13] = rng.normal(size=6)
    s.dof_state[0, :, 0] = rng.normal(size=69) * 0.3
    s.dof_state[0, :, 1] = rng.normal(size=69)
    s.pd_target[0] = rng.normal(size=69) * 0.2
    M, rhs = s.dense_dynamics()
    assert np.abs(M - M.T).max() < 1e-6 and np.all(np.linalg.eigvalsh(M) > 0)
    x = np.linalg.solve(M, rhs)
    np.testing.assert_allclose(s.free_accel(), x, rtol=2e-5, atol=2e-5 * np.abs(x).max())


def test_free_fall_matches_semi_implicit_euler():
    s = _sim()
    s.root_state[0, :3] = [0, 0, 100.0]
    for _ in range(30):
        s.step()
    t, h = 30 * 4 / 120.0, 1 / 120.0
    assert abs(s.root_state[0, 2] - (100 - 0.5 * 9.81 * t * (t + h))) < 2e-3
    assert abs(s.root_state[0, 9] + 9.81 * t) < 1e-3
    np.testing.assert_allclose(s.dof_state[0, :, 0], 0, atol=1e-5)      # no internal motion in free fall with targets at rest


def test_pd_stand_carries_the_weight_without_penetrating():
    m = smpl_humanoid()
    s = _sim([m])
    s.root_state[0, :3] = [0, 0, 0.95]
    for _ in range(168):
        s.step()
    fz = s.contact_force[0, :, 2].sum()
    assert abs(fz - m.total_mass() * 9.81) / (m.total_mass() * 9.81) < 0.01
    assert np.abs(s.contact_force[0][[0, 1, 2, 5, 6, 9, 10, 11, 12, 13]]).max() == 0     # only the feet touch
    rb = s.rb_state[0]
    assert rb[:, 2].min() > -0.01 and 0.85 < rb[0, 2] < 0.95      # toe-box origin sits 7 mm below its sole
    assert np.abs(s.dof_state[0, :, 1]).max() < 1e-2                                    # at rest


def test_pd_drive_step_response_is_stable_and_converges():
    """One joint commanded to 0.5 rad in free fall: the implicit drive converges without overshoot blow-up."""
    s = _sim()
    s.root_state[0, :3] = [0, 0, 50.0]
    j = 3 * 15 + 1                     # L_Elbow y
    s.pd_target[0, j] = 0.5
    trace = []
    for _ in range(60):
        s.step()
        trace.append(s.dof_state[0, j, 0])
    trace = np.array(trace)
    assert abs(trace[-1] - 0.5) < 0.02 and trace.max() < 0.6 and np.isfinite(trace).all()


def test_random_targets_every_step_stay_bounded():
    """Exploration-sized random targets (sigma 0.3 rad, new every control step) on light, stiffly driven links: the implicit
    drives must not pump energy in -- in free space and on the ground alike.  (A drive clamped on its explicit torque
    estimate does: kd * qd alone exceeds the limit on a fast light link and the clamped torque then overshoots every
    substep -- the reason the effort limit is applied to the implicit torque.)"""
    m = smpl_humanoid()
    for z0, g in ((50.0, 0.0), (0.95, -9.81)):
        s = oracle.Sim(pack_models([m] * 2), oracle.default_params(gravity_z=g))
        s.root_state[:, :3] = [52.0, 55.0, z0]
        s.root_state[:, 7] = 1.5
        rng = np.random.default_rng(0)
        peak = 0.0
        for _ in range(90):
            s.pd_target[:] = rng.normal(size=(2, 69)) * 0.3
            s.step()
            peak = max(peak, float(np.abs(s.rb_state[:, :, 7:13]).max()))
        assert np.isfinite(s.rb_state).all() and peak < 40.0, (z0, peak)
        assert np.abs(s.dof_force).max() <= m.effort.max() * (1 + 1e-5)


def test_absurd_targets_degrade_gracefully():
    """targets thrown +-1 rad (sigma) around every control step, on sloped bumpy terrain with self-collision: far outside
    anything a policy does, but the state must stay finite and bounded (link-speed cap + implicit effort limits), since one
    non-finite env would poison a whole training batch"""
    from helpers import bumpy_heightfield
    from emloco_amd.model import pack_self_collision
    m = smpl_humanoid()
    E = 6
    s = oracle.Sim(pack_models([m] * E), oracle.default_params(), self_collision=pack_self_collision([m] * E),
                   heightfield=bumpy_heightfield(seed=5, amp=0.12, slope=0.2))
    s.root_state[:, :3] = [52.0, 55.0, 1.5]
    rng = np.random.default_rng(2)
    vmax = 0.0
    for _ in range(120):
        s.pd_target[:] = rng.normal(size=(E, 69))
        s.step()
        vmax = max(vmax, float(np.linalg.norm(s.root_state[:, 7:10], axis=1).max()))
    assert np.isfinite(s.rb_state).all() and np.isfinite(s.dof_state).all() and vmax < 40.0
    assert np.abs(s.rb_state[:, :, 10:13]).max() < 200.0


def test_effort_limit_caps_the_delivered_drive_torque():
    """a hip commanded 2.5 rad away: kp * err = 2000 N m >> effort 500: the drive delivers exactly the limit (constant
    torque), its neighbours stay implicit and below their limits, and the motion stays smooth"""
    m = smpl_humanoid()
    s = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0, n_sub=1))
    s.root_state[0, :3] = [52.0, 55.0, 50.0]
    j = (m.names.index("L_Hip") - 1) * 3 + 1
    s.pd_target[0, j] = 2.5
    s.step()                                         # one substep: dof_force reports the torque applied over it
    assert s.dof_force[0, j] == np.float32(m.effort[j])
    others = np.delete(s.dof_force[0], j)
    assert np.abs(others).max() < m.effort.max()
    trace = []
    for _ in range(240):
        s.step()
        trace.append(s.dof_state[0, j, 0])
    trace = np.array(trace)
    assert abs(trace[-1] - 2.5) < 0.05 and trace.max() < 2.7 and np.all(np.diff(trace[:10]) > 0)


def _com_pos(m, rb):
    rb = rb.astype(np.float64)
    out = np.zeros(3)
    for i in range(24):
        x, y, z, w = rb[i, 3:7]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        out += m.mass[i] * (rb[i, :3] + R @ m.com[i])
    return out / m.mass.sum()


def _momenta(m, rb):
    """total linear momentum, angular momentum about the centre of mass and kinetic energy from the body states (float64)"""
    def rot(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    rb = rb.astype(np.float64)
    R = [rot(rb[i, 3:7]) for i in range(24)]
    c = np.stack([rb[i, :3] + R[i] @ m.com[i] for i in range(24)])
    v = np.stack([rb[i, 7:10] + np.cross(rb[i, 10:13], R[i] @ m.com[i]) for i in range(24)])
    P = (m.mass[:, None] * v).sum(0)
    com = (m.mass[:, None] * c).sum(0) / m.mass.sum()
    L, T = np.zeros(3), 0.0
    for i in range(24):
        xx, yy, zz, xy, xz, yz = m.inertia[i]
        Iw = R[i] @ np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]) @ R[i].T
        w = rb[i, 10:13]
        L += m.mass[i] * np.cross(c[i] - com, v[i]) + Iw @ w
        T += 0.5 * m.mass[i] * v[i] @ v[i] + 0.5 * w @ Iw @ w
    return P, L, T


def test_free_floating_ragdoll_conserves_momentum_and_energy():
    """drives off, no gravity, no damping, tumbling and flailing at a few rad/s: over one second (120 substeps) the linear
    momentum is kept exactly (momentum balance), angular momentum stays within 4 %, kinetic energy within 5 % (first-order
    integrator).  Guards the re-basing of the root twist to the moving root origin as well: without it the linear momentum
    turns with w x v (87 % off here before the balance existed)."""
    m = smpl_humanoid().scaled(1.0, 1.0)
    m.kp, m.kd = m.kp * 0, m.kd * 0
    s = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0, ang_damping=0.0))
    rng = np.random.default_rng(0)
    s.root_state[0, :3] = [52, 55, 50]
    s.root_state[0, 7:10] = [0.5, -0.2, 0.1]
    s.root_state[0, 10:13] = rng.normal(size=3)
    s.dof_state[0, :, 1] = rng.normal(size=69)
    s.step()
    P0, L0, T0 = _momenta(m, s.rb_state[0])
    for _ in range(30):
        s.step()
    P1, L1, T1 = _momenta(m, s.rb_state[0])
    assert np.linalg.norm(P0) > 20 and np.linalg.norm(L0) > 5
    assert np.linalg.norm(P1 - P0) < 2e-5 * np.linalg.norm(P0)
    assert np.linalg.norm(L1 - L0) < 0.04 * np.linalg.norm(L0)
    assert abs(T1 - T0) < 0.05 * T0


def test_tumbling_flailing_free_fall_keeps_the_centre_of_mass_on_its_parabola():
    """gravity is the only external force: whatever the limbs do (targets thrown 0.3 rad off, root spinning at 4 rad/s),
    the velocity of the centre of mass is v0 + g t -- exactly, by the linear-momentum balance (it drifted by 5 % of g t
    without it) -- and its position follows the discrete parabola to a few centimetres over a second"""
    m = smpl_humanoid()
    s = oracle.Sim(pack_models([m]), oracle.default_params())
    s.root_state[0, :3] = [52, 55, 100]
    s.root_state[0, 7:10] = [1.0, -0.5, 2.0]
    s.root_state[0, 10:13] = [2.0, -3.0, 1.5]
    s.pd_target[0] = np.random.default_rng(0).normal(size=69) * 0.3
    M = m.mass.sum()

    def com(rb):
        return _com_pos(m, rb)
    s.step()
    c0, v0 = com(s.rb_state[0]), _momenta(m, s.rb_state[0])[0] / M
    n = 4 * 30
    for _ in range(30):
        s.step()
    c1, v1 = com(s.rb_state[0]), _momenta(m, s.rb_state[0])[0] / M
    h, g = 1.0 / 120.0, np.array([0, 0, -9.81])
    np.testing.assert_allclose(v1, v0 + g * h * n, atol=2e-4)
    np.testing.assert_allclose(c1, c0 + v0 * h * n + g * h * h * n * (n + 1) / 2, atol=0.12)


def test_hand_drive_step_response_is_the_one_dof_implicit_recurrence():
    """a distal light link on a heavy arm: the L_Hand joint commanded 0.4 rad follows the closed-form recurrence of the
    implicit PD drive of ONE degree of freedom, (I + h kd + h^2 kp) v+ = I v + h kp (x* - x), with I = hand inertia about
    the joint axis + armature -- to 1e-4 of the step"""
    m = smpl_humanoid()
    s = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0, n_sub=1))
    s.root_state[0, :3] = [52, 55, 50]
    b = m.names.index("L_Hand")
    j = (b - 1) * 3 + 1
    s.pd_target[0, j] = 0.4
    got = []
    for _ in range(60):
        s.step()
        got.append(s.dof_state[0, j, 0])
    yy, c = m.inertia[b][1], m.com[b]
    I = yy + m.mass[b] * (c[0] ** 2 + c[2] ** 2) + m.armature[j]
    h, x, v, ref = 1.0 / 120.0, 0.0, 0.0, []
    for _ in range(60):
        v = (I * v + h * m.kp[j] * (0.4 - x)) / (I + h * m.kd[j] + h * h * m.kp[j])
        x += h * v
        ref.append(x)
    np.testing.assert_allclose(got, ref, atol=1e-4 * 0.4)


def test_friction_holds_on_flat_ground_and_tangential_push_decays():
    s = _sim()
    s.root_state[0, :3] = [0, 0, 0.92]
    for _ in range(60):
        s.step()
    x0 = s.root_state[0, :2].copy()
    s.root_state[0, 7] = 0.3            # shove the pelvis forward
    for _ in range(90):
        s.step()
    assert np.linalg.norm(s.root_state[0, 7:9]) < 0.05         # friction (mu = 1) brought it back to rest
    assert np.linalg.norm(s.root_state[0, :2] - x0) < 0.3
    lam = s.contact_force[0]
    assert np.all(np.hypot(lam[:, 0], lam[:, 1]) <= 1.0 * lam[:, 2] + 1e-3)     # net force inside the friction cone


def test_left_right_mirror_symmetry():
    """Mirroring state and targets across the sagittal plane mirrors the trajectory (humanoid.py:334 permutation)."""
    m = smpl_humanoid()
    # make the model exactly symmetric first (the SMPL mean shape is not), then compare a motion with its mirror
    l2r = [0, 5, 6, 7, 8, 1, 2, 3, 4, 9, 10, 11, 12, 13, 19, 20, 21, 22, 23, 14, 15, 16, 17, 18]
    sym = m.scaled(1.0, 1.0)
    flip = np.array([1, -1, 1.0])
    for b in range(24):
        if l2r[b] > b:
            o = l2r[b]
            sym.joint_off[o] = sym.joint_off[b] * flip
            sym.com[o] = sym.com[b] * flip
            sym.geom_a[o], sym.geom_b[o] = sym.geom_a[b] * flip, sym.geom_b[b] * (flip if sym.geom_type[b] == 1 else 1)
            sym.geom_r[o], sym.mass[o] = sym.geom_r[b], sym.mass[b]
            sym.inertia[o] = sym.inertia[b] * np.array([1, 1, 1, -1, 1, -1])
        elif l2r[b] == b:
            sym.joint_off[b][1] = 0
            sym.com[b][1] = 0
            sym.geom_a[b][1] = 0
            sym.geom_b[b][1] = 0 if sym.geom_type[b] == 1 else sym.geom_b[b][1]
            sym.inertia[b][[3, 5]] = 0
    rng = np.random.default_rng(3)
    tgt = rng.normal(size=(23, 3)) * 0.2
    jm = [j - 1 for j in l2r[1:]]
    # (1) floating in zero gravity (no contact switching to amplify rounding): the mirrored run stays the mirror image
    # (2) dropped onto its feet: exact up to rounding at first, then contact switching amplifies last-ulp differences
    for grav, z0, early, late in ((0.0, 5.0, 2e-6, 2e-4), (-9.81, 0.95, 1e-6, 4e-2)):
        a = oracle.Sim(pack_models([sym]), oracle.default_params(gravity_z=grav))
        b = oracle.Sim(pack_models([sym]), oracle.default_params(gravity_z=grav))
        for s_, sgn in ((a, 1.0), (b, -1.0)):
            s_.root_state[0, :3] = [0, 0, z0]
            s_.root_state[0, 7:10] = [0.8, 0.3 * sgn, 0.1]                 # true vector: y flips
            s_.root_state[0, 10:13] = [0.5 * sgn, 0.7, -0.9 * sgn]         # pseudo vector: x and z flip
        a.pd_target[0] = tgt.reshape(-1)
        b.pd_target[0] = (tgt[jm] * np.array([-1, 1, -1.0])).reshape(-1)   # mirrored rotation vectors
        for k in range(40):
            a.step()
            b.step()
            if k == 1:
                np.testing.assert_allclose(a.rb_state[0, :, :3], b.rb_state[0, l2r, :3] * flip, atol=early)
        np.testing.assert_allclose(a.rb_state[0, :, :3], b.rb_state[0, l2r, :3] * flip, atol=late)
        np.testing.assert_allclose(a.rb_state[0, :, 7:10], b.rb_state[0, l2r, 7:10] * flip, atol=50 * late)


def _slope_field(slope, n=1100):
    """plane z = slope * (x - 50 m) on the 0.1 m / 0.005 m grid (slope * 20 must be an integer number of units per cell)"""
    per_cell = slope * 0.1 / 0.005
    assert abs(per_cell - round(per_cell)) < 1e-9
    col = (np.arange(n) - 500) * int(round(per_cell))
    return dict(samples=np.repeat(col[:, None], n, 1).astype(np.int16), horizontal_scale=0.1, vertical_scale=0.005)


def test_flat_heightfield_equals_the_plane_bit_for_bit():
    hf = dict(samples=np.zeros((1100, 1100), np.int16), horizontal_scale=0.1, vertical_scale=0.005)
    a = oracle.Sim(pack_models([smpl_humanoid()]), oracle.default_params(), heightfield=hf)
    b = _sim()
    for s_ in (a, b):
        s_.root_state[0, :3] = [52.3, 57.1, 0.93]
        s_.pd_target[0, ::5] = 0.2
        for _ in range(30):
            s_.step()
    assert a.rb_state.tobytes() == b.rb_state.tobytes() and a.contact_force.tobytes() == b.contact_force.tobytes()
    assert np.abs(a.contact_force).max() > 100


@pytest.mark.parametrize("slope", [0.05, -0.2, 0.4])
def test_resting_on_a_slope_holds_by_friction_and_leans_its_contact_forces(slope):
    """|slope| < mu = 1: the humanoid comes to rest and stays (the zero-target PD mannequin, soft in the ankles, cannot keep
    its balance on an incline: it topples and lies on the slope).  In equilibrium the summed contact force is the weight
    straight up, i.e. a normal part m g cos(a) plus an up-slope friction part m g sin(a)"""
    m = smpl_humanoid()
    s = oracle.Sim(pack_models([m]), oracle.default_params(), heightfield=_slope_field(slope))
    x0 = 52.0
    s.root_state[0, :3] = [x0, 55.0, 0.95 + slope * (x0 - 50.0)]
    for _ in range(260):
        s.step()
    W = m.total_mass() * 9.81
    f = s.contact_force[0].sum(0)
    assert abs(f[2] - W) / W < 0.05 and abs(f[0]) / W < 0.05 and abs(f[1]) / W < 0.02     # net force = weight, straight up
    n = np.array([-slope, 0, 1]) / np.hypot(slope, 1)
    fn = f @ n
    ft = np.linalg.norm(f - fn * n)
    assert abs(fn - W * np.cos(np.arctan(slope))) / W < 0.05 and abs(ft - W * np.sin(np.arctan(abs(slope)))) / W < 0.05
    assert np.linalg.norm(s.root_state[0, 7:10]) < 0.05                                   # at rest
    rb = s.rb_state[0]
    gap = rb[:, 2] - slope * (rb[:, 0] - 50.0)
    assert gap.min() > -0.02                                                              # nothing sank into the slope
    x1 = s.root_state[0, 0]
    for _ in range(30):
        s.step()
    assert abs(s.root_state[0, 0] - x1) < 1e-3                                            # held by friction


@pytest.mark.parametrize("slope,mu", [(1.5, 1.0), (1.0, 0.5), (0.5, 0.2)])
def test_sliding_down_a_slope_obeys_coulomb(slope, mu):
    """tan(a) > mu: once the body lies on the slope and slides, its centre of mass accelerates down the slope with
    g (sin a - mu cos a) and the ground carries m g cos a -- the friction-cone projection of the Gauss-Seidel solver,
    the per-contact tangent frames and the momentum balance together, to 1 %"""
    m = smpl_humanoid()
    M = m.mass.sum()
    s = oracle.Sim(pack_models([m]), oracle.default_params(mu=mu), heightfield=_slope_field(slope))
    x0 = 60.0
    s.root_state[0, :3] = [x0, 55.0, 0.95 + slope * (x0 - 50.0)]
    v = []
    for _ in range(81):
        s.step()
        v.append(_momenta(m, s.rb_state[0])[0] / M)
    al = np.arctan(slope)
    t = np.array([1, 0, slope]) / np.hypot(slope, 1)            # up-slope tangent
    n = np.array([-slope, 0, 1]) / np.hypot(slope, 1)
    acc = (v[80] - v[50]) @ t / 1.0                              # 30 control steps = 1 s
    want = -9.81 * (np.sin(al) - mu * np.cos(al))
    assert abs(acc - want) < 0.01 * abs(want)
    assert abs(v[80] @ n) < 0.02                                  # stays on the surface
    assert abs(s.contact_force[0].sum(0) @ n / (M * 9.81) - np.cos(al)) < 0.01


def test_too_steep_a_slope_slides_downhill():
    """slope 1.5 (56 deg) > mu = 1: friction cannot hold, the body accelerates down the slope (-x)"""
    slope = 1.5
    s = oracle.Sim(pack_models([smpl_humanoid()]), oracle.default_params(), heightfield=_slope_field(slope))
    x0 = 55.0
    s.root_state[0, :3] = [x0, 55.0, 0.95 + slope * (x0 - 50.0)]
    for _ in range(60):
        s.step()
    assert s.root_state[0, 0] < x0 - 1.0 and s.root_state[0, 7] < -1.0 and np.isfinite(s.rb_state).all()


def test_dropped_on_stairs_comes_to_rest_above_the_steps():
    """pyramid stairs (0.15 m risers): a ragdoll-ish drop ends with every body above the terrain under it (small
    penetration slack) and the whole weight carried"""
    from emloco_amd.gym import terrain_utils as T
    t = T.SubTerrain("terrain", width=160, length=160, vertical_scale=0.005, horizontal_scale=0.1)
    T.pyramid_stairs_terrain(t, step_width=0.31, step_height=0.15, platform_size=3.)
    field = np.zeros((1100, 1100), np.int16)
    field[470:630, 470:630] = t.height_field_raw
    m = smpl_humanoid()
    s = oracle.Sim(pack_models([m]), oracle.default_params(), heightfield=dict(samples=field, horizontal_scale=0.1, vertical_scale=0.005))
    W = m.total_mass() * 9.81
    for x, y in ((49.2, 55.0), (48.05, 55.0)):            # over the 7th / 3rd step of the pyramid's side
        s = oracle.Sim(pack_models([m]), oracle.default_params(), heightfield=dict(samples=field, horizontal_scale=0.1, vertical_scale=0.005))
        s.root_state[0, :3] = [x, y, field[int(x / 0.1), int(y / 0.1)] * 0.005 + 1.1]
        for _ in range(200):                              # lands on the steps, topples, tumbles down and stops
            s.step()
        rb = s.rb_state[0]
        ix = np.clip((rb[:, 0] / 0.1).astype(int), 0, 1098)
        iy = np.clip((rb[:, 1] / 0.1).astype(int), 0, 1098)
        ground = np.minimum(field[ix, iy], field[ix + 1, iy + 1]) * 0.005
        assert (rb[:, 2] - ground).min() > -0.02 and np.isfinite(rb).all()
        assert abs(s.contact_force[0, :, 2].sum() - W) / W < 0.02 and np.abs(s.root_state[0, 7:13]).max() < 0.05


def test_same_inputs_same_bytes():
    a, b = _sim(), _sim()
    for s_ in (a, b):
        s_.root_state[0, :3] = [0, 0, 0.93]
        s_.pd_target[0, ::7] = 0.3
        for _ in range(25):
            s_.step()
    assert a.rb_state.tobytes() == b.rb_state.tobytes()


def names_index(m, name):
    return m.names.index(name)


def test_ankle_boxes_collide_with_their_width():
    """The SMPL ankle boxes are 17 x 9.7 x 4.2 cm: one capsule down the middle (rounds 1-3) leaves 2.8 cm of each side uncovered.
    Two capsules along the long edges: with the legs adducted so that the two ankle boxes overlap sideways by ~1 cm while their
    centre lines are still 8.7 cm apart (far more than two 2.1 cm radii), a limb-limb contact between L_Ankle and R_Ankle exists and
    pushes them apart along y; with the feet a box width + 2 cm apart there is none."""
    from emloco_amd.model import pack_self_collision
    m = smpl_humanoid()
    la, ra = names_index(m, "L_Ankle"), names_index(m, "R_Ankle")

    def foot_contacts(adduct):
        s = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0), self_collision=pack_self_collision([m]))
        s.root_state[0, :3] = [0, 0, 3.0]
        hl, hr = names_index(m, "L_Hip"), names_index(m, "R_Hip")
        s.dof_state[0, (hl - 1) * 3 + 0, 0] = -adduct          # hip rotation about x swings the leg sideways
        s.dof_state[0, (hr - 1) * 3 + 0, 0] = adduct
        s.fk()
        rb = s.rb_state[0]
        gap = abs((rb[la, 1] + m.geom_a[la][1]) - (rb[ra, 1] + m.geom_a[ra][1]))       # centre-line distance of the two boxes (y)
        return gap, [c for c in s.self_contacts() if {int(c[0]), int(c[1])} == {la, ra}]
    # find the adduction that brings the box centre lines to ~8.7 cm (boxes are 9.66 cm wide: ~1 cm of overlap)
    lo, hi = 0.0, 0.3
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        gap, _c = foot_contacts(mid)
        lo, hi = (mid, hi) if gap > 0.087 else (lo, mid)
    gap, cont = foot_contacts(hi)
    assert 0.080 < gap < 0.0875 and len(cont) >= 1
    n = cont[0][5:8]
    assert abs(n[1]) > 0.9 and cont[0][8] > 0                  # pushed apart sideways
    gap0, none = foot_contacts(0.0)
    assert gap0 > 0.0966 + 0.02 and none == []


def test_self_collision_keeps_limbs_apart():
    """Limb-limb penalty contacts (has_self_collision): an arm driven into the trunk is held near the surface instead of
    passing through (with self-collision off the same drive penetrates), and the equal-and-opposite contact wrenches do
    not push the floating body as a whole."""
    from emloco_amd.model import collision_capsules, pack_self_collision, self_collision_pairs
    m = smpl_humanoid()
    pairs = self_collision_pairs(m)
    sb = collision_capsules(m)[3]                               # 26 segments: one per body + the second capsule of each ankle box
    assert len(sb) == 26 and list(sb[24:]) == [names_index(m, "L_Ankle"), names_index(m, "R_Ankle")]
    assert len(pairs) == 286 and all(sb[i] != sb[j] and m.parent[sb[j]] != sb[i] and m.parent[sb[i]] != sb[j] for i, j in pairs)   # no same-body / parent-child pairs
    assert (11, 13) not in set(map(tuple, pairs.tolist()))                                          # Chest-Head: filter bits 192 & 64
    names = m.names
    sh = names.index("L_Shoulder")

    def run(sc):
        s = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0), self_collision=pack_self_collision([m]) if sc else None)
        s.root_state[0, :3] = [0, 0, 3.0]                       # floating, no ground contact, no gravity
        s.pd_target[0, (sh - 1) * 3 + 0] = -2.5                 # drive the left arm (T-pose) down and into the trunk / hip
        p0 = None
        depth = []
        a, b, r, _sb = collision_capsules(m)
        for k in range(60):
            s.step()
            rb = s.rb_state[0]
            def rot(q):
                x, y, z, w = q
                return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                                 [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                                 [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            vc = np.stack([rb[i, 7:10] + np.cross(rb[i, 10:13], rot(rb[i, 3:7]) @ m.com[i]) for i in range(24)])
            com_v = (m.mass[:, None] * vc).sum(0) / m.mass.sum()       # true linear momentum / mass
            # deepest overlap between the left fore-arm / hand chain and the torso chain capsules
            import itertools
            worst = 0.0
            for i, j in itertools.product([names.index(n) for n in ("L_Elbow", "L_Wrist", "L_Hand")],
                                          [names.index(n) for n in ("Torso", "Spine", "Chest", "Pelvis", "L_Hip", "R_Hip")]):
                from emloco_amd.model import _segment_distance
                def world(bi, pt):
                    q = rb[bi, 3:7]
                    x, y, z, w = q
                    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
                    return rb[bi, :3] + R @ pt
                dist = _segment_distance(world(i, a[i]), world(i, b[i]), world(j, a[j]), world(j, b[j]))
                worst = max(worst, r[i] + r[j] - dist)
            depth.append(worst)
        return np.array(depth), com_v, s

    d_on, v_on, s_on = run(True)
    d_off, v_off, _ = run(False)
    assert d_off.max() > 0.04                                   # without self-collision the arm sinks into the torso
    assert d_on.max() < 0.5 * d_off.max() and d_on[-10:].max() < 0.03      # with it the overlap stays shallow
    assert np.abs(v_on - v_off).max() < 5e-3                    # internal forces only: no net push on the floating body
    assert np.isfinite(s_on.rb_state).all()


def test_limb_limb_friction_is_coulomb_capped_by_the_damper_and_internal():
    """Friction in the limb-limb contacts: at every contact the force on body i is
    n F_n - min(mu F_n / |v_t|, c) v_t with v_t the tangential relative velocity of the two bodies at the contact point
    (recomputed here in float64 from the body states), i.e. inside the Coulomb cone, opposing the sliding, and never stiffer
    than the contact's normal damper (what keeps the explicit force stable); mu = 0 gives the bare normal force; being equal
    and opposite at one point it leaves the momentum of the floating body alone; its power is negative (it only dissipates)."""
    from emloco_amd.model import pack_self_collision
    m = smpl_humanoid()
    sh = m.names.index("L_Shoulder")

    def run(mu, steps=60, check=False):
        sc = pack_self_collision([m], mu=mu)
        s = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0), self_collision=sc)
        s.root_state[0, :3] = [0, 0, 3.0]
        s.pd_target[0, (sh - 1) * 3 + 0] = -2.5                 # the left arm swings down into the trunk / hip ...
        s.pd_target[0, (sh - 1) * 3 + 1] = 0.6                  # ... and drags along it
        slide, checked, sliding_rows, capped_rows = [], 0, 0, 0
        for k in range(steps):
            s.pd_target[0, (sh - 1) * 3 + 1] = 0.6 + 0.5 * np.sin(0.5 * k)      # keep it sliding
            s.step()
            rows = s.self_contacts()
            rb = s.rb_state[0].astype(np.float64)
            for r in rows:
                bi, bj = int(r[0]), int(r[1])
                p = rb[0, :3] + r[2:5]
                n, Fn, F = r[5:8].astype(np.float64), float(r[8]), r[9:12].astype(np.float64)
                vi = rb[bi, 7:10] + np.cross(rb[bi, 10:13], p - rb[bi, :3])
                vj = rb[bj, 7:10] + np.cross(rb[bj, 10:13], p - rb[bj, :3])
                vr = vi - vj
                vt = vr - vr.dot(n) * n
                slide.append(float((F - F.dot(n) * n).dot(vt)))          # power of the tangential force [W]
                if check:
                    g = min(mu * Fn / max(np.linalg.norm(vt), 1e-9), sc["c"])
                    want = n * Fn - g * vt
                    np.testing.assert_allclose(F, want, rtol=2e-3, atol=2e-3 * max(Fn, 1.0))
                    ft = F - F.dot(n) * n
                    assert np.linalg.norm(ft) <= mu * Fn * (1 + 1e-3) + 1e-3 and ft.dot(vt) <= 1e-6
                    sliding_rows += np.linalg.norm(vt) > 1e-3
                    capped_rows += mu * Fn / max(np.linalg.norm(vt), 1e-9) > sc["c"]
                    checked += 1
                elif mu == 0.0:
                    np.testing.assert_allclose(F, n * Fn, rtol=1e-6, atol=1e-6)
        return s, np.array(slide), checked, sliding_rows, capped_rows

    s1, slide1, checked, sliding_rows, capped_rows = run(1.0, check=True)
    s0, slide0, _, _, _ = run(0.0)
    assert checked > 40 and sliding_rows > 20, (checked, sliding_rows)
    # at mu = 1 and these contact forces (100-300 N) the damper cap is the active bound below ~3 m/s of sliding; a slippery
    # mu puts the same contacts on the Coulomb cone
    _, _, checked_s, _, capped_s = run(0.01, check=True)
    assert capped_rows > 0.5 * checked and checked_s > 40 and checked_s - capped_s >= 5, (capped_rows, checked, capped_s, checked_s)
    P1 = _momenta(m, s1.rb_state[0])[0]
    assert np.linalg.norm(P1) < 2e-3 * m.mass.sum()                             # internal forces: the floating body does not drift
    assert slide1.sum() < -1.0 and np.all(slide1 <= 1e-6) and np.abs(slide0).max() < 1e-3, (slide1.sum(), np.abs(slide0).max())   # dissipative; none without mu


def test_effort_drive_is_the_constant_torque_recurrence_and_respects_the_limit():
    """Effort drives (drive_mode = 1, `pdControl: False`, humanoid.py:1203-1207): a constant torque on the L_Hand joint of a
    free-floating humanoid gives that joint the closed-form constant acceleration tau / (I + armature) -- the hand is light on a
    heavy arm, angular damping off -- with no drive stiffness or damping in the way, the reported dof force is the commanded
    torque, and a command beyond the effort limit is applied (and reported) at the limit."""
    m = smpl_humanoid()
    s = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0, n_sub=1, ang_damping=0.0, drive_mode=1))
    s.root_state[0, :3] = [52, 55, 50]
    b = m.names.index("L_Hand")
    j = (b - 1) * 3 + 1
    tau = 0.02
    s.pd_target[0, j] = tau
    got = []
    for _ in range(30):
        s.step()
        got.append(s.dof_state[0, j, 1])
    yy, c = m.inertia[b][1], m.com[b]
    I = yy + m.mass[b] * (c[0] ** 2 + c[2] ** 2) + m.armature[j]
    ref = tau / I * (1.0 / 120.0) * np.arange(1, 31)
    np.testing.assert_allclose(got, ref, rtol=0.02)                    # the arm gives way a little: 2 %
    assert abs(s.dof_force[0, j] - tau) < 1e-7
    others = np.delete(np.arange(69), [j])
    assert np.abs(s.dof_force[0, others]).max() == 0.0
    s2 = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0, n_sub=1, drive_mode=1))
    s2.root_state[0, :3] = [52, 55, 50]
    s2.pd_target[0, j] = 10.0 * m.effort[j]
    s2.step()
    assert s2.dof_force[0, j] == np.float32(m.effort[j])
    assert np.isfinite(s2.dof_state).all()


# ------------------------------------------------------------------------------------------------ angular-momentum balance
def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _composite_inertia(m, rb):
    """inertia tensor of the whole humanoid about its centre of mass with the joints locked (float64, world axes)"""
    rb = rb.astype(np.float64)
    R = [_rot(rb[i, 3:7]) for i in range(24)]
    c = np.stack([rb[i, :3] + R[i] @ m.com[i] for i in range(24)])
    com = (m.mass[:, None] * c).sum(0) / m.mass.sum()
    J = np.zeros((3, 3))
    for i in range(24):
        xx, yy, zz, xy, xz, yz = m.inertia[i]
        d = c[i] - com
        J += R[i] @ np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]) @ R[i].T + m.mass[i] * ((d @ d) * np.eye(3) - np.outer(d, d))
    return J


@pytest.mark.parametrize("spin", [13.0, 40.0, 80.0])
def test_spinning_humanoid_keeps_its_angular_momentum_up_to_the_assets_speed_limit(spin):
    """A free-floating humanoid holding its pose with the shipped PD drives, spun at 13 / 40 / 80 rad/s (the asset's limit is 100,
    humanoid.py:685-688; the scheme capped links at 48 rad/s before the angular-momentum balance existed and a body tumbling at
    13 rad/s gained 50 % kinetic energy in a second): over one second the angular momentum about the centre of mass is kept to
    1e-4 in size AND direction, the linear momentum to 1e-4, no link is slowed by a cap, and the kinetic energy never RISES --
    it settles as the limbs are flung outwards against the drives' dampers (L fixed, inertia up, T = L^2 / 2I down)."""
    m = smpl_humanoid().scaled(1.0, 1.0)
    s = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0, ang_damping=0.0))
    rng = np.random.default_rng(0)
    s.root_state[0, :3] = [52, 55, 50]
    s.root_state[0, 7:10] = [0.5, -0.2, 0.1]
    ax = rng.normal(size=3)
    s.root_state[0, 10:13] = ax / np.linalg.norm(ax) * spin
    s.fk()
    _, Linit, _ = _momenta(m, s.rb_state[0])                           # the momentum the initial state carries ...
    s.step()
    P0, L0, T0 = _momenta(m, s.rb_state[0])
    assert np.linalg.norm(L0 - Linit) < 2e-3 * np.linalg.norm(Linit)   # ... is there after the first step (the balance starts with the
    Ts = []                                                            # second substep of a call): no cap took any of it away
    for _ in range(30):
        s.step()
        P1, L1, T1 = _momenta(m, s.rb_state[0])
        Ts.append(T1 / T0)
        assert np.linalg.norm(L1 - L0) < 1e-4 * np.linalg.norm(L0)
    assert np.linalg.norm(P1 - P0) < 1e-4 * np.linalg.norm(P0)
    assert max(Ts) < 1.005 and min(Ts) > 0.3 and np.isfinite(s.rb_state).all()


def test_torque_free_precession_converges_to_the_closed_form_at_first_order():
    """Torque-free motion of an asymmetric rigid body against the closed form (Jacobi elliptic functions): the humanoid with drives
    300x stiffer than shipped is one rigid body with principal moments 2.0 < 11.2 < 12.9 kg m^2; spun at 10 rad/s about an axis
    between its minor and major axes its body-frame rate is (A1 cn(pt), A2 sn(pt), A3 dn(pt)).  The scheme -- angular momentum
    held exactly, orientation advanced with the rate that momentum implies -- is a first-order (Lie-Euler) integrator of that motion:
    over 0.2 s (40 % of the precession period) the error is 14 % of the spin at the shipped h = 1/120 s and halves with h
    (7.5 %, 3.9 %, 2.0 % at h / 2, h / 4, h / 8) -- it converges to the closed form, at the order the documentation claims."""
    from scipy.special import ellipj

    def run(n_sub, W=10.0):
        m = smpl_humanoid()
        m.kp, m.kd, m.effort = m.kp * 300, m.kd * 300, m.effort * 1e6
        s = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0, ang_damping=0.0, n_sub=n_sub, h=1.0 / 30 / n_sub))
        s.root_state[0, :3] = [52, 55, 50]
        s.fk()
        I, Pm = np.linalg.eigh(_composite_inertia(m, s.rb_state[0]))      # I1 < I2 < I3, principal axes in the root frame
        if np.linalg.det(Pm) < 0:
            Pm[:, 2] *= -1
        wb0 = np.array([0.5, 0.0, 0.8])
        wb0 = wb0 / np.linalg.norm(wb0) * W
        s.root_state[0, 10:13] = Pm @ wb0
        L2, T2 = ((I * wb0) ** 2).sum(), (I * wb0 * wb0).sum()
        I1, I2, I3 = I
        A1 = np.sqrt((T2 * I3 - L2) / (I1 * (I3 - I1)))
        A2 = np.sqrt((T2 * I3 - L2) / (I2 * (I3 - I2)))
        A3 = np.sqrt((L2 - T2 * I1) / (I3 * (I3 - I1)))
        p = np.sqrt((I3 - I2) * (L2 - T2 * I1) / (I1 * I2 * I3))
        k2 = (I2 - I1) * (T2 * I3 - L2) / ((I3 - I2) * (L2 - T2 * I1))
        assert 0 < k2 < 1 and abs(A1 - wb0[0]) < 1e-9 and abs(A3 - wb0[2]) < 1e-9
        err = 0.0
        for step in range(1, 7):
            s.step()
            R0 = _rot(s.rb_state[0][0, 3:7].astype(np.float64))
            wb = Pm.T @ R0.T @ s.rb_state[0][0, 10:13].astype(np.float64)
            sn, cn, dn, _ = ellipj(p * step / 30.0, k2)
            err = max(err, np.linalg.norm(wb - np.array([A1 * cn, A2 * sn, A3 * dn])) / W)
        _, L, T = _momenta(m, s.rb_state[0])
        assert abs(np.linalg.norm(L) / np.sqrt(L2) - 1) < 2e-4 and 0.9 < 2 * T / T2 < 1.001
        return err
    e = [run(n) for n in (4, 8, 16, 32)]
    assert e[0] < 0.16 and e[3] < 0.025
    for a, b in zip(e[:-1], e[1:]):
        assert 0.4 < b / a < 0.62, e                   # first order: halving h halves the error


def test_free_ragdoll_at_moderate_spin_keeps_momentum_exactly_and_energy_within_the_first_order_bound():
    """Drives off, no gravity, no damping -- a ragdoll tumbling at 13 rad/s with flailing limbs, the case DESIGN.md used to quote as
    +50 % kinetic energy / -30 % angular momentum per second: the angular momentum is now exact (1e-5); the kinetic energy of the
    free INTERNAL motion is still first order in h (the joints' velocity products are explicit): it stays within [-5 %, +45 %] over
    the second here.  (Above ~20 rad/s a torque-free ragdoll's internal motion is not integrated faithfully -- documented; with
    the PD drives of the product the test above holds to 80 rad/s.)"""
    m = smpl_humanoid().scaled(1.0, 1.0)
    m.kp, m.kd = m.kp * 0, m.kd * 0
    s = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0, ang_damping=0.0))
    rng = np.random.default_rng(0)
    s.root_state[0, :3] = [52, 55, 50]
    s.root_state[0, 7:10] = [0.5, -0.2, 0.1]
    ax = rng.normal(size=3)
    s.root_state[0, 10:13] = ax / np.linalg.norm(ax) * 13.0
    s.dof_state[0, :, 1] = rng.normal(size=69) * 0.5
    s.step()
    P0, L0, T0 = _momenta(m, s.rb_state[0])
    Ts = []
    for _ in range(30):
        s.step()
        P1, L1, T1 = _momenta(m, s.rb_state[0])
        Ts.append(T1 / T0)
    assert np.linalg.norm(L1 - L0) < 1e-5 * np.linalg.norm(L0) and np.linalg.norm(P1 - P0) < 1e-4 * np.linalg.norm(P0)
    assert 0.95 < min(Ts) and max(Ts) < 1.45


def test_sphere_sliding_into_a_step_stops_at_the_face():
    """Terrain contacts of a sphere with a radius look one radius ahead (four probes along +-x / +-y): the head
    (r = 0.101 m) of a humanoid sliding on its back, head first, over low-friction ground into a 20 cm riser (one 10 cm cell wide: a
    63-degree ramp in the height field) is stopped when its SURFACE reaches the ramp -- centre 0.618 r before the ramp's foot, the point
    where a sphere resting on the floor touches a plane of slope 2 -- not when its centre has crossed into the ramp's cell (measured
    without the probes: the centre reaches x = 52.001, 57 mm inside the ramp).  No sphere or capsule end of the body is ever deeper
    than a few millimetres in the terrain solid."""
    m = smpl_humanoid()
    field = np.zeros((1100, 1100), np.int16)
    field[521:, :] = 40                                           # 40 x 0.005 = 0.2 m from x = 52.1 on; the cell [52.0, 52.1) is the ramp
    s = oracle.Sim(pack_models([m]), oracle.default_params(mu=0.05), heightfield=dict(samples=field, horizontal_scale=0.1, vertical_scale=0.005))
    s.root_state[0, :3] = [50.6, 55.0, 0.16]
    s.root_state[0, 3:7] = [0.0, np.sin(np.pi / 4), 0.0, np.cos(np.pi / 4)]      # on its back, head towards +x
    s.root_state[0, 7] = 2.0

    def rot(q, v):
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        return R @ v

    def seg(p, a, b):
        ab = b - a
        return np.linalg.norm(p - (a + np.clip((p - a) @ ab / (ab @ ab), 0, 1) * ab))

    F0, A, B, T = np.array([0.0, 0.0]), np.array([52.0, 0.0]), np.array([52.1, 0.2]), np.array([70.0, 0.2])   # the profile in (x, z)
    head = m.names.index("Head")
    r_head = float(m.geom_r[head])
    worst, head_x, first_touch = 0.0, [], None
    for _ in range(60):
        s.step()
        rb = s.rb_state[0]
        for b in range(24):
            gt, r = int(m.geom_type[b]), float(m.geom_r[b])
            ends = [m.geom_a[b]] if gt == 0 else ([m.geom_a[b], m.geom_b[b]] if gt == 1 else [])
            for lp in ends:
                c = rb[b, :3] + rot(rb[b, 3:7], np.asarray(lp, float))
                p = np.array([c[0], c[2]])
                ground = 0.0 if p[0] < 52.0 else (2.0 * (p[0] - 52.0) if p[0] < 52.1 else 0.2)
                dist = min(seg(p, F0, A), seg(p, A, B), seg(p, B, T)) * (-1.0 if p[1] < ground else 1.0)
                worst = max(worst, r - dist)
        hc = rb[head, :3] + rot(rb[head, 3:7], np.asarray(m.geom_a[head], float))
        head_x.append(float(hc[0]))
        if first_touch is None and seg(np.array([hc[0], hc[2]]), A, B) - r_head < 0.004:
            first_touch = (float(hc[0]), float(hc[2]))
    touch = 52.0 - r_head * (np.sqrt(5.0) - 1.0) / 2.0             # a sphere on the floor touches a plane of slope 2 at 0.618 r before its foot
    assert worst < 0.008, worst
    assert first_touch is not None and abs(first_touch[0] - touch) < 0.012 and abs(first_touch[1] - r_head) < 0.01, (first_touch, touch)
    assert np.isfinite(s.rb_state).all()


# ------------------------------------------------------------------ the slope-corrected terrain mesh (round 5)
def _riser_field(up):
    """A 20 cm step across x on a 110 m map, as the raw field and with the vertex moves of its slope-corrected mesh
    (terrain_utils.convert_heightfield_to_trimesh, threshold 0.9).  up: low ground for x < 52.1, the riser's face at x = 52.1
    (the raw field has a ramp over [52.0, 52.1] instead); not up: a 0.2 m tread up to x = 51.9, the face there, low ground beyond."""
    from emloco_amd.gym import terrain_utils as T
    field = np.zeros((1100, 1100), np.int16)
    if up:
        field[521:, :] = 40
    else:
        field[:520, :] = 40
    verts, _ = T.convert_heightfield_to_trimesh(field, 0.1, 0.005, 0.9)
    mx, my = T.mesh_vertex_moves(verts, field.shape, 0.1)
    face_x = np.unique(verts.reshape(1100, 1100, 3)[519:522, 0, 0])
    assert (mx != 0).sum() == 1100 and np.allclose(face_x, [51.9, 52.1])       # (the corner rule moves the same vertices one cell along y too)
    return dict(samples=field, horizontal_scale=0.1, vertical_scale=0.005, move_x=mx, move_y=my)


def _qrot(q, v):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R @ v


def _box_corners(m, rb):
    out = []
    for b in range(24):
        if int(m.geom_type[b]) == 2:
            for k in range(8):
                sgn = np.array([1 if k & 1 else -1, 1 if k & 2 else -1, 1 if k & 4 else -1])
                out.append(rb[b, :3] + _qrot(rb[b, 3:7], np.asarray(m.geom_a[b], float) + np.asarray(m.geom_b[b], float) * sgn))
    return np.array(out)


def test_sphere_sliding_into_a_riser_of_the_corrected_mesh_stops_with_its_surface_at_the_face():
    """Collision with the slope-corrected mesh (emloco_sim_set_ground_mesh_moves / oracle hf_mv): the scene of
    test_sphere_sliding_into_a_step_stops_at_the_face, but the riser is the mesh's VERTICAL face at x = 52.1 -- the head (r = 0.101 m)
    sliding in at 2 m/s is stopped with its surface at the face (not 0.62 r early on a ramp, not late), never deeper than 4 mm in it,
    and stays at the height it slid at: there is no ramp to climb."""
    m = smpl_humanoid()
    s = oracle.Sim(pack_models([m]), oracle.default_params(mu=0.05), heightfield=_riser_field(True))
    s.root_state[0, :3] = [50.6, 55.0, 0.16]
    s.root_state[0, 3:7] = [0.0, np.sin(np.pi / 4), 0.0, np.cos(np.pi / 4)]      # on its back, head towards +x
    s.root_state[0, 7] = 2.0
    head = m.names.index("Head")
    r = float(m.geom_r[head])
    surf, zc = [], []
    for _ in range(60):
        s.step()
        rb = s.rb_state[0]
        hc = rb[head, :3] + _qrot(rb[head, 3:7], np.asarray(m.geom_a[head], float))
        surf.append(float(hc[0]) + r)
        zc.append(float(hc[2]))
    assert max(surf) < 52.1 + 0.004, max(surf)                           # never more than 4 mm into the face
    assert abs(surf[-1] - 52.1) < 0.003 and abs(surf[-1] - surf[-10]) < 1e-3, surf[-10:]     # at rest AT the face
    assert max(np.abs(np.array(zc[15:]) - r)) < 0.004                    # on the floor throughout
    assert np.abs(s.root_state[0, 7:13]).max() < 0.02 and np.isfinite(s.rb_state).all()


def test_foot_box_one_centimetre_from_a_tread_edge_stays():
    """A humanoid standing on a 0.2 m tread with the front corners of its toe boxes 1 cm from the edge (the corrected mesh's face at
    x = 51.9): the corners rest a hair inside the tread AND 1 cm behind the riser's face -- the contact stays with the tread (the
    nearest surface), the feet are not pushed off sideways: nothing moves, the whole weight is carried, no horizontal force."""
    m = smpl_humanoid()
    s = oracle.Sim(pack_models([m]), oracle.default_params(), heightfield=_riser_field(False))
    s.root_state[0, :3] = [51.9 - 0.01 - 0.1500, 55.0, 0.95 + 0.2]        # a standing humanoid's front corners sit 0.1500 m ahead of its start
    xs = []
    for k in range(240):
        s.step()
        xs.append(_box_corners(m, s.rb_state[0])[:, 0].max())
    c = _box_corners(m, s.rb_state[0])
    W = m.total_mass() * 9.81
    assert abs(xs[-1] - 51.89) < 0.002 and max(xs) < 51.893 and min(xs[20:]) > 51.887, (xs[-1], max(xs), min(xs[20:]))
    assert abs(c[:, 2].min() - 0.2) < 0.002
    assert abs(s.contact_force[0, :, 2].sum() - W) / W < 0.01 and np.abs(s.contact_force[0, :, :2].sum(0)).max() < 2.0
    assert np.abs(s.root_state[0, 7:13]).max() < 0.01


def test_box_corner_inside_a_riser_is_pushed_back_out_through_the_face():
    """A standing humanoid whose toe boxes start 8 mm INSIDE the riser (face at x = 52.1, the tread's top 0.2 m above the corners): the
    corners are 0.2 m below the tread's plane and 8 mm behind the face -- the face is the nearest surface, the contact normal is the
    face's and the feet are pushed back out horizontally (front corners end just before the face, the humanoid keeps standing on the
    low ground).  Picking the DEEPEST penetration instead would fire them up through the tread."""
    m = smpl_humanoid()
    s = oracle.Sim(pack_models([m]), oracle.default_params(), heightfield=_riser_field(True))
    s.root_state[0, :3] = [52.1 + 0.008 - 0.1500, 55.0, 0.95]
    for k in range(120):
        s.step()
        c = _box_corners(m, s.rb_state[0])
        assert c[:, 2].min() > -0.005 and c[:, 2].max() < 0.08, (k, c[:, 2].min(), c[:, 2].max())    # the feet never leave the low ground
    assert 52.09 < c[:, 0].max() < 52.1005, c[:, 0].max()
    W = m.total_mass() * 9.81
    assert abs(s.contact_force[0, :, 2].sum() - W) / W < 0.01 and 0.85 < s.root_state[0, 2] < 0.95
