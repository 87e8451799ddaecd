"""Shared scene builders for the parity tests (oracle vs HIP / emulated HIP)."""
import numpy as np

from emloco_amd.model import pack_models, smpl_humanoid


def varied_models(E, seed=0):
    """Synthetic 'AMASS-shaped' humanoids: limb scale U[0.9,1.1], mass scale U[0.7,1.4], PD gains x mass/77
    (humanoid.py:905-911)."""
    rng = np.random.default_rng(seed)
    base = smpl_humanoid()
    out = []
    for e in range(E):
        m = base.scaled(rng.uniform(0.9, 1.1), rng.uniform(0.7, 1.4)) if e else base.scaled(1.0, 1.0)
        s = m.total_mass() / 77.0
        m.kp = m.kp * s
        m.kd = m.kd * s
        out.append(m)
    return out


def scene_state(E, seed=1, height=0.93, perturbed_from=1):
    """root/dof/target arrays: env 0 stands still, the others start perturbed."""
    rng = np.random.default_rng(seed)
    root = np.zeros((E, 13), np.float32)
    root[:, 6] = 1.0
    root[:, 2] = height
    root[:, 0] = np.arange(E) * 0.5 + 50.0
    root[:, 1] = 55.0
    dof = np.zeros((E, 69, 2), np.float32)
    tgt = np.zeros((E, 69), np.float32)
    for e in range(perturbed_from, E):
        dof[e, :, 0] = rng.normal(size=69) * 0.15
        dof[e, :, 1] = rng.normal(size=69) * 0.8
        root[e, 7:13] = rng.normal(size=6) * 0.4
        yaw = rng.uniform(-np.pi, np.pi)
        root[e, 3:7] = [0, 0, np.sin(yaw / 2), np.cos(yaw / 2)]
        tgt[e] = rng.normal(size=69) * 0.15
    return root, dof, tgt


def bumpy_heightfield(n=1100, seed=0, amp=0.08, slope=0.0):
    """int16 field (0.1 m grid, 0.005 m units): smooth random bumps of +-amp metres over a constant slope along x,
    covering the 50..60 m region the test scenes stand in."""
    rng = np.random.default_rng(seed)
    x = np.arange(n)[:, None] * 0.1
    y = np.arange(n)[None, :] * 0.1
    z = slope * (x - 50.0) + 0 * y
    for _ in range(6):
        kx, ky, ph = rng.uniform(0.5, 4.0), rng.uniform(0.5, 4.0), rng.uniform(0, 6.28)
        z = z + (amp / 3) * np.sin(kx * x + ky * y + ph)
    return dict(samples=np.rint(z / 0.005).astype(np.int16), horizontal_scale=0.1, vertical_scale=0.005)


def oracle_sim(models, root, dof, tgt, self_collision=None, heightfield=None, **params):
    import oracle
    if self_collision is True:
        from emloco_amd.model import pack_self_collision
        self_collision = pack_self_collision(models)
    s = oracle.Sim(pack_models(models), oracle.default_params(**params), self_collision=self_collision, heightfield=heightfield)
    s.root_state[:] = root
    s.dof_state[:] = dof
    s.pd_target[:] = tgt
    return s
