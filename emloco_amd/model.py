"""Humanoid model description: MJCF loading, mass properties, per-env ("AMASS-shaped") variation.

This is the host-side asset layer behind `gym.load_asset` (reference call site
pacer/pacer/env/tasks/humanoid.py:720; the asset it loads is data/assets/mjcf/smpl_humanoid.xml).
The reference delegates MJCF import to Isaac Gym; here the importer is ours:

* bodies in MJCF depth-first order (= rigid-body order the task code relies on, humanoid.py:264);
* one geom per body, mass = density x primitive volume, inertia of the primitive about its centre;
* the three hinge joints of a body form one 3-DoF joint whose coordinates are the rotation vector
  (SURVEY.md section 7 "3-DoF joints as exp-map DoFs"); per-DoF stiffness/damping/armature/range
  come from the hinge attributes.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

GEOM_SPHERE, GEOM_CAPSULE, GEOM_BOX = 0, 1, 2
DEFAULT_EFFORT = 500.0  # MJCF motor gear x ctrlrange of the shipped humanoid (smpl_humanoid.xml:6,171)


@dataclass
class HumanoidModel:
    """One humanoid (numpy, float64 for the derivation; packed to float32 for the device)."""
    names: list
    parent: np.ndarray        # (nb,) int32
    joint_off: np.ndarray     # (nb,3)
    mass: np.ndarray          # (nb,)
    com: np.ndarray           # (nb,3)
    inertia: np.ndarray       # (nb,6) xx yy zz xy xz yz about com, body frame
    geom_type: np.ndarray     # (nb,) int32
    geom_a: np.ndarray        # (nb,3)
    geom_b: np.ndarray        # (nb,3)
    geom_r: np.ndarray        # (nb,)
    kp: np.ndarray            # (ndof,)
    kd: np.ndarray
    armature: np.ndarray
    effort: np.ndarray
    lim_lower: np.ndarray     # (ndof,) rad
    lim_upper: np.ndarray
    density: np.ndarray = field(default=None)

    @property
    def num_bodies(self):
        return len(self.names)

    @property
    def num_dof(self):
        return 3 * (len(self.names) - 1)

    @property
    def dof_names(self):
        return [f"{n}_{a}" for n in self.names[1:] for a in "xyz"]

    def total_mass(self):
        return float(self.mass.sum())

    def depth(self):
        d = np.zeros(self.num_bodies, dtype=np.int32)
        for i in range(1, self.num_bodies):
            d[i] = d[self.parent[i]] + 1
        return d

    def scaled(self, length_scale=1.0, mass_scale=1.0):
        """Uniform limb-length scale and an extra mass scale (synthetic AMASS-shaped variation)."""
        s, ms = float(length_scale), float(mass_scale)
        m = HumanoidModel(**{k: (v.copy() if isinstance(v, np.ndarray) else list(v))
                             for k, v in self.__dict__.items()})
        m.joint_off *= s
        m.com *= s
        m.geom_a *= s
        m.geom_b *= s
        m.geom_r *= s
        m.mass *= s ** 3 * ms
        m.inertia *= s ** 5 * ms
        if m.density is not None:
            m.density = m.density * ms       # keeps a written MJCF consistent with the scaled masses
        return m


def _sphere_props(rho, c, r):
    m = rho * 4.0 / 3.0 * np.pi * r ** 3
    i = 0.4 * m * r * r
    return m, np.asarray(c, float), np.diag([i, i, i])


def _capsule_props(rho, p0, p1, r):
    p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
    L = np.linalg.norm(p1 - p0)
    u = (p1 - p0) / L if L > 0 else np.array([0.0, 0.0, 1.0])
    mc = rho * np.pi * r * r * L
    ms = rho * 4.0 / 3.0 * np.pi * r ** 3
    ia = 0.5 * mc * r * r + 0.4 * ms * r * r
    ip = mc * (L * L / 12.0 + r * r / 4.0) + ms * (0.4 * r * r + L * L / 4.0 + 3.0 * L * r / 8.0)
    uu = np.outer(u, u)
    return mc + ms, 0.5 * (p0 + p1), ip * (np.eye(3) - uu) + ia * uu


def _box_props(rho, c, h):
    h = np.asarray(h, float)
    m = rho * 8.0 * h[0] * h[1] * h[2]
    i = m / 3.0 * np.array([h[1] ** 2 + h[2] ** 2, h[0] ** 2 + h[2] ** 2, h[0] ** 2 + h[1] ** 2])
    return m, np.asarray(c, float), np.diag(i)


def _from_rows(rows):
    nb = len(rows)
    names, parent = [], np.zeros(nb, np.int32)
    off, mass, com, inr = np.zeros((nb, 3)), np.zeros(nb), np.zeros((nb, 3)), np.zeros((nb, 6))
    gt, ga, gb, gr = np.zeros(nb, np.int32), np.zeros((nb, 3)), np.zeros((nb, 3)), np.zeros(nb)
    dens = np.zeros(nb)
    ndof = 3 * (nb - 1)
    kp, kd, arm, lim = np.zeros(ndof), np.zeros(ndof), np.zeros(ndof), np.zeros(ndof)
    for i, (name, par, pos, rho, geom, skp, skd, sarm, rng) in enumerate(rows):
        names.append(name)
        parent[i] = par
        off[i] = pos
        dens[i] = rho
        kind = geom[0]
        if kind == "sphere":
            m, c, I = _sphere_props(rho, geom[1], geom[2])
            gt[i], ga[i], gr[i] = GEOM_SPHERE, geom[1], geom[2]
        elif kind == "capsule":
            m, c, I = _capsule_props(rho, geom[1], geom[2], geom[3])
            gt[i], ga[i], gb[i], gr[i] = GEOM_CAPSULE, geom[1], geom[2], geom[3]
        elif kind == "box":
            m, c, I = _box_props(rho, geom[1], geom[2])
            gt[i], ga[i], gb[i] = GEOM_BOX, geom[1], geom[2]
        else:
            raise ValueError(f"unsupported geom type {kind!r}")
        mass[i], com[i] = m, c
        inr[i] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
        if i > 0:
            sl = slice(3 * (i - 1), 3 * i)
            kp[sl], kd[sl], arm[sl], lim[sl] = skp, skd, sarm, np.deg2rad(rng)
    off[0] = 0.0  # the root pose comes from the actor start pose, not from the MJCF body pos
    return HumanoidModel(names=names, parent=parent, joint_off=off, mass=mass, com=com, inertia=inr,
                         geom_type=gt, geom_a=ga, geom_b=gb, geom_r=gr, kp=kp, kd=kd, armature=arm,
                         effort=np.full(ndof, DEFAULT_EFFORT), lim_lower=-lim, lim_upper=lim, density=dens)


def smpl_humanoid() -> HumanoidModel:
    """The mean-shape SMPL humanoid (24 bodies, 69 DoF, 71.78 kg)."""
    from .assets.smpl_humanoid_table import BODIES
    return _from_rows(BODIES)


def _floats(s):
    return [float(x) for x in s.split()]


def load_mjcf(path) -> HumanoidModel:
    """Parse an MJCF file of the family the reference generates (one geom per body, three hinges per
    non-root body, `freejoint` root).  Angles are degrees (MJCF default)."""
    root = ET.parse(path).getroot()
    wb = root.find("worldbody")
    top = wb.find("body")
    if top is None:
        raise ValueError(f"{path}: no body under <worldbody>")
    rows = []

    def walk(b, par):
        idx = len(rows)
        g = b.find("geom")
        if g is None:
            raise ValueError(f"{path}: body {b.get('name')} has no geom")
        rho = float(g.get("density", "1000"))
        t = g.get("type", "capsule")
        if t == "sphere":
            spec = ("sphere", _floats(g.get("pos", "0 0 0")), float(g.get("size")))
        elif t == "capsule":
            ft = _floats(g.get("fromto"))
            spec = ("capsule", ft[:3], ft[3:], float(g.get("size")))
        elif t == "box":
            spec = ("box", _floats(g.get("pos", "0 0 0")), _floats(g.get("size")))
        else:
            raise ValueError(f"{path}: unsupported geom type {t}")
        js = b.findall("joint")
        if par >= 0 and len(js) != 3:
            raise ValueError(f"{path}: body {b.get('name')} needs three hinge joints, has {len(js)}")
        if js:
            j = js[0]
            gains = (float(j.get("stiffness", 0)), float(j.get("damping", 0)), float(j.get("armature", 0)),
                     float(j.get("range", "-180 180").split()[1]))
        else:
            gains = (0.0, 0.0, 0.0, 0.0)
        rows.append((b.get("name"), par, _floats(b.get("pos", "0 0 0")), rho, spec) + gains)
        for c in b.findall("body"):
            walk(c, idx)

    walk(top, -1)
    return _from_rows(rows)


def write_mjcf(model: HumanoidModel, path):
    """Write the model as an MJCF file (so `load_asset(sim, root, file)` has a file to read)."""
    kids = {i: [] for i in range(model.num_bodies)}
    for i in range(1, model.num_bodies):
        kids[int(model.parent[i])].append(i)
    out = ['<mujoco model="humanoid">', '  <compiler coordinate="local"/>', "  <worldbody>"]

    def f3(v):
        return " ".join(f"{x:.6g}" for x in v)

    def emit(i, ind):
        p = "  " * ind
        out.append(f'{p}<body name="{model.names[i]}" pos="{f3(model.joint_off[i])}">')
        if i == 0:
            out.append(f'{p}  <freejoint name="{model.names[i]}"/>')
        else:
            for k, ax in enumerate(("1 0 0", "0 1 0", "0 0 1")):
                d = 3 * (i - 1) + k
                rng = np.rad2deg(model.lim_upper[d])
                out.append(f'{p}  <joint name="{model.names[i]}_{"xyz"[k]}" type="hinge" pos="0 0 0" axis="{ax}" '
                           f'stiffness="{model.kp[d]:g}" damping="{model.kd[d]:g}" armature="{model.armature[d]:g}" '
                           f'range="{-rng:.4f} {rng:.4f}"/>')
        rho = model.density[i]
        if model.geom_type[i] == GEOM_SPHERE:
            out.append(f'{p}  <geom type="sphere" density="{rho:.10g}" size="{model.geom_r[i]:g}" pos="{f3(model.geom_a[i])}"/>')
        elif model.geom_type[i] == GEOM_CAPSULE:
            out.append(f'{p}  <geom type="capsule" density="{rho:.10g}" fromto="{f3(model.geom_a[i])} {f3(model.geom_b[i])}" size="{model.geom_r[i]:g}"/>')
        else:
            out.append(f'{p}  <geom type="box" density="{rho:.10g}" pos="{f3(model.geom_a[i])}" size="{f3(model.geom_b[i])}"/>')
        for c in kids[i]:
            emit(c, ind + 1)
        out.append(f"{p}</body>")

    emit(0, 2)
    out += ["  </worldbody>", "</mujoco>"]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")


# per-shape collision filter bitmasks of the SMPL humanoid (pacer/pacer/env/tasks/humanoid.py:926, no mesh, no master
# foot): two shapes collide only if (filter_a & filter_b) == 0
SMPL_SHAPE_FILTERS = [0, 0, 7, 16, 12, 0, 56, 2, 33, 128, 0, 192, 0, 64, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]


SC_MAXSEG = 32          # EMLOCO_SC_MAXSEG


def collision_topology(model: HumanoidModel):
    """The SHAPE of a humanoid's collision segments, decided once per simulator from a canonical model (the first env's): per box
    body the axis order (thinnest, middle, longest) and whether the box is split into two capsules.  Every env of the sim then uses
    this topology with its own dimensions (`collision_capsules(m, topology)`): a population of body shapes in which one foot box
    crosses the split threshold, or swaps two nearly equal axes, still shares one segment list and one pair table."""
    topo = {}
    for i in range(model.num_bodies):
        if model.geom_type[i] == GEOM_BOX:
            h = np.abs(model.geom_b[i])
            order = np.argsort(h)                              # thinnest, middle, longest axis
            topo[i] = (tuple(int(x) for x in order), bool(h[order[1]] > 1.5 * float(h[order[0]])))
    return topo


def collision_capsules(model: HumanoidModel, topology=None):
    """Sphere-swept segments of the bodies for limb-limb contact, in the body frame: spheres and capsules as they are; a box as
    the capsule along its longest axis with the smallest half extent as radius -- and, where the box is much WIDER than that
    capsule (middle half extent > 1.5 x the smallest: the SMPL ankle boxes, 17 x 9.7 x 4.2 cm; the toe boxes are as thick as wide
    and keep one capsule), as TWO such capsules along its two long edges, so that the foot collides with its real width.
    `topology` (collision_topology of a canonical model; default: this model's own) fixes which boxes are split and along which
    axes; only end points and radii follow this model's dimensions.
    Returns (a, b, r, seg_body): n_seg = 24 + number of second capsules rows each; seg_body[i] = i for i < 24."""
    nb = model.num_bodies
    topology = collision_topology(model) if topology is None else topology
    a, b, r = [model.geom_a[i].copy() for i in range(nb)], [model.geom_a[i].copy() for i in range(nb)], [float(model.geom_r[i]) for i in range(nb)]
    seg_body = list(range(nb))
    for i in range(nb):
        if model.geom_type[i] == GEOM_CAPSULE:
            b[i] = model.geom_b[i].copy()
        elif model.geom_type[i] == GEOM_BOX:
            h = np.abs(model.geom_b[i])
            order, split = topology[i]
            rad = float(h[order[0]])
            half = np.zeros(3)
            half[order[2]] = max(h[order[2]] - rad, 0.0)
            if split:
                side = np.zeros(3)
                side[order[1]] = max(h[order[1]] - rad, 0.0)       # the two capsules touch the box's long faces from inside
                a[i], b[i], r[i] = model.geom_a[i] - half - side, model.geom_a[i] + half - side, rad
                a.append(model.geom_a[i] - half + side); b.append(model.geom_a[i] + half + side); r.append(rad)
                seg_body.append(i)
            else:
                a[i], b[i], r[i] = model.geom_a[i] - half, model.geom_a[i] + half, rad
    if len(seg_body) > SC_MAXSEG:
        raise ValueError(f"{len(seg_body)} collision segments; the kernels hold {SC_MAXSEG}")
    return np.asarray(a, np.float64), np.asarray(b, np.float64), np.asarray(r, np.float64), np.asarray(seg_body, np.uint8)


def _segment_distance(p0, p1, q0, q1):
    d1, d2, rr = p1 - p0, q1 - q0, p0 - q0
    aa, ee, ff = d1 @ d1, d2 @ d2, d2 @ rr
    if aa <= 1e-12 and ee <= 1e-12:
        s = t = 0.0
    elif aa <= 1e-12:
        s, t = 0.0, np.clip(ff / ee, 0, 1)
    else:
        cc = d1 @ rr
        if ee <= 1e-12:
            t, s = 0.0, np.clip(-cc / aa, 0, 1)
        else:
            bb = d1 @ d2
            den = aa * ee - bb * bb
            s = np.clip((bb * ff - cc * ee) / den, 0, 1) if den > 1e-12 else 0.0
            t = (bb * s + ff) / ee
            if t < 0:
                t, s = 0.0, np.clip(-cc / aa, 0, 1)
            elif t > 1:
                t, s = 1.0, np.clip((bb - cc) / aa, 0, 1)
    return float(np.linalg.norm((p0 + d1 * s) - (q0 + d2 * t)))


def self_collision_pairs(model: HumanoidModel, filters=None, margin=0.01):
    """SEGMENT pairs tested for self-contact (`has_self_collision`, humanoid.py:917-944): segments of two different bodies that are
    not parent / child (articulation links never collide with their parent), whose filter bitmasks are disjoint, and that do not
    already touch in the bind pose (all joint angles zero) -- a penalty contact on a permanent overlap would push the limbs apart
    for ever, so those pairs are filtered the way asset importers do.  Returns uint8 [n][2] with i < j, ascending."""
    filters = SMPL_SHAPE_FILTERS if filters is None else filters
    nb = model.num_bodies
    pw = np.zeros((nb, 3))
    for i in range(1, nb):
        pw[i] = pw[model.parent[i]] + model.joint_off[i]
    a, b, r, sb = collision_capsules(model)
    pairs = []
    for i in range(len(sb)):
        for j in range(i + 1, len(sb)):
            bi, bj = int(sb[i]), int(sb[j])
            if bi == bj or model.parent[bj] == bi or model.parent[bi] == bj or (filters[bi] & filters[bj]) != 0:
                continue
            if _segment_distance(pw[bi] + a[i], pw[bi] + b[i], pw[bj] + a[j], pw[bj] + b[j]) < r[i] + r[j] + margin:
                continue
            pairs.append((i, j))
    return np.asarray(pairs, dtype=np.uint8).reshape(-1, 2)


def pack_self_collision(models, k=1.5e4, c=60.0, max_pen=0.04, filters=None, mu=1.0):
    """Arrays of `EmlocoSelfCollisionDesc` (include/emloco_sim.h): the segment-pair table and segment -> body map of the first model
    (one table per sim: the kernels share it across envs) and per-env collision segments."""
    if any((m.geom_type != models[0].geom_type).any() for m in models):
        raise ValueError("the envs' humanoids must have the same geom types per body (one segment topology per sim)")
    topo = collision_topology(models[0])              # which boxes are split, along which axes: decided ONCE, dimensions per env
    caps = [collision_capsules(m, topo) for m in models]
    seg_body = caps[0][3]
    f32 = lambda idx: np.ascontiguousarray(np.stack([cpl[idx] for cpl in caps]).astype(np.float32))
    return dict(pairs=np.ascontiguousarray(self_collision_pairs(models[0], filters)), cap_a=f32(0), cap_b=f32(1), cap_r=f32(2),
                k=float(k), c=float(c), max_pen=float(max_pen), mu=float(mu), seg_body=np.ascontiguousarray(seg_body))


def pack_models(models):
    """Stack per-env models into the float32 arrays the C ABI takes (`EmlocoModelDesc`, include/emloco_sim.h)."""
    m0 = models[0]
    f32 = lambda name: np.ascontiguousarray(np.stack([getattr(m, name) for m in models]).astype(np.float32))
    return dict(
        parent=np.ascontiguousarray(m0.parent.astype(np.int32)),
        geom_type=np.ascontiguousarray(m0.geom_type.astype(np.int32)),
        joint_off=f32("joint_off"), mass=f32("mass"), com=f32("com"), inertia=f32("inertia"),
        geom_a=f32("geom_a"), geom_b=f32("geom_b"), geom_r=f32("geom_r"),
        kp=f32("kp"), kd=f32("kd"), armature=f32("armature"), effort=f32("effort"),
    )
