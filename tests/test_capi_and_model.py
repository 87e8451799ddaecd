"""CPU-only: the C-ABI library loads and exports every symbol include/*.h declares; the MJCF importer reproduces
the humanoid's known answers (SURVEY.md section 8a cheat-sheet)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(emloco_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from emloco_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `python -m emloco_amd.build` (hipcc cross-compiles without a GPU)"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for header in ("emloco_sim.h", "emloco_task.h", "emloco_predictor.h"):
        names = _declared(header)
        assert len(names) >= 5
        for n in names:
            assert hasattr(lib, n), f"{n} declared in include/{header} but not exported"
    assert set(_lib.SYMBOLS_SIM) <= set(_declared("emloco_sim.h"))
    assert set(_lib.SYMBOLS_TASK) <= set(_declared("emloco_task.h"))
    # ... and the other way round: nothing is exported behind the headers' back (round-4 review: four diagnostic exports were)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("emloco_")}
    declared = set(_declared("emloco_sim.h")) | set(_declared("emloco_task.h")) | set(_declared("emloco_predictor.h"))
    assert exported <= declared, f"exported but not declared in include/*.h: {sorted(exported - declared)}"


def test_product_path_fails_loudly_without_a_gpu():
    import torch
    from emloco_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.EmlocoError):
        _lib.require_device()
    from emloco_amd.gym import gymapi
    assert gymapi.acquire_gym().create_sim(0, -1, gymapi.SIM_PHYSX, gymapi.SimParams()) is None   # reference convention: None on failure


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "emloco_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "oracle/" not in src.replace("oracle/ ", ""), f"{f} references the oracle"


def test_smpl_humanoid_known_answers():
    from emloco_amd.model import smpl_humanoid
    m = smpl_humanoid()
    assert m.num_bodies == 24 and m.num_dof == 69
    assert m.names[:5] == ['Pelvis', 'L_Hip', 'L_Knee', 'L_Ankle', 'L_Toe'] and m.names[13] == 'Head'
    assert list(m.parent) == [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 11, 14, 15, 16, 17, 11, 19, 20, 21, 22]
    assert int(m.depth().max()) == 8
    assert abs(m.total_mass() - 71.78) < 0.01
    assert abs(m.mass[0] - 16.21) < 0.01 and abs(m.mass[11] - 9.60) < 0.01 and abs(m.mass[13] - 4.33) < 0.01
    assert m.kp[0] == 800 and m.kd[0] == 80 and m.kp[3 * 8] == 1000 and m.kp[3 * 17] == 300 and np.all(m.armature == 0.02)
    assert np.allclose(m.lim_upper[:3], np.pi) and np.allclose(m.lim_upper[3 * 14:3 * 14 + 3], 4 * np.pi)   # shoulders +-720 deg


def test_mjcf_round_trip(tmp_path):
    from emloco_amd.model import load_mjcf, smpl_humanoid, write_mjcf
    m = smpl_humanoid().scaled(1.07, 1.2)
    path = tmp_path / "mjcf" / "h.xml"
    write_mjcf(m, str(path))
    r = load_mjcf(str(path))
    assert r.names == m.names and np.array_equal(r.parent, m.parent)
    np.testing.assert_allclose(r.joint_off[1:], m.joint_off[1:], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r.mass, m.mass, rtol=1e-4)
    np.testing.assert_allclose(r.inertia, m.inertia, rtol=2e-4, atol=1e-9)
    np.testing.assert_allclose(r.kp, m.kp)


def test_motion_lib_frame_blend_matches_reference(golden):
    import torch
    from emloco_amd.utils.motion_lib_synthetic import MotionLibSynthetic
    g = golden("frame_blend")
    ml = MotionLibSynthetic.__new__(MotionLibSynthetic)
    i0, i1, bl = ml._calc_frame_blend(torch.from_numpy(g["time"]), torch.from_numpy(g["length"]),
                                      torch.from_numpy(g["num_frames"]), torch.from_numpy(g["dt"]))
    np.testing.assert_array_equal(i0.numpy(), g["idx0"])       # bit-exact frame indices
    np.testing.assert_array_equal(i1.numpy(), g["idx1"])
    np.testing.assert_allclose(bl.numpy(), g["blend"], rtol=1e-6, atol=1e-6)


def test_host_torch_utils_match_reference(golden):
    import torch
    from emloco_amd.gym import torch_utils as tu
    g = golden("quat_utils")
    q, q2, v = (torch.from_numpy(g[k]) for k in ("q", "q2", "v"))
    tol = dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tu.quat_mul(q, q2).numpy(), g["quat_mul"], **tol)
    np.testing.assert_allclose(tu.quat_apply(q, v).numpy(), g["quat_apply"], **tol)
    np.testing.assert_allclose(tu.quat_rotate(q, v).numpy(), g["my_quat_rotate"], **tol)
    np.testing.assert_allclose(tu.calc_heading(q).numpy(), g["calc_heading"], **tol)
    np.testing.assert_allclose(tu.calc_heading_quat_inv(q).numpy(), g["calc_heading_quat_inv"], **tol)
    np.testing.assert_allclose(tu.exp_map_to_quat(torch.from_numpy(g["exp_map"])).numpy(), g["exp_map_to_quat"], **tol)
    np.testing.assert_allclose(tu.quat_to_exp_map(q).numpy(), g["quat_to_exp_map"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(tu.slerp(q, q2, torch.from_numpy(g["t"])).numpy(), g["slerp"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tu.quat_apply_yaw(q.clone(), v).numpy(), g["quat_apply_yaw"], **tol)


def test_collision_topology_is_decided_once_for_all_envs():
    """model.pack_self_collision: which boxes are split in two capsules, and along which axes, comes from the FIRST env's model; an
    env whose foot box falls on the other side of the split threshold (or whose two smaller half extents swap order) shares that
    topology with its own dimensions -- before round 5 such a population made sim creation raise."""
    import copy
    from emloco_amd.model import GEOM_BOX, collision_capsules, collision_topology, pack_self_collision, smpl_humanoid
    base = smpl_humanoid()
    ankle = [i for i in range(base.num_bodies) if base.geom_type[i] == GEOM_BOX and "Ankle" in base.names[i]][0]
    odd = copy.deepcopy(base)
    h = np.abs(odd.geom_b[ankle]).copy()
    order = np.argsort(h)
    h[order[1]] = 1.2 * h[order[0]]                      # middle half extent below 1.5 x the thinnest: on its own this box is ONE capsule
    odd.geom_b[ankle] = h * np.sign(np.where(odd.geom_b[ankle] == 0, 1.0, odd.geom_b[ankle]))
    assert len(collision_capsules(odd)[3]) == len(collision_capsules(base)[3]) - 1
    sc = pack_self_collision([base, odd, base])
    n_seg = len(sc["seg_body"])
    assert sc["cap_a"].shape == (3, n_seg, 3) and n_seg == len(collision_capsules(base)[3])
    topo = collision_topology(base)
    a, b, r, sb = collision_capsules(odd, topo)
    assert list(sb) == list(sc["seg_body"])
    second = [i for i in range(24, n_seg) if sb[i] == ankle][0]
    # the two capsules of the narrow box still straddle its centre line along the canonical middle axis, by its OWN half width
    mid = topo[ankle][0][1]
    assert abs((a[second][mid] - a[ankle][mid]) - 2 * (h[order[1]] - h[order[0]])) < 1e-12
    assert np.allclose(sc["cap_a"][0], sc["cap_a"][2]) and not np.allclose(sc["cap_a"][0], sc["cap_a"][1])
