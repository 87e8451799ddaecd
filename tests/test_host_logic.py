"""Host-side mirrors of the reference's per-episode logic, pinned against golden vectors (CPU only)."""
import os
import pytest
import numpy as np
import torch

from emloco_amd.utils.flags import Flags


def _flags(**kw):
    base = dict(real_path=False, jta_path=False, jrdb_path=False, pred_path=False, fixed_path=False, slow=False,
                adjust_root_vel=False, init_heading=False, heading_inversion=False, add_noise=False, vru=False)
    base.update(kw)
    return Flags(base)


def _gen(flags, E=16):
    from emloco_amd.env.util.traj_generator import TrajGenerator
    dt = 2 * (1.0 / 60.0)
    return TrajGenerator(E, 168 * dt, 101, "cpu", 2.0, 0.0005, 3.0, 2.0, 0.02, None, hybridInitProb=0.5, flags=flags)


def _draws(g, *extra):
    keys = ["r_dtheta", "r_dtheta_sharp", "bern_sharp", "r_heading", "r_dspeed", "r_speed0"] + list(extra)
    return {k: torch.from_numpy(g[k]) for k in keys}


def test_traj_reset_plain_matches_reference(golden):
    g = golden("traj_reset_plain")
    tg = _gen(_flags())
    ids = torch.arange(16)
    tg.reset(ids, torch.from_numpy(g["init_pos"]), torch.from_numpy(g["root_vel"]), draws=_draws(g))
    np.testing.assert_allclose(tg._verts.numpy(), g["verts"], rtol=1e-6, atol=1e-5)
    assert abs(tg._dt - float(g["dt_vert"])) < 1e-12


def test_traj_reset_same_seed_same_stream(golden):
    """With torch's own CPU generator the mirror consumes the draws in the reference's order."""
    g = golden("traj_reset_plain")
    tg = _gen(_flags())
    torch.manual_seed(int(g["rng_seed"]))
    tg.reset(torch.arange(16), torch.from_numpy(g["init_pos"]), torch.from_numpy(g["root_vel"]))
    np.testing.assert_allclose(tg._verts.numpy(), g["verts"], rtol=1e-6, atol=1e-5)


def test_traj_reset_heading_inversion_matches_reference(golden):
    g = golden("traj_reset_heading")
    tg = _gen(_flags(init_heading=True, heading_inversion=True, adjust_root_vel=True))
    tg.reset(torch.arange(16), torch.from_numpy(g["init_pos"]), torch.from_numpy(g["root_vel"]),
             draws=_draws(g, "r_inversion"))
    np.testing.assert_allclose(tg._verts.numpy(), g["verts"], rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(tg.show_inverted().long().numpy(), g["inverted"])   # bit-exact mask


@pytest.mark.parametrize("name", ["traj_reset_real1", "traj_reset_real2", "traj_reset_real2_noadj"])
def test_traj_reset_real_path_matches_reference(golden, name):
    """flags.real_path (traj_generator.py:120-160; what configs[1] runs): one and two datasets, with and without
    adjust_root_vel, a first segment of zero length included; the reference's random.sample is replayed through `real_rids`."""
    from helpers import TRAJ_CASES
    from emloco_amd.env.util.traj_generator import TrajGenerator
    g = golden(name)
    n_jta = int(g["n_jta"])
    tables = [g["real_table"][:n_jta]] + ([g["real_table"][n_jta:]] if g["real_table"].shape[0] > n_jta else [])
    data = [{i: {"pose": None, "traj": t[i]} for i in range(len(t))} for t in tables]
    dt = 2 * (1.0 / 60.0)
    tg = TrajGenerator(16, 168 * dt, 101, "cpu", 2.0, 0.0005, 3.0, 2.0, 0.02, None, hybridInitProb=0.5,
                       flags=_flags(**TRAJ_CASES[name]), traj_data=data)
    d = _draws(g, "r_real", "r_inversion")
    d["real_rids"] = g["real_rids"].tolist()
    tg.reset(torch.arange(16), torch.from_numpy(g["init_pos"]), torch.from_numpy(g["root_vel"]), draws=d)
    assert int((g["r_real"] > 0.5).sum()) == len(g["real_rids"]) > 3 and len(set(g["real_rids"].tolist())) == len(g["real_rids"])
    np.testing.assert_allclose(tg._verts.numpy(), g["verts"], rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(tg.show_inverted().long().numpy(), g["inverted"])
    np.testing.assert_array_equal(tg.real_rows(), g["real_table"].astype(np.float32))


def test_real_pick_permutation_is_a_bijection():
    """The keyed permutation that stands where the reference calls random.sample: distinct rows within a call (sampling
    without replacement), every row reachable, different keys give different samples."""
    from emloco_amd._lib import real_pick_perm
    for n in (1, 2, 3, 17, 40, 1000, 4096, 8191):
        for key in (0, 1, 0xDEADBEEF):
            out = [real_pick_perm(i, n, key) for i in range(n)]
            assert sorted(out) == list(range(n))
    a = [real_pick_perm(i, 8192, 5) for i in range(2048)]
    b = [real_pick_perm(i, 8192, 6) for i in range(2048)]
    assert a != b and len(set(a) & set(b)) < 800                   # ~ 2048^2 / 8192 = 512 expected in common
    assert abs(np.mean(a) - 4096) < 300 and abs(np.corrcoef(np.arange(2048), a)[0, 1]) < 0.08


def test_calc_pos_matches_reference(golden):
    g = golden("traj_samples")
    tg = _gen(_flags())
    tg._verts[:] = torch.from_numpy(g["verts"])
    times = torch.from_numpy(g["progress"]) * float(g["dt"])
    out = tg.calc_pos(torch.arange(16), times)
    np.testing.assert_allclose(out.numpy(), g["tar_pos"], rtol=1e-6, atol=1e-6)
    assert abs(tg.get_traj_duration() - float(g["traj_dur"])) < 1e-12


def test_jta_dataset_pickles_and_collate(tmp_path):
    """On-disk JTA format (dataset_jta.py:97-103): pickled lists of scenes -> people -> (joints (21,49,4), mask (21,49));
    collate pads scenes to the largest person count and flags padded persons."""
    import torch
    from torch.utils.data import DataLoader
    from emloco_amd.predictor.dataset_jta import collate_batch, create_dataset, get_datasets, write_synthetic_split
    write_synthetic_split(str(tmp_path), "train", 23, max_people=5, seed=1, part_size=10)
    ds = create_dataset("jta_all_visual_cues", split="train", track_size=21, track_cutoff=9, preprocessed=True, root=str(tmp_path))
    assert len(ds) == 23 and len(os.listdir(tmp_path / "jta_all_visual_cues" / "preprocess_smpl" / "train")) == 3
    j, m = ds[0]
    assert j.shape[1:] == (21, 49, 4) and m.shape[1:] == (21, 49)
    joints, masks, pad = next(iter(DataLoader(ds, batch_size=8, collate_fn=collate_batch, shuffle=False)))
    n_people = [ds[i][0].shape[0] for i in range(8)]
    assert joints.shape == (8, max(n_people), 21, 49, 4) and pad.dtype == torch.bool
    for i, n in enumerate(n_people):
        assert not pad[i, :n].any() and pad[i, n:].all() and (joints[i, n:] == 0).all()
    cfg = {"TRAIN": {"input_track_size": 9, "output_track_size": 12}, "DATA": {"preprocessed": True}}
    assert len(get_datasets(["jta_all_visual_cues"], cfg, root=str(tmp_path))[0]) == 23
    with pytest.raises(ValueError):
        create_dataset("nope", split="train", preprocessed=True, root=str(tmp_path))
def test_predictor_config_and_cli_plumbing(tmp_path):
    """The shipped yaml files parse into the dict create_model reads, the command line folds into it as train_jta.py:465-488
    does, and the checkpoint picker follows evaluate_jta.py:537-553 -- host logic only, no GPU."""
    import os
    from emloco_amd.predictor import evaluate_jta as ev
    from emloco_amd.predictor import train_jta as tr
    args = tr.build_arg_parser().parse_args(["--exp_name", "x", "--valueloss_w", "1.0", "--multi_modal", "--out_root", str(tmp_path)])
    cfg = tr.config_from_args(args)
    assert cfg["MODEL"]["num_layers_local"] == 6 and cfg["MODEL"]["num_layers_global"] == 3 and cfg["MODEL"]["num_modes"] == 20
    assert cfg["MODEL"]["seq_len"] == 453 and cfg["MODEL"]["token_num"] == 49 and cfg["TRAIN"]["max_grad_norm"] == 1.0
    assert cfg["USE_VALUELOSS"] and cfg["MULTI_MODAL"] and cfg["TRAIN"]["valuenet_weight"] == 1.0 and cfg["RESUME"] == -1
    assert os.path.exists(os.path.join(cfg["OUTPUT"]["ckpt_dir"], "config.yaml"))
    j = tr.load_config("configs/jrdb_all_visual_cues.yaml", exp_name="y", dataset_name="JRDB", out_root=str(tmp_path))
    assert j["MODEL"]["seq_len"] == 246 and j["MODEL"]["token_num"] == 26 and j["TRAIN"]["batch_size"] == 20
    ea = ev.build_arg_parser().parse_args(["--exp_name", "x", "--out_root", str(tmp_path), "--epoch", "7"])
    d = os.path.join(str(tmp_path), "JTA", "x", "checkpoints")
    open(os.path.join(d, "checkpoint_7epoch.pth.tar"), "w").close()
    assert ev.find_checkpoint(ea).endswith("checkpoint_7epoch.pth.tar")
    open(os.path.join(d, "best_val_checkpoint_7epoch.pth.tar"), "w").close()
    assert ev.find_checkpoint(ea).endswith("best_val_checkpoint_7epoch.pth.tar")


def test_jrdb_dataset_pickles_collate_and_cli(tmp_path):
    """On-disk JRDB format (dataset_jrdb.py:129-210): `preprocess_smpl_filtered_v4/<split>` pickles of scenes -> people ->
    (joints (21,26,4), mask, ids); items carry the scene index, collate returns the index list as a fourth entry; the JRDB
    command line (train_jrdb.py:353-397) folds into the config -- host logic only, no GPU."""
    import torch
    from torch.utils.data import DataLoader
    from emloco_amd.predictor import train_jrdb as tj
    from emloco_amd.predictor.dataset_jrdb import collate_batch, create_dataset, get_datasets, write_synthetic_split
    write_synthetic_split(str(tmp_path), "train", 11, max_people=4, seed=2)
    ds = create_dataset("jrdb_all_visual_cues", split="train", track_size=21, track_cutoff=9, preprocessed=True, root=str(tmp_path))
    assert len(ds) == 11 and os.path.isdir(tmp_path / "jrdb_all_visual_cues" / "preprocess_smpl_filtered_v4" / "train")
    j, m, idx = ds[3]
    assert j.shape[1:] == (21, 26, 4) and m.shape[1:] == (21, 26) and idx == 3 and len(ds.show_meta_info(3)) == j.shape[0]
    joints, masks, pad, idxs = next(iter(DataLoader(ds, batch_size=4, collate_fn=collate_batch, shuffle=False)))
    assert joints.shape[0] == 4 and joints.shape[2:] == (21, 26, 4) and pad.dtype == torch.bool and idxs == [0, 1, 2, 3]
    cfg = {"TRAIN": {"input_track_size": 9, "output_track_size": 12}, "DATA": {"preprocessed": True}}
    assert len(get_datasets(["jrdb_all_visual_cues"], cfg, root=str(tmp_path))[0]) == 11
    with pytest.raises(NotImplementedError):
        create_dataset("jta_all_visual_cues", split="train", preprocessed=True, root=str(tmp_path))
    a = tj.build_arg_parser().parse_args(["--exp_name", "j", "--valueloss_w", "0.5", "--valueloss_only", "--noisy_traj", "--out_root", str(tmp_path)])
    c = tj.config_from_args(a)
    assert c["MODEL"]["seq_len"] == 246 and c["USE_VALUELOSS"] and c["VAL_LOSS_ONLY"] and c["NOISY_TRAJ"] is True
    assert c["TRAIN"]["valuenet_weight"] == 0.5 and os.path.exists(os.path.join(c["OUTPUT"]["ckpt_dir"], "config.yaml"))
    assert c["OUTPUT"]["ckpt_dir"].endswith(os.path.join("JRDB", "j", "checkpoints"))
    with pytest.raises(NotImplementedError):
        tj.config_from_args(tj.build_arg_parser().parse_args(["--use_hypara_best", "--out_root", str(tmp_path)]))
    # the JRDB loop's LocoVal inputs: normalised tokens 2:26 of the last observed frame, no z flip (train_jrdb.py:186)
    tr = tj.JrdbTrainer.__new__(tj.JrdbTrainer)
    tr.config = {"DEVICE": "cpu"}
    inj = torch.randn(3, 9, 2 * 26, 4)
    pose, vel = tr.primary_state(None, inj)
    assert torch.equal(pose, inj[:, 8, 2:26, :3]) and torch.allclose(vel, (inj[:, 8, 0, :2] - inj[:, 7, 0, :2]) * 2.5)
    assert tj.JrdbTrainer.value_loss_with_multi_modal is False


def test_package_import_leaves_the_environment_alone_and_entry_points_opt_in():
    """emloco_amd/__init__.py (round 6): importing the package never touches GPU_MAX_HW_QUEUES -- it records the caller's value (else the
    runtime default, 4) -- and `configure_runtime()`, which the entry points call ahead of their first GPU call, raises it to 16 unless the
    caller has chosen a value (the HIP runtime reads the variable once, when it initialises)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "EMLOCO_KEEP_HW_QUEUES")}
    env["PYTHONPATH"] = root
    run = lambda code, e: subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True).stdout.split()
    code = "import os, emloco_amd; print(os.environ.get('GPU_MAX_HW_QUEUES'), emloco_amd.hw_queues())"
    assert run(code, env) == ["None", "4"]                                   # plain import: nothing set, the default recorded
    assert run(code, dict(env, GPU_MAX_HW_QUEUES="8")) == ["8", "8"]          # the caller's choice is recorded
    code2 = "import os, emloco_amd; print(emloco_amd.configure_runtime(), os.environ.get('GPU_MAX_HW_QUEUES'), emloco_amd.hw_queues())"
    assert run(code2, env) == ["16", "16", "16"]                             # an entry point opts in
    assert run(code2, dict(env, GPU_MAX_HW_QUEUES="4")) == ["4", "4", "4"]   # ... and never overrides an exported value
    # the package records what the runtime will be initialised with: schedules key off that, not off the environment of the moment
    code3 = "import os, emloco_amd; emloco_amd.configure_runtime(); os.environ['GPU_MAX_HW_QUEUES'] = '2'; print(emloco_amd.hw_queues())"
    assert run(code3, env) == ["16"]
    # the benchmark and the smoke entry opt in
    for f in ("bench.py", "__graft_entry__.py", os.path.join("emloco_amd", "run.py")):
        assert "configure_runtime()" in open(os.path.join(root, f)).read(), f


def test_bench_summary_is_a_compact_digest_of_every_leg():
    """bench.summary_of: the numbers of every leg, no notes -- printed as the LAST key of the line (configs[2] was outside
    the tail of stdout the driver keeps)."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    src = open(os.path.join(root, "bench.py")).read()
    assert src.index('out["summary"] = summary_of(out)') < src.index("print(json.dumps(out))")
    ns = {}
    start = src.index("def summary_of(out):")
    exec(src[start:src.index("def main():")], ns)
    out = {"value": 9.3e6, "ms_per_step": 0.44, "n_gpus": 1, "roofline": {"kernel_ms": 0.359, "frac": 0.0133, "note": "x" * 900},
           "env_step_only": {"value": 9.35e6}, "policy": {"value": 5.9e6, "policy_ms": 0.27, "with_discriminator_and_locoval_fit": {"value": 4.78e6, "ms_per_step": 0.856},
                                                          "ppo": {"fps_total": 96e3, "fps_step": 2e6, "update_ms_per_optimizer_step": 3.4}},
           "jta": {"value": 1971.0, "ms_per_step": 129.9, "steps": 10, "roofline": {"frac": 0.37}, "bf16": {"value": 3396.0, "ms_per_step": 75.4, "roofline": {"frac": 0.08}},
                   "eval": {"value": 10735.0}}}
    sm = ns["summary_of"](out)
    assert sm["configs2_policy_disc_locoval_fit"] == 4.78e6 and sm["ppo_update_ms_per_optimizer_step"] == 3.4 and sm["jta_steps_timed"] == 10
    assert sm["headline_env_steps_per_s"] == 9.3e6 and "cpu_baseline_env_steps_per_s" not in sm          # absent legs are left out
    assert len(json.dumps(sm)) < 1000


def test_flat_grad_bucket_release_and_gather_on_host_tensors():
    """Round 6: `FlatGradBucket.release()` detaches every .grad ahead of a backward pass (autograd then KEEPS the incoming gradients
    instead of adding each into its slice) and `gather()` copies them into the bucket and re-aliases every .grad -- same bucket contents
    as the accumulate-into-views way, a parameter without a gradient keeps its zeros.  (Host tensors: the path of the multi-process CPU
    tests; on the GPU one `emloco_gather_flat` launch per 96 tensors does the copies, tests/test_gpu_predictor.py.)"""
    import torch
    from emloco_amd.dist import FlatGradBucket
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2))]
    x = torch.randn(5)

    def loss_of():
        return (ps[0].t() @ x).sum() * 2.0 + (ps[1] ** 2).sum()          # ps[2] gets no gradient

    b = FlatGradBucket(ps, align=4)
    b.zero()
    loss_of().backward()
    want = b.flat.clone()
    b.zero()
    b.release()
    assert all(p.grad is None for p in ps)
    loss_of().backward()
    assert ps[2].grad is None and ps[0].grad.data_ptr() != b.flat.data_ptr()
    b.gather()
    assert torch.equal(b.flat, want) and want.abs().sum() > 0
    for p, o in zip(b.params, b.offsets):
        assert p.grad.data_ptr() == b.flat.data_ptr() + 4 * o and p.grad.shape == p.shape
    assert torch.equal(ps[2].grad, torch.zeros(2, 2))
