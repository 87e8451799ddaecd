"""Host logic of the vectorised evaluation / LocoVal filter (row B10) on CPU: the accumulator is fed by the stock-torch
restatements of the predictor and LocoVal (oracle/predictor_torch.py, test infrastructure) and must reproduce the numbers
the reference's evaluate_ade_fde logged (tests/golden/gen_golden_eval.py) -- in one process and as a 2-rank gloo
data-parallel evaluation (each rank one batch, only scalars / histogram counts exchanged)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("ade", "fde", "min_ade", "min_fde", "worst_ade", "worst_fde", "iye", "ade_value", "fde_value", "ade_random", "fde_random",
        "minade_value", "minfde_value", "ade_rejected", "fde_rejected", "chi_velocity", "chi_acceleration", "chi_ang_velocity",
        "chi_ang_acceleration")


def _models():
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle.predictor_torch import LocoValOracle, TransMotionJTAOracle
    gm = np.load(os.path.join(HERE, "golden", "predictor_multi.npz"))
    model = TransMotionJTAOracle(nhid=32, nhead=4, dim_feedfwd=64, nlayers_local=2, nlayers_global=1, nmode=4, num_tokens=49,
                                 multi_modal=True)
    model.load_state_dict({k[4:].replace("__", "."): torch.from_numpy(gm[k]) for k in gm.files if k.startswith("sd__")}, strict=True)
    vnet = LocoValOracle()
    vnet.load_state_dict({k[4:].replace("__", "."): torch.from_numpy(gm[k]) for k in gm.files if k.startswith("vn__")}, strict=True)
    return model.eval(), vnet.eval()


def _run(batch_ids, thr, distributed):
    from emloco_amd.predictor.evaluate_jta import EvalAccumulator
    from emloco_amd.predictor.train_jta import batch_process_coords
    g = np.load(os.path.join(HERE, "golden", "eval_filter.npz"))
    model, vnet = _models()
    cfg = {"DEVICE": "cpu", "TRAIN": {"input_track_size": 9, "output_track_size": 12}}
    acc = EvalAccumulator(thr)
    sizes = []
    i = 0
    while f"batch{i}.joints" in g.files:
        sizes.append(g[f"batch{i}.joints"].shape[0])
        i += 1
    offs = np.concatenate([[0], np.cumsum(sizes)])
    for b in batch_ids:
        joints = torch.from_numpy(g[f"batch{b}.joints"])
        masks, pm = torch.from_numpy(g[f"batch{b}.masks"]), torch.from_numpy(g[f"batch{b}.padding_mask"])
        pose = joints[:, 0, 8, 3:27, :3].clone()
        ij, _, oj, _, pmf = batch_process_coords(joints, masks, pm, cfg, "traj+all")
        with torch.no_grad():
            pred = model(ij, pmf)[:, -12:]
        acc.update(ij, oj, pred, pose, vnet, torch.from_numpy(g["random_ids"][offs[b]:offs[b + 1]]), "jta")
    return acc.summary(distributed=distributed), g


def _check(res, g, thr):
    tag = f"thr{int(thr * 100)}"
    assert res["samples"] == int(g[f"{tag}.samples"])
    for key in KEYS:
        ref = float(g[f"{tag}.{key}"])
        assert abs(res[key] - ref) <= 2e-4 * max(1.0, abs(ref)), f"{key}: {res[key]:.6f} vs reference log {ref:.5f}"
    np.testing.assert_allclose(res["des"], g[f"{tag}.des"], atol=2e-4)
    assert int(res["value_hist"].sum()) == res["samples"] * 4


@pytest.mark.parametrize("thr", [0.5, 0.52])
def test_eval_accumulator_matches_reference_logs_single_process(thr):
    res, g = _run([0, 1], thr, distributed=False)
    _check(res, g, thr)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    res, g = _run([rank], 0.5, distributed=True)
    _check(res, g, 0.5)                                   # every rank holds the whole-evaluation result
    q.put((rank, float(res["ade_value"]), float(res["chi_velocity"])))
    dist.barrier()
    dist.destroy_process_group()


def test_eval_accumulator_two_rank_gloo_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:] == res[1][1:]
