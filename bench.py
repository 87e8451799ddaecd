#!/usr/bin/env python3
"""bench.py -- env-steps/s of the EmLoco rollout hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one iteration of the rollout loop AMPValueAgent.play_steps drives (amp_continuous_value.py:45-145): reset the
envs that finished, pre-physics -> 4 fused physics substeps -> fused post-physics, then the LocoVal bookkeeping (discounted
returns, 144-step cut-off) and the LocoVal fit on the episodes that ended, whose gradient bucket (6 174 floats + loss +
count) is the one all-reduce of the step (RCCL at N > 1; the identical code runs at N = 1 with the collective a no-op).
Workload at N=1 = BASELINE.json configs[1]'s environment: 4096 SMPL humanoids, random_heading, JTA+JRDB-shaped real paths
(synthetic: the datasets do not ship), flat terrain, self-collision on.  Policy inference (A19) is not part of env.step and
is excluded from `value` (reported beside it): actions are pre-sampled N(0, e^-2.9) like the frozen policy's noise.
For N>1 envs are sharded (4096 per rank, weak scaling), seeds are base + rank.  An untimed pre-roll staggers the episode
ages so that any --steps window sees the steady reset rate.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (sim_step_kernel): algorithmic bytes per launch
 over its HIP-event duration, against the 8 TB/s HBM peak -- the kernel is latency / issue bound, so the
fraction is tiny by construction.  `cpu_baseline` times the CPU oracle (this repo's C restatement, OpenMP over envs; the
reference's own PhysX path does not exist here) on a bounded sample of the same workload on all host cores.
The JTA train-step and JRDB evaluation legs run data-parallel at N > 1 (256 / 512 samples per rank).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import emloco_amd  # noqa: E402
emloco_amd.configure_runtime()   # ahead of the first GPU call: 16 hardware queues for the side streams / graph arms (emloco_amd/__init__.py)

SIM_BYTES_PER_ENV = 9296          # state in/out + per-env model (+ collision capsules) + warm-start, per launch
PROFILE_ROUND = "r06"             # profiles/<round>_*: the committed rocprofv3 summaries of this round's kernels
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16), no sparsity
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)


def synthetic_real_paths(n, seed=0):
    """JTA/JRDB-shaped paths (load_jta_traj.py:72-114 format): 13 waypoints @0.4 s of a constant-turn-rate
    walker, natural cubic spline to 101 vertices, z = 0."""
    from scipy.interpolate import CubicSpline
    rng = np.random.default_rng(seed)
    out = {}
    t = np.arange(13) * 0.4
    td = np.linspace(0, t[-1], 101)
    for i in range(n):
        v, w, h0 = rng.uniform(0.3, 2.0), rng.uniform(-0.35, 0.35), rng.uniform(-np.pi, np.pi)
        hd = h0 + w * t
        xy = np.cumsum(np.stack([np.cos(hd), np.sin(hd)], 1) * v * 0.4, 0)
        xy = np.concatenate([[[0.0, 0.0]], xy[:-1]], 0)
        cs = CubicSpline(t, xy, bc_type="natural")
        traj = np.zeros((101, 3), np.float32)
        traj[:, :2] = cs(td)
        out[i] = {"traj": traj, "pose": None}
    return out


def make_env(num_envs, rank):
    from emloco_amd.run import create_rlgpu_env, fill_flags
    from emloco_amd.utils.config import get_args, load_cfg
    args = get_args(["--num_envs", str(num_envs), "--seed", "0", "--random_heading", "--init_heading",
                     "--heading_inversion", "--adjust_root_vel", "--real_path", "JTA+JRDB",
                     "--sim_device", f"cuda:{rank_local()}", "--rl_device", f"cuda:{rank_local()}"])
    cfg, cfg_train, _ = load_cfg(args)
    n_paths = max(2000, int(num_envs))     # the host reset draws real paths without replacement (traj_generator.py:109): pool >= envs
    cfg["env"]["traj_data"] = [synthetic_real_paths(n_paths, seed=1), synthetic_real_paths(n_paths, seed=2)]
    fill_flags(args)
    return create_rlgpu_env(args, cfg, cfg_train, rank=rank)


def stagger_episodes(env, steps=168, seed=0, sigma=float(np.exp(-2.9))):
    """Untimed pre-roll: every env is force-reset once at a step drawn uniformly from [0, steps), while the rollout runs.
    After `steps` = one episode length the episode ages are spread uniformly over [0, 168), i.e. the timed window sees the
    steady reset rate (~E/168 time-outs per step plus the early terminations) from its first step, whatever --steps is.
    (Writing progress_buf directly would not do: the target runs along the path with progress, every env would terminate
    on the 4 m distance test of humanoid_pedestrain_terrain.py:1468-1530.)"""
    import torch
    task = env.task
    dev = task.device
    E = task.num_envs
    g = torch.Generator(device=dev)
    g.manual_seed(977 + seed)
    phase = torch.randint(0, steps, (E,), device=dev, generator=g)
    pool = torch.randn(8, E, 69, device=dev, generator=g) * sigma
    for k in range(steps):
        task.reset_buf[phase == k] = 1
        env.reset_done()
        env.step(pool[k % 8])


def rank_local():
    # EMLOCO_BENCH_SHARE_GPU=1 (testing only): every rank uses cuda:0 and gloo instead of RCCL, so the multi-process path
    # (launch, sharding, barriers, max-over-ranks timing, aggregation) can be exercised on a 1-GPU box.  The line says so.
    if os.environ.get("EMLOCO_BENCH_SHARE_GPU") == "1":
        return 0
    return int(os.environ.get("LOCAL_RANK", "0"))


def cpu_baseline(sample_envs=4096, sample_steps=100, budget_s=22.0):
    """The CPU oracle (C restatement, OpenMP over envs in the physics step AND in every observation / reward / reset function) on
    the same workload size -- 4096 envs, physics + post-physics maths.  The thread count is swept over {8, 16, 32, 64, all CPUs this
    process may run on (os.sched_getaffinity)} with 2 steps each and the best one runs for `budget_s` seconds or `sample_steps`
    steps: the boxes of the pool expose 256 CPUs in the affinity mask but are shared, and a team larger than the cores that are
    actually free loses to OpenMP's barriers (the sweep is in `sample`)."""
    import oracle
    from emloco_amd.model import pack_models, pack_self_collision
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import varied_models
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    models = varied_models(64, seed=0)
    models = [models[i % 64] for i in range(sample_envs)]
    s = oracle.Sim(pack_models(models), oracle.default_params(n_sub=4), self_collision=pack_self_collision(models))   # as the GPU run
    s.root_state[:, 2] = 0.93
    rng = np.random.default_rng(0)
    betas = rng.normal(size=(sample_envs, 17)).astype(np.float32)
    verts = np.cumsum(rng.normal(size=(sample_envs, 101, 3)).astype(np.float32) * 0.05, 1)
    hf = np.zeros((1200, 1200), np.int16)
    prog = np.zeros(sample_envs, np.int64)
    subset = np.concatenate([np.arange(3 * j, 3 * j + 3) for j in range(23) if j not in (3, 7, 17, 22)]).astype(np.int32)
    targets = (rng.normal(size=(8, sample_envs, 69)) * 0.055 * np.pi).astype(np.float32)
    k_step = [0]

    def one():
        k = k_step[0]
        k_step[0] += 1
        s.pd_target[:] = targets[k % 8]
        s.step(1)
        prog[:] += 1
        rb = s.rb_state
        a = (rb[:, :, 0:3], rb[:, :, 3:7], rb[:, :, 7:10], rb[:, :, 10:13], betas)
        oracle.self_obs(*a)
        oracle.self_obs(*a, flip=True)
        smp = oracle.traj_samples(verts, prog, 1 / 30., 5.656)
        loc = oracle.location_obs(s.root_state, smp)
        head = np.concatenate([rb[:, 13, 0:3], rb[:, 13, 3:7]], -1)
        ho = oracle.height_obs(oracle.get_center_heights(s.root_state, hf), oracle.get_heights(head, hf))
        oracle.flip_task_obs(np.concatenate([loc, ho], 1))
        oracle.reward(rb[:, 0, :3], smp[:, 0], s.dof_force, s.dof_state[:, :, 1])
        oracle.reset(prog, s.contact_force, rb[:, :, :3], smp[:, 0])
        oracle.amp_obs(rb[:, 0, 0:3], rb[:, 0, 3:7], rb[:, 0, 7:10], rb[:, 0, 10:13], s.dof_state[:, :, 0],
                       s.dof_state[:, :, 1], rb[:, [7, 3, 22, 17], 0:3], betas, subset)
        prog[prog >= 167] = 0

    sweep = {}
    for th in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu), ncpu}):
        oracle.set_threads(th)
        one()
        t0 = time.perf_counter()
        one(); one()
        sweep[th] = (time.perf_counter() - t0) / 2
    best = min(sweep, key=sweep.get)
    cores = oracle.set_threads(best)
    t0 = time.perf_counter()
    done = 0
    while done < sample_steps and time.perf_counter() - t0 < budget_s:
        one()
        done += 1
    dt = time.perf_counter() - t0
    return {"value": round(sample_envs * done / dt, 1), "unit": "env-steps/s", "cores": cores, "cpus_in_affinity_mask": ncpu, "kind": "port",
            "sample": f"{sample_envs} envs x {done} steps of the same env.step maths (oracle/ C restatement, OpenMP over envs; thread sweep "
                      + ", ".join(f"{k}: {sample_envs / v:,.0f}/s" for k, v in sweep.items())
                      + f"; {ncpu} CPUs in this process's affinity mask; best = {cores} threads, {dt:.1f} s)"}


def synthetic_jta_batch(B, seed=0, max_people=8, nan_frac=0.02):
    """JTA-shaped batch: joints (B, N, 21, 49, 4), N ~ U{1..8} padded to the batch max, 2.5 fps walker
    trajectories, 24 SMPL-like 3-D joints around the pelvis, boxes / 2-D pose ~ N(0,1), 2 % NaN rows."""
    import torch
    g = torch.Generator().manual_seed(seed)
    n_people = torch.randint(1, max_people + 1, (B,), generator=g)
    N = int(n_people.max())
    joints = torch.randn(B, N, 21, 49, 4, generator=g)
    speed = torch.rand(B, N, 1, 1, generator=g) * 1.5 + 0.3
    heading = (torch.rand(B, N, 1, generator=g) * 2 - 1) * 3.14159 + torch.cumsum(torch.randn(B, N, 21, generator=g) * 0.05, dim=2)
    step = torch.stack([torch.cos(heading), torch.sin(heading)], -1) * speed * 0.4
    joints[:, :, :, 0, :2] = torch.cumsum(step, dim=2) + torch.randn(B, N, 1, 2, generator=g) * 3
    joints[:, :, :, 0, 2:] = 0
    joints[:, :, :, 3:27, :3] = joints[:, :, :, 0:1, :3] + torch.randn(B, N, 21, 24, 3, generator=g) * 0.3
    pad = torch.arange(N).unsqueeze(0) >= n_people.unsqueeze(1)
    # 2 % of the scenes carry a NaN in the primary person's 3-D pose at t = 8 or in its trajectory at t = 7 (a missing
    # annotation): the rows train_jta.py's nan_handler (:143-165) drops from the EmLoco loss
    bad = torch.rand(B, generator=g) < nan_frac
    which = torch.rand(B, generator=g) < 0.5
    joints[bad & which, 0, 8, 5, 1] = float("nan")
    joints[bad & ~which, 0, 7, 0, 0] = float("nan")
    return joints, torch.ones(B, N, 21, 49), pad


def _timed(fn, steps, warmup, world, dev):
    """warm-up, then `steps` calls bracketed by barrier + synchronize; MAX over ranks of the elapsed time."""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        fn()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        from emloco_amd.dist import all_reduce_
        all_reduce_(t, dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def gemm_peak(ops):
    """(peak TFLOP/s, note) of the matrix path the fp32-class GEMMs run on in the current precision mode."""
    if ops.get_matmul_precision() == "fp32_split":
        return round(MFMA_BF16_PEAK_TFLOPS / 6.0, 1), ("fp32-class products from three bf16 pieces per operand: six v_mfma_f32_32x32x16_bf16 per 16 k, fp32 "
                                                       "accumulation, error against float64 at or below the fp32 matrix instruction's (profiles/r03_gemm_split.txt); "
                                                       "peak = dense bf16 matrix peak / 6 piece products; the fp32 matrix instruction's own peak is 157.3.  Round 6: "
                                                       "the FORWARD runs on three pieces (logits 5e-7 of the reference's); the backward's gradient products (attention "
                                                       "dQ / dK / dV, dx, dW) on TWO pieces = three piece products (gradients 4-9e-6 of a tensor's scale next to the loss, "
                                                       "bar 2e-4: profiles/r06_ab_bwd_pieces.txt; EMLOCO_BWD_PIECES=3 EMLOCO_ATTN_BWD_PIECES=3 restore three) -- the "
                                                       "peak quoted here stays the six-product one, so the backward launches count against a ceiling they do not have")
    return 157.3, "fp32-in fp32-accumulate MFMA (v_mfma_f32_32x32x2_f32), peak = fp32 matrix peak"


def jta_leg(dev, steps=10, warmup=2, B=256, rank=0, world=1):
    """train_jta.py EmLoco step (configs[3]): fwd + MSE + LocoVal loss + bwd + clip + Adam, batch 256 per GPU, fp32 MFMA.
    world > 1: data parallel (EmLocoTrainer(data_parallel=True): one flat 3.2 M-float gradient all-reduce per step)."""
    import torch
    from emloco_amd.learning.value_pose_net import ValuePoseNet
    from emloco_amd.predictor import ops
    from emloco_amd.predictor.model_jta import TransMotionJTA
    from emloco_amd.predictor.train_jta import EmLocoTrainer
    torch.manual_seed(0)
    cfg = {"DEVICE": str(dev), "MULTI_MODAL": False, "USE_FRAME_MASK": False,
           "TRAIN": {"input_track_size": 9, "output_track_size": 12, "lr": 1e-4, "max_grad_norm": 1.0, "valuenet_weight": 1.0}}
    model = TransMotionJTA(tok_dim=453, nhid=128, nhead=4, dim_feedfwd=1024, nlayers_local=6, nlayers_global=3, nmode=20,
                           output_scale=1, obs_and_pred=21, num_tokens=49, device=str(dev)).to(dev)
    trainer = EmLocoTrainer(model, ValuePoseNet(True, True).to(dev), cfg, data_parallel=world > 1)
    joints, masks, pad = synthetic_jta_batch(B, seed=rank)
    _timed(lambda: trainer.step(joints, masks, pad), 0, warmup, world, dev)
    ops.gemm_timing(True)
    dt = _timed(lambda: trainer.step(joints, masks, pad), steps, 0, world, dev)
    n, ms, fl = ops.gemm_timing()
    gemm_bytes = ops.gemm_timing_bytes()
    ops.gemm_timing(False)
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    gbs = gemm_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    peak, peak_note = gemm_peak(ops)
    traffic = None            # HBM-side bytes of the step's GEMM launches from the committed PMC passes (pointer, not a measurement of this run)
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_jta_gemm_hbm_bytes.json"))).get("gemm_bytes_per_step")
    except Exception:
        pass
    out = {"metric": "JTA samples/sec (train_jta EmLoco step)", "value": round(B * world * steps / dt, 2), "unit": "samples/s", "n_gpus": world,
           "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps, "warmup": warmup, "dtype": "f32",
           "config": {"workload": "configs[3]: Social-Transmotion train_jta.py with EmLoco loss (valueloss_w=1.0), batch 256, "
                                  "9-in/12-out frames, people/scene U{1..8} padded, 2 % NaN rows, d=128 h=4 ff=1024, 6+3 layers"
                                  + (f", data-parallel x{world} (256 per GPU)" if world > 1 else ""),
                      "batch": B, "people_padded": int(joints.shape[1]), "tokens_per_person": 453},
           "precision": ops.get_matmul_precision(),
           "backward_pieces": {"attention": int(os.environ.get("EMLOCO_ATTN_BWD_PIECES", "2")), **ops.backward_pieces()},
           # The step's GEMM launches as a class (45 % of its kernel time; the fused attention, VALU / matrix bound, is the other half and
           # is reported under `attention`).  Their reductions are short (K = 128 for five of the seven products of a layer) and their
           # outputs large: what bounds them is data movement, not the matrix pipe (round-4 review) -- hence `bound: hbm`.
           "roofline": {"bound": "hbm", "kernel": "gemm_split_kernel / gemm_f32_kernel (all GEMM launches of the step)",
                        "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "traffic_source": f"profiles/{PROFILE_ROUND}_jta_gemm_hbm_bytes.json (committed --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per step; NOT measured in this run)",
                        "algorithmic_bytes_per_step": round(gemm_bytes / steps), "gemm_launches": n, "gemm_ms_per_step": round(ms / steps, 2),
                        "mfma_view": {"achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                                      "frac_of_fp32_instruction_peak": round(tf / 157.3, 4), "note": peak_note},
                        "note": "achieved = algorithmic bytes (every operand and the output once, in their memory dtypes) of the timed GEMM launches / "
                                "their HIP-event time; per-kernel TB/s from the counters: profiles/" + PROFILE_ROUND + "_jta_hbm_fp32_split.txt"}}
    # The reduced-precision mode of BASELINE configs[3] ("bf16 MFMA attention"), reported beside the fp32 figure under its own
    # parity bar (2e-2 on activations against the fp32 fixtures, tests/test_gpu_predictor.py): every linear-layer GEMM and
    # the fused attention's tile products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the two large activations of a layer
    # (feed-forward hidden M x 1024, q|k|v M x 384) and their gradients live in HBM as bf16; softmax statistics, LayerNorm, the
    # residual stream, weight gradients, losses and the optimiser stay fp32.
    try:
        ops.set_matmul_precision("bf16")
        _timed(lambda: trainer.step(joints, masks, pad), 0, warmup, world, dev)
        ops.gemm_timing(True)
        dt = _timed(lambda: trainer.step(joints, masks, pad), steps, 0, world, dev)
        n, ms, fl = ops.gemm_timing()
        gemm_bytes16 = ops.gemm_timing_bytes()
        ops.gemm_timing(False)
        # algorithmic flops of the step = padded person-sequences x 7.4 GFLOP (forward + backward of the 6 + 3 layers)
        seqs = B * int(joints.shape[1])
        step_tf = seqs * 7.4e9 / (dt / steps) / 1e12
        out["bf16"] = {"value": round(B * world * steps / dt, 2), "unit": "samples/s", "ms_per_step": round(dt / steps * 1e3, 2), "dtype": "bf16",
                       "roofline": {"bound": "mfma", "kernel": "whole step (GEMMs + fused attention)", "achieved": round(step_tf * world, 2),
                                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(step_tf / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                                    "gemm_tflops": round(fl / (ms * 1e-3) / 1e12, 2) if ms > 0 else None, "gemm_ms_per_step": round(ms / steps, 2),
                                    "gemm_gbs": round(gemm_bytes16 / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                                    "note": "achieved = padded person-sequences x 7.4 GFLOP (SURVEY 8d) / step time, against the dense bf16 MFMA "
                                            "peak; the step is far from it: at head dim 32 the attention kernels are bound by the vector work per "
                                            "probability (a quarter-rate exponential, the dropout hash, the softmax arithmetic: ~12 vector cycles against 2 "
                                            "matrix cycles), the GEMMs (feed-forward chained in registers, ffn_chain_kernel) by their data movement, "
                                            "LayerNorm / bias-gradient / residual passes are separate launches"},
                       "note": "ops.set_matmul_precision('bf16'): bf16 MFMA operands in the linear layers and the fused attention (round 5: two blocks of 32 "
                               "rows per wave, bf16 tile images), the feed-forward block as two chained launches (hidden tile in registers), the "
                               "hidden layer and q|k|v as bf16 in HBM; parity bar 2e-2 (SURVEY 8c), NOT the 1e-4 of the fp32 path that `value` reports"}
        out["bf16_operands"] = out["bf16"]                      # the name earlier rounds reported this leg under
    finally:
        ops.set_matmul_precision(ops.DEFAULT_PRECISION)
    # NOT the reference's model function: the same step with collate_batch's BOOL padding mask (cfg MASK_PADDED_PERSONS) -- padded
    # persons masked with -inf instead of biased by +1 (the reference's loops pass the float copy, dataset_jta.py:84), which lets the
    # local former skip them (model_jta.py `_transform`; pinned to the reference run with a bool mask, predictor_boolmask_jta.npz)
    try:
        cfg["MASK_PADDED_PERSONS"] = True
        live = int((~pad).sum())
        masked = {"live_person_sequences": live, "padded_person_sequences": int(pad.numel()),
                  "note": "extension, off by default: bool padding mask = padded persons masked (-inf) rather than biased (+1) as in the "
                          "reference's training loop; the local former then runs on the live person-sequences only.  A different "
                          "model function from `value` -- reported beside it, never as it"}
        for mode in (ops.DEFAULT_PRECISION, "bf16"):
            ops.set_matmul_precision(mode)
            _timed(lambda: trainer.step(joints, masks, pad), 0, warmup, world, dev)
            dt = _timed(lambda: trainer.step(joints, masks, pad), steps, 0, world, dev)
            masked[mode] = {"value": round(B * world * steps / dt, 2), "unit": "samples/s", "ms_per_step": round(dt / steps * 1e3, 2)}
        out["masked_padded_persons"] = masked
    finally:
        cfg["MASK_PADDED_PERSONS"] = False
        ops.set_matmul_precision(ops.DEFAULT_PRECISION)
    return out


def eval_leg(dev, B=512, batches=2, rank=0, world=1):
    """configs[4]: multi-modal JRDB predictor (20 heads) + LocoVal filter evaluation, batch 512 per GPU.  world > 1: the batches
    are dealt round-robin to the ranks (evaluate_ade_fde's shard), only the summary scalars / histograms are all-reduced."""
    import torch.distributed as dist
    import torch
    from emloco_amd.learning.value_pose_net import ValuePoseNet
    from emloco_amd.predictor.evaluate_jta import evaluate_ade_fde
    from emloco_amd.predictor.model_jrdb import TransMotionJRDB
    torch.manual_seed(0)
    cfg = {"DEVICE": str(dev), "MULTI_MODAL": True, "NOISY_TRAJ": 0, "TRAIN": {"input_track_size": 9, "output_track_size": 12},
           "MODEL": {"value_threshold": 0.8}, "DATA": {"train_datasets": ["jrdb_all_visual_cues"]}}
    model = TransMotionJRDB(tok_dim=246, nhid=128, nhead=4, dim_feedfwd=1024, nlayers_local=6, nlayers_global=3, nmode=20,
                            output_scale=1, obs_and_pred=21, num_tokens=26, device=str(dev), multi_modal=True).to(dev)
    vnet = ValuePoseNet(True, True).to(dev)
    g = torch.Generator().manual_seed(5)
    data = []
    for _ in range(batches * world):
        N = 8
        joints = torch.randn(B, N, 21, 26, 4, generator=g) * 0.3
        joints[:, :, :, 0, :2] = torch.cumsum(torch.randn(B, N, 21, 2, generator=g) * 0.4, dim=2)
        n_people = torch.randint(1, N + 1, (B,), generator=g)
        pad = torch.arange(N)[None, :] >= n_people[:, None]
        data.append((joints, torch.ones(B, N, 21, 26), pad))
    if world > 1:
        from emloco_amd.dist import broadcast_parameters
        broadcast_parameters(model, vnet)
    evaluate_ade_fde(model, vnet, "test", "traj+all", data[:world], B, cfg, dataset="jrdb")      # warm-up
    res = {}

    def run():
        res.update(evaluate_ade_fde(model, vnet, "test", "traj+all", data, B, cfg, dataset="jrdb"))
    dt = _timed(run, 1, 0, world, dev)
    return {"metric": "JRDB eval samples/sec (multi-modal predictor + LocoVal filter)", "value": round(B * batches * world / dt, 1),
            "unit": "samples/s", "n_gpus": world, "ms_per_batch": round(dt / batches * 1e3, 2), "dtype": "f32",
            "config": {"workload": f"configs[4] on {world} GPU(s): TransMotionJRDB 20 modes, batch 512 per GPU, people/scene U{{1..8}} padded to 8, "
                                   "246 tokens/person, LocoVal filter threshold 0.8 (host->device copy of the batch included)",
                       "batch": B, "batches": batches, "locoval_calls_per_batch": B * 40},
            "ade": round(float(res["ade"]), 4), "ade_value_sampling": round(float(res.get("ade_value", float("nan"))), 4)}


def jta_cpu_baseline(B=32, budget_s=60.0):
    """The same EmLoco train step on the host cores with the stock-torch restatement (oracle/predictor_torch.py): batch 32,
    thread count swept upwards over {8, 16, 32, all cores} while it helps (one warm-up + one timed iteration each), the best
    one timed again."""
    import torch
    from oracle.predictor_torch import LocoValOracle, TransMotionJTAOracle, emloco_train_step
    from emloco_amd.predictor.train_jta import batch_process_coords
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.manual_seed(0)
    model = TransMotionJTAOracle(dropout=0.0)
    vnet = LocoValOracle()
    for p in vnet.parameters():
        p.requires_grad_(False)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    joints, masks, pad = synthetic_jta_batch(B, seed=1, nan_frac=0.0)
    cfg = {"DEVICE": "cpu", "TRAIN": {"input_track_size": 9, "output_track_size": 12}}
    i, _, o, _, pm = batch_process_coords(joints, masks, pad, cfg, training=True)
    pose = joints[:, 0, 8, 3:27, :3].clone()
    vel = (i[:, 8, 0, :2] - i[:, 7, 0, :2]) * 2.5
    t_start = time.perf_counter()
    sweep = {}
    # more threads are tried only while they help: on the 256-thread host of the GPU box the step runs 3.9 samples/s on 8
    # threads, 3.6 on 16 and 0.23 on all 256 (profiles/r02_bench_default.log of the earlier build: two iterations at 256
    # threads alone took 4.6 minutes of the default run)
    prev = None
    for th in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu), ncpu}):
        torch.set_num_threads(th)
        emloco_train_step(model, vnet, opt, i, o, pm, pose, vel)            # warm-up at this thread count
        t0 = time.perf_counter()
        emloco_train_step(model, vnet, opt, i, o, pm, pose, vel)
        sweep[th] = time.perf_counter() - t0
        if time.perf_counter() - t_start > budget_s * 0.6 or (prev is not None and sweep[th] > prev):
            break
        prev = sweep[th]
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    iters, tot = 0, 0.0
    while iters < 3 and (iters == 0 or time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        emloco_train_step(model, vnet, opt, i, o, pm, pose, vel)
        tot += time.perf_counter() - t0
        iters += 1
    dt = min(tot / iters, sweep[best])
    return {"value": round(B / dt, 3), "unit": "samples/s", "cores": best, "kind": "port",
            "sample": f"the same train step at batch {B} x {joints.shape[1]} people, stock torch.nn fp32 restatement; thread sweep "
                      + ", ".join(f"{k}: {B / v:.2f}/s" for k, v in sweep.items()) + f" (more threads are tried while they help and the time budget lasts) of {ncpu} host cores; best = {best} threads, "
                      f"{iters} more timed iterations ({dt:.2f} s / iteration)"}


def policy_leg(env, E, dev, steps, warmup):
    """Row A19, reported beside the headline: the frozen PACER policy (random-init weights of the shipped architecture:
    obs-norm -> task MLP 1054-512-256 -> actor MLP 624-2048-1024-69, sigma = e^-2.9) drives the same rollout."""
    import torch
    import yaml
    from emloco_amd.learning.amp_network_sept_builder import AMPSeptBuilder
    from emloco_amd.learning.policy_runner import FrozenPolicy
    from emloco_amd.utils.running_mean_std import RunningMeanStd
    task = env.task
    cfg = yaml.safe_load(open(os.path.join(ROOT, "emloco_amd", "data", "cfg", "train", "rlg", "amp_humanoid_smpl_sept_task.yaml")))
    torch.manual_seed(0)
    rms = RunningMeanStd((1422,)).to(dev)
    rms.eval()
    b = AMPSeptBuilder()
    b.load(cfg["params"]["network"])
    net = b.build("amp", actions_num=69, input_shape=(1422,), num_seqs=1, value_size=1, amp_input_shape=(3090,),
                  self_obs_size=368, task_obs_size=1054, task_obs_size_detail={"traj": 30, "heightmap": 1024}, mean_std=rms).to(dev)
    pol = FrozenPolicy(net, rms, E, dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def one_step(timed):
        env.reset_done()
        task.wait_obs()                     # the observation launch of the last step runs on a side stream (task.overlap_obs)
        if hasattr(task, "wait_reset"):
            task.wait_reset()               # the policy reads the reset envs' fresh observations
        if timed:
            ev0.record()
        act = pol.act(task.obs_buf, deterministic=False, generator=gen)
        if timed:
            ev1.record()
        env.step(act)

    for _ in range(warmup):
        one_step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step(False)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pol_ms = []
    for _ in range(20):
        one_step(True)
        torch.cuda.synchronize()
        pol_ms.append(ev0.elapsed_time(ev1))
    pm = float(np.median(pol_ms))
    tf = pol.flops_per_env * E / (pm * 1e-3) / 1e12
    # the same loop with the opt-in bf16 operand mode of the GEMMs (reported beside the fp32 figure, never instead of it)
    from emloco_amd.predictor import ops
    peak, peak_note = gemm_peak(ops)
    precision = ops.get_matmul_precision()
    bf = None
    try:
        ops.set_matmul_precision("bf16")
        for _ in range(warmup):
            one_step(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step(False)
        torch.cuda.synchronize()
        el_bf = time.perf_counter() - t0
        pol_bf = []
        for _ in range(20):
            one_step(True)
            torch.cuda.synchronize()
            pol_bf.append(ev0.elapsed_time(ev1))
        bf = {"value": round(E * steps / el_bf, 1), "unit": "env-steps/s", "ms_per_step": round(el_bf / steps * 1e3, 4),
              "policy_ms": round(float(np.median(pol_bf)), 4),
              "note": "opt-in ops.set_matmul_precision('bf16'): ~2e-3 relative error per GEMM, outside the 1e-4 parity bar"}
    finally:
        ops.set_matmul_precision(ops.DEFAULT_PRECISION)
    return {"metric": "env-steps/sec with the frozen policy in the loop", "value": round(E * steps / elapsed, 1), "unit": "env-steps/s",
            "ms_per_step": round(elapsed / steps * 1e3, 4), "policy_ms": round(pm, 4), "bf16_operands": bf,
            "policy_flops_per_env": pol.flops_per_env,
            "precision": precision,
            "roofline": {"bound": "mfma", "kernel": "gemm_f32_kernel", "achieved": round(tf, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(tf / peak, 4), "frac_of_fp32_instruction_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                         "note": "normalise + 5 GEMM launches (bias / ReLU fused; round 5: the frozen weights read as piece images cut once), "
                                 "median of 20 HIP-event timings; " + peak_note},
            "weights": "random init (no checkpoint ships)"}


def ppo_leg(env, E, dev, epochs=2, warmup=1):
    """configs[1] END TO END as the reference reports it: one `train_epoch` = a 32-step rollout of all envs with the policy / critic
    acting (amp_continuous.py:98-321 play_steps) + the PPO / AMP update over the collected batch (6 mini-epochs of minibatches of
    2 560, amp_humanoid_smpl_sept_task.yaml:102-115; amp_continuous.py:335-479), and the two rates common_agent.py:183-194 logs:
    fps_step = frames / play_time, fps_total = frames / (play_time + update_time).  Random-init networks of the shipped architecture
    (11.2 M parameters: actor, critic, discriminator, task MLP), synthetic motion library for the AMP demonstrations."""
    import torch
    import yaml
    from emloco_amd.learning.amp_agent import AMPAgent
    from emloco_amd.predictor import ops
    cfg = yaml.safe_load(open(os.path.join(ROOT, "emloco_amd", "data", "cfg", "train", "rlg", "amp_humanoid_smpl_sept_task.yaml")))
    task = env.task
    for attr in ("fused_chain", "overlap_reset"):           # the learner's loop resets through env.reset(ids): the plain launches
        if hasattr(task, attr):
            setattr(task, attr, False)
    if hasattr(task, "attach_returns"):
        task.attach_returns(None)
    c = cfg["params"]["config"]
    batch = int(c["horizon_length"]) * E
    mb_cfg = int(c["minibatch_size"])
    if batch % mb_cfg:        # the shipped 2 560 divides the reference's 1 600 x 32 (pacer.yaml:10); rl_games asserts divisibility: largest divisor below it
        c["minibatch_size"] = max(d for d in range(1, mb_cfg + 1) if batch % d == 0)
    agent = AMPAgent(env, cfg)
    n_params = sum(p.numel() for p in agent.a2c_network.parameters())
    for _ in range(warmup):
        agent.train_epoch()                                  # (with use_graph: three eager optimiser steps, the capture, then replays)
    infos = []
    for _ in range(epochs):
        infos.append(agent.train_epoch())
    # GEMM rate of an epoch: HIP events around every GEMM launch cannot ride inside a replayed graph, so one more epoch runs eagerly for it
    graphed, agent.use_graph = agent.use_graph, False
    ops.gemm_timing(True)
    eager = agent.train_epoch()
    n, ms, fl = ops.gemm_timing()
    ops.gemm_timing(False)
    agent.use_graph = graphed
    play = sum(i["play_time"] for i in infos)
    total = sum(i["total_time"] for i in infos)
    frames = agent.batch_size * epochs
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    peak, peak_note = gemm_peak(ops)
    steps_per_epoch = agent.mini_epochs_num * (agent.batch_size // agent.minibatch_size)
    last = infos[-1]
    return {"metric": "env-steps/sec of PPO + AMP policy pretraining (configs[1] end to end: rollout + update)",
            "fps_step": round(frames / play, 1), "fps_total": round(frames / total, 1), "unit": "env-steps/s",
            "play_ms_per_epoch": round(play / epochs * 1e3, 1), "update_ms_per_epoch": round((total - play) / epochs * 1e3, 1),
            "epochs_timed": epochs, "horizon_length": agent.horizon_length, "batch_size": agent.batch_size, "minibatch_size": agent.minibatch_size,
            "minibatch_size_configured": mb_cfg, "mini_epochs": agent.mini_epochs_num, "optimizer_steps_per_epoch": steps_per_epoch, "parameters": n_params,
            "update_ms_per_optimizer_step": round((total - play) / epochs / steps_per_epoch * 1e3, 3),
            "optimizer_step_as_hip_graph": bool(graphed and agent._graph is not None),
            "graph_arms": ("actor | critic | discriminator | symmetry loss on streams of their own (parallel arms of the graph)"
                           if (graphed and getattr(agent, "_g_arms", False)) else "one chain"),
            "graph_trial_ms": getattr(agent, "_g_trial_ms", None),
            "eager_update_ms_per_optimizer_step": round(eager["update_time"] / steps_per_epoch * 1e3, 3),
            "losses": {k: round(float(last[k]), 5) for k in ("actor_loss", "critic_loss", "disc_loss", "kl") if k in last},
            "roofline": {"bound": "mfma", "kernel": "gemm_f32_kernel (rollout + update GEMMs of one eagerly issued epoch)", "achieved": round(tf, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(tf / peak, 4), "gemm_launches": n, "gemm_ms_per_epoch": round(ms, 1),
                         "traffic": None, "note": peak_note},
            "note": "fps_step / fps_total as common_agent.py:183-194 defines them (frames / play_time, frames / total_time), the host "
                    "synchronisations of the reference's loop included (dones.nonzero() every step, the epoch's statistics); the learner steps the "
                    "env through env.reset(ids) + env.step, not through the fused chain of the LocoVal loop.  Round 5: clip + Adam as four "
                    "launches on flat buffers with a device-side step count, the actor / critic / discriminator loss heads as fused launches "
                    "(profiles/r05_ab_ppo_update.txt, r05_ppo_step_trace.txt)"}


def locoval_policy_leg(env, E, dev, steps, warmup):
    """configs[2] as the reference runs it on one GPU (amp_continuous_value.py:45-145): the frozen policy acts on the observations,
    the AMP discriminator scores each step's AMP observations (the style half of the LocoVal return), LocoVal is fitted on the
    finished episodes -- `python -m emloco_amd.run --policy_random_init`'s loop.  Random-init weights of the shipped architectures."""
    import torch
    from emloco_amd.learning.amp_policy import AMPPolicyBundle
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    task = env.task
    bundle = AMPPolicyBundle(task, seed=0)
    agent = LocoValRollout(env, horizon_length=32, policy=bundle.policy, disc_reward=bundle.disc_reward, overlap_reset=False)
    agent.started = True
    agent._sched_live = True
    for k in range(warmup):
        agent.step_once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        agent.step_once()
        if (k + 1) % 32 == 0:
            agent.end_epoch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    agent._sync_fit()
    return {"metric": "env-steps/sec, frozen policy + AMP discriminator reward + LocoVal fit in the loop (configs[2] on one GPU)",
            "value": round(E * steps / dt, 1), "unit": "env-steps/s", "ms_per_step": round(dt / steps * 1e3, 4),
            "episodes_fitted": agent.fitted_episodes,
            "discriminator_deferred": agent._disc_halves is not None,
            "note": "policy forward (5 GEMMs) on the chain; the discriminator (3 GEMMs, 3090 -> 1024 -> 512 -> 1, split-mode matrix path) OFF it: "
                    "its style reward feeds the return bookkeeping only, so the flags launch stages the step (LocoVal inputs, penalised reward, done "
                    "flag), one launch takes the normalised GEMM operand out of the AMP observations before the resets, and a side stream runs the "
                    "GEMMs, emloco_locoval_returns_finish and the fit beside the resets / policy / next rigid-body launch; results bit-equal to "
                    "the sequential order (tests/test_gpu_env.py); EMLOCO_DEFER_DISC=0 puts the discriminator back between env.step and the reset"}


def summary_of(out):
    """Compact digest of the line (numbers only, no notes), printed as its last key so that a reader of the TAIL of stdout sees every leg:
    headline, the rigid-body kernel, env_step_only, policy, configs[2] (frozen policy + discriminator reward + LocoVal fit), the PPO
    learner, the JTA train step in both precision classes, evaluation, the CPU baselines."""
    def g(d, *ks):
        for k in ks:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    s = {"headline_env_steps_per_s": out.get("value"), "ms_per_step": out.get("ms_per_step"), "n_gpus": out.get("n_gpus"),
         "sim_kernel_ms": g(out, "roofline", "kernel_ms"), "roofline_frac": g(out, "roofline", "frac"),
         "env_step_only": g(out, "env_step_only", "value"),
         "policy": g(out, "policy", "value"), "policy_ms": g(out, "policy", "policy_ms"),
         "configs2_policy_disc_locoval_fit": g(out, "policy", "with_discriminator_and_locoval_fit", "value"),
         "configs2_ms_per_step": g(out, "policy", "with_discriminator_and_locoval_fit", "ms_per_step"),
         "ppo_fps_total": g(out, "policy", "ppo", "fps_total"), "ppo_fps_step": g(out, "policy", "ppo", "fps_step"),
         "ppo_update_ms_per_optimizer_step": g(out, "policy", "ppo", "update_ms_per_optimizer_step"),
         "cpu_baseline_env_steps_per_s": g(out, "cpu_baseline", "value"), "cpu_baseline_cores": g(out, "cpu_baseline", "cores"),
         "jta_samples_per_s": g(out, "jta", "value"), "jta_ms_per_step": g(out, "jta", "ms_per_step"), "jta_steps_timed": g(out, "jta", "steps"),
         "jta_roofline_frac": g(out, "jta", "roofline", "frac"),
         "jta_bf16_samples_per_s": g(out, "jta", "bf16", "value"), "jta_bf16_ms_per_step": g(out, "jta", "bf16", "ms_per_step"),
         "jta_bf16_roofline_frac": g(out, "jta", "bf16", "roofline", "frac"),
         "eval_samples_per_s": g(out, "jta", "eval", "value"),
         "jta_cpu_baseline_samples_per_s": g(out, "jta", "cpu_baseline", "value")}
    return {k: v for k, v in s.items() if v is not None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--num_envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_policy", action="store_true", help="skip the frozen-policy leg (row A19, reported separately)")
    ap.add_argument("--no_jta", action="store_true", help="skip the train_jta samples/s leg (run on rank 0 at N=1)")
    ap.add_argument("--no_ppo", action="store_true", help="skip the PPO + AMP train_epoch leg (configs[1] end to end; rank 0 at N=1, under `policy`)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher -- one process per GPU, as the driver's
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` does
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    from emloco_amd import _lib as L
    from emloco_amd.dist import init_from_env
    L.require_device()                                  # fail loudly: no CPU fallback
    share = os.environ.get("EMLOCO_BENCH_SHARE_GPU") == "1"
    rank, local_rank, world = init_from_env("gloo" if share else "nccl")
    local_rank = rank_local()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    E = a.num_envs
    env = make_env(E, rank)
    task = env.task
    # The schedule: the reference's order -- reset chain of the finished envs, then one rigid-body launch for all envs, dispatched
    # most-contact-work-first (emloco_sim_set_cost_order).  Every consumer can use it: the observations of the reset envs exist
    # before the policy runs (amp_continuous_value.py:46-52).  (Rounds 2-3 also reported a two-chain schedule for observation-blind
    # consumers and a two-shard pipeline: both lost to this one once the chain between two steps was fused, and are gone.)
    n_parts = int(os.environ.get("EMLOCO_SPLIT", "4"))
    task.sim.native.set_cost_order(os.environ.get("EMLOCO_COST_ORDER", "1") != "0")
    env.reset(torch.arange(E, device=dev))
    stagger_episodes(env, seed=rank)                     # untimed: episode ages uniform over [0, 168) before the warm-up
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    pool = torch.randn(64, E, 69, device=dev, generator=g) * float(np.exp(-2.9))
    # the LocoVal-training rollout loop (amp_continuous_value.py:45-145) around env.step, identical code at every N:
    # reset_done -> env.step -> returns / cut-off bookkeeping -> LocoVal fit -> one all-reduce (6 176 floats) -> gated AdamW
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    counter = [0]

    def noise_policy(obs):          # stands where the frozen policy stands (its network is the `policy` leg); like a real policy it
        counter[0] += 1             # is handed the observations only once the reset chain has written the reset envs' rows
        return pool[counter[0] % 64]

    horizon = 32

    def timed_loop(agent):
        def one_step(k):
            agent.step_once()
            if (k + 1) % horizon == 0:
                agent.end_epoch()
        # untimed, part of the set-up like the pre-roll above: the loop object has just been built (hundreds of ms of host work with
        # the GPU idle: its clocks have dropped, the loop's staging rings and the dispatch order are cold).  128 steps of the same
        # loop bring both to their steady state, so that a short window (--steps 20 --warmup 5) measures what a long one does
        # (measured: 0.465 ms per step without them at --warmup 5, 0.453 at --warmup 20, 0.448 at --steps 200).
        for k in range(int(os.environ.get("EMLOCO_BENCH_SETTLE", "128"))):
            one_step(k)
        for k in range(a.warmup):
            one_step(k)
        task.sim.native.enable_timing(True, every=8)     # HIP events around every 8th launch of the timed region (an event record costs ~6 us of stream time)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            one_step(k)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_l, ms_l = task.sim.native.timing_stats()
        return dt, n_l, ms_l

    agent = LocoValRollout(env, horizon_length=horizon, policy=noise_policy, overlap_reset=False)
    agent.started = True                                 # the envs are already reset and staggered
    agent._sched_live = True                             # (the schedule's first-episode check is a host read; not in the timed loop)
    elapsed, n_l, ms_l = timed_loop(agent)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        from emloco_amd.dist import all_reduce_
        all_reduce_(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fitted, vloss = agent.fitted_episodes, agent.vnet_loss

    # env.step alone in the same (sequential) schedule, no LocoVal bookkeeping / fit: reported beside the headline at N = 1
    env_only = None
    if world == 1:
        agent._sync_fit()
        agent.detach()                                   # the LocoVal return bookkeeping leaves the task's flags launch
        for k in range(a.warmup):
            env.reset_done(); env.step(pool[k % 64])
        task.sim.native.enable_timing(True, every=8)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(a.steps):
            env.reset_done(); env.step(pool[k % 64])
        torch.cuda.synchronize()
        env_only = time.perf_counter() - t1
        n_s, ms_s = task.sim.native.timing_stats()

    if rank == 0:
        kernel_ms = ms_l / max(n_l, 1)
        achieved = SIM_BYTES_PER_ENV * E / (kernel_ms * 1e-3) / 1e9 if n_l else 0.0
        prof = {}                 # numbers of the committed counter passes of this round's kernel: pointers, not measurements of this run
        try:
            prof["traffic"] = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_sim_step_hbm_bytes.json"))).get("hbm_bytes_per_launch")
        except Exception:
            prof["traffic"] = None
        try:
            for line in open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_sim_step_valu.txt")):
                if line.startswith("VALU issue utilisation"):
                    prof["valu_issue_utilisation"] = float(line.split("=")[-1].split()[0])
        except Exception:
            pass
        out = {
            "metric": "env-steps/sec (SMPL humanoid, num_envs)", "value": round(E * world * a.steps / elapsed, 1),
            "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1] env: PACER rollout env.step, {E} SMPL humanoids per GPU, random_heading, "
                                   "JTA+JRDB-shaped real_path (synthetic), flat terrain, self-collision on, steady-state resets included "
                                   "(episode ages pre-staggered), inside the LocoVal-training loop of configs[2] (returns bookkeeping + LocoVal "
                                   "fit + gradient all-reduce every step), policy network excluded (its forward is the `policy` leg); "
                                   "schedule: the reference's order (reset chain of the finished envs, observations, then ONE rigid-body launch "
                                   "for all envs, dispatched most-contact-work-first) -- what a policy that reads the observations can use; "
                                   "the launches between two rigid-body steps folded into four on one stream (task.fused_chain)",
                       "locoval": {"episodes_fitted": fitted, "last_fit_loss": round(vloss, 5), "exchange_floats_per_step": 6176},
                       "num_envs_per_gpu": E, "substeps_per_step": 4, "workgroups_per_env": n_parts, "parallelism": f"env-sharded x{world}" + (" (TEST MODE: all ranks share cuda:0, gloo)" if share else "")},
            "roofline": {"bound": "hbm", "kernel": "sim_step_kernel", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": prof.get("traffic"),
                         "traffic_source": f"profiles/{PROFILE_ROUND}_sim_step_hbm_bytes.json (committed PMC passes of this kernel; NOT measured in this run)",
                         "kernel_ms": round(kernel_ms, 4), "launches_timed": n_l,
                         "from_profiles": {"valu_issue_utilisation": prof.get("valu_issue_utilisation"), "source": f"profiles/{PROFILE_ROUND}_sim_step_valu.txt (committed SQ-counter pass; NOT measured in this run)"},
                         "note": "achieved = algorithmic bytes (9 296 B per env per launch, DESIGN.md section 5) / kernel_ms, both live: HIP events on "
                                 "every 8th launch of the timed region.  The kernel is instruction / latency bound, not bandwidth bound: ~9 KB of state "
                                 "per env per launch against ~47 k fp32 VALU wave-instructions (level-synchronous tree passes; 3 waves per SIMD, 12 envs "
                                 "per CU; each env's 4 substeps run as four dependent workgroups of one launch); VALU issue utilisation (wave64 = 2 issue cycles on a "
                                 "SIMD-32) in `from_profiles`: about half the issue slots, the rest is dependent-instruction latency and LDS"},
        }
        if env_only is not None:
            k_alone = ms_s / max(n_s, 1)
            out["env_step_only"] = {"value": round(E * a.steps / env_only, 1), "unit": "env-steps/s", "ms_per_step": round(env_only / a.steps * 1e3, 4),
                                    "kernel_ms": round(k_alone, 4), "launches_timed": n_s,
                                    "roofline_frac_alone": round(SIM_BYTES_PER_ENV * E / (k_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if n_s else None,
                                    "note": "reset_done + env.step in the same sequential schedule without the LocoVal bookkeeping / fit; kernel_ms = "
                                            "sim_step_kernel with nothing but the observation launch of the previous step beside it"}
            out["sequential"] = dict(out["env_step_only"])         # the name earlier rounds reported this leg under
        if world == 1 and not a.no_policy:
            # a policy reads the reset envs' fresh observations, so the reset chain cannot hide beside the step: the sequential
            # schedule (observation launch of the live envs beside the reset chain, cost-ordered dispatch) is the faster one here
            # (measured: 3.95 M against 3.78 M env-steps/s)
            task.overlap_obs = True
            out["policy"] = policy_leg(env, E, dev, a.steps, a.warmup)
            out["policy"]["schedule"] = "sequential, cost-ordered dispatch"
            out["policy"]["with_discriminator_and_locoval_fit"] = locoval_policy_leg(env, E, dev, a.steps, a.warmup)
            if not a.no_ppo:
                out["policy"]["ppo"] = ppo_leg(env, E, dev)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    # the JTA train-step and evaluation legs run on every rank (data parallel at N > 1); rank 0 reports
    if not a.no_jta:
        del agent, env, task, pool
        torch.cuda.empty_cache()
        jta = jta_leg(dev, rank=rank, world=world)
        jta["eval"] = eval_leg(dev, rank=rank, world=world)
        if rank == 0:
            out["jta"] = jta
            if world == 1 and not a.no_cpu_baseline:
                out["jta"]["cpu_baseline"] = jta_cpu_baseline()
    if rank == 0:
        out["summary"] = summary_of(out)              # LAST key of the line: the numbers of every leg inside the final kilobyte of stdout
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
