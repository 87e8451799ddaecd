"""Rigid-body launch time against the number of envs (standing humanoids with self-collision, 4 parts per env as in the bench): does the
launch pay for whole rounds of the 3 072 resident waves (12 per CU x 256), i.e. is there a tail to win?    python tools/exp/probe_sim_scaling.py"""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from emloco_amd import _lib as L                    # noqa: E402
from emloco_amd.sim import NativeSim                # noqa: E402
from helpers import varied_models                   # noqa: E402

for E in (768, 1536, 3072, 3584, 4096, 4608, 6144, 8192, 12288):
    models = varied_models(64, seed=11)
    models = [models[i % 64] for i in range(E)]
    sim = NativeSim(models, L.default_sim_params(), self_collision=True)
    sim.set_split(4)
    sim.root_state[:, 2] = 0.93
    for _ in range(20):
        sim.step(2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(30):
        e0.record()
        sim.step(2)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    t = ts[len(ts) // 2]
    print(f"E={E:6d}  wave-tasks {4 * E:6d} = {4 * E / 3072:5.2f} rounds of 3072   launch {t * 1e3:7.1f} us   {t * 1e6 / E:6.2f} ns per env   {t * 1e3 / (4 * E / 3072):6.1f} us per round")
    del sim
