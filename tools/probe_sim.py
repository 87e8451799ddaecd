import sys, time, torch, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from emloco_amd import _lib as L
from emloco_amd.sim import NativeSim
from helpers import varied_models
for E in (4096, 16384):
    models = varied_models(64, seed=11); models = [models[i % 64] for i in range(E)]
    sim = NativeSim(models, L.default_sim_params())
    sim.root_state[:, 2] = 0.93
    sim.enable_timing(True)
    for _ in range(5): sim.step(2)
    torch.cuda.synchronize()
    t0 = time.time(); n = 50
    for _ in range(n): sim.step(2)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print(f"E={E}: {dt*1e3:.3f} ms per env.step (event: {sim.last_step_ms():.3f} ms) -> {E/dt:,.0f} env-steps/s")
