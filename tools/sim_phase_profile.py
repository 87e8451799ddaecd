"""Per-phase cycle profile of sim_step_kernel (env 0, lane 0; wall_clock64 ticks at 100 MHz).

Run on a GPU box:   EMLOCO_HIPCC_EXTRA="-DEMLOCO_SIM_PROFILE=1" python -m emloco_amd.build && python tools/sim_phase_profile.py
The stamps exist only in a library built with -DEMLOCO_SIM_PROFILE=1 (the product build carries none).
"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from emloco_amd import _lib as L
from emloco_amd.sim import NativeSim
from helpers import varied_models

PHASES = ["kinematics", "inertia+bias", "factorise", "down pass", "contact candidates", "rows/chain y", "A build",
          "PGS", "impulse solve", "integrate"]
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
models = varied_models(64, seed=11); models = [models[i % 64] for i in range(E)]
sim = NativeSim(models, L.default_sim_params())
sim.root_state[:, 2] = 0.93
lib = L.load()
lib.emloco_sim_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
buf = (C.c_longlong * 256)()
lib.emloco_sim_profile(sim._h, buf, 256)          # first call allocates the stamp buffer
for _ in range(40): sim.step(2)                    # settle onto the ground so contacts are active
torch.cuda.synchronize()
print("contacts of env 0 (bodies with |F| > 0):", int((sim.contact_force.view(E, 24, 3)[0].abs().sum(1) > 0).sum()))
lib.emloco_sim_profile(sim._h, buf, 256)
t = np.array(buf[:], dtype=np.int64).reshape(16, 16)
n_sub = 4
print(f"E={E}: ticks (100 MHz) per phase, substeps 0..{n_sub-1}")
for i, name in enumerate(PHASES):
    d = [int(t[s, i + 1] - t[s, i]) for s in range(n_sub)]
    print(f"  {name:20s} " + " ".join(f"{x:6d}" for x in d))
print(f"  {'PGS: setup+warm start':20s} " + " ".join(f"{int(t[s, 11] - t[s, 7]):6d}" for s in range(n_sub)))
print(f"  {'PGS: first sweep':20s} " + " ".join(f"{int(t[s, 12] - t[s, 11]):6d}" for s in range(n_sub)))
print(f"  {'substep total':20s} " + " ".join(f"{int(t[s, 10] - t[s, 0]):6d}" for s in range(n_sub)))
