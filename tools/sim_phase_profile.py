"""Per-phase cycle profile of sim_step_kernel (env 0, lane 0; wall_clock64 ticks at 100 MHz).

Run on a GPU box:   EMLOCO_HIPCC_EXTRA="-DEMLOCO_SIM_PROFILE=1" python -m emloco_amd.build && python tools/sim_phase_profile.py [4096] [--pair]
(--pair: the library was built with EMLOCO_HIPCC_EXTRA_SIM="-DEMLOCO_SIM_PAIR=1 -DEMLOCO_SIM_PROFILE=1")
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
E = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
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
if "--pair-all" in sys.argv:      # segment times summed over ALL pairs since the buffer was cleared (PACC in sim_pair_kernels.hip)
    names = ["1 kinematics + 1b (joint)", "2 drive (joint)", "2b-3 inertia, factorise (joint)", "4 down pass (joint)", "5 candidates (joint)",
             "fast: 6a rows / chain y", "fast: 6b matrices -> registers", "fast: 6c setup + warm start", "fast: 6c sweeps", "fast: 7a impulses",
             "full-size path: 6a-7a, both envs", "7 tree passes (joint)", "8 integrate (joint)"]
    tt, cnt = np.array(buf[128:128 + 13], dtype=np.float64), np.array(buf[160:160 + 13], dtype=np.float64)
    n_pairsub = cnt[0]
    print(f"pair-substeps profiled: {int(n_pairsub)}; fast path {int(cnt[9])} ({cnt[9] / max(n_pairsub, 1):.3f}), full-size path {int(cnt[10])}")
    print("segment: mean ticks (10 ns) when it runs | share of all pair-substep time")
    for i, nm in enumerate(names):
        if cnt[i] > 0:
            print(f"  {nm:36s} {tt[i] / cnt[i]:8.1f}   {tt[i] / tt.sum():6.3f}")
    print(f"  mean pair-substep: {tt.sum() / max(n_pairsub, 1):.1f} ticks")
    sys.exit(0)
def row(name, a, b):
    print(f"  {name:34s} " + " ".join(f"{int(t[s, b] - t[s, a]):6d}" for s in range(n_sub)))
if "--pair" not in sys.argv:        # sim_kernels.hip (one env per wave)
    for i, name in enumerate(PHASES):
        row(name, i, i + 1)
    row("PGS: setup+warm start", 7, 11)
    row("PGS: first sweep", 11, 12)
    row("substep total", 0, 10)
    sys.exit(0)
# --pair: sim_pair_kernels.hip (-DEMLOCO_SIM_PAIR=1, two envs per wave): stamps of the pair that holds env 0; joint = both envs in one
# instruction stream
row("1 kinematics + 1b limb-limb (joint|x2)", 0, 1)
row("2 drive (joint)", 1, 2)
row("2b-3 inertia, factorise (joint)", 2, 3)
row("4 down pass (joint)", 3, 4)
row("5 candidates, env 0", 4, 5)
row("6a rows / chain y, env 0", 5, 6)
row("6b matrix -> registers, env 0", 6, 13)
row("5 candidates, env 1", 13, 14)
row("6a rows / chain y, env 1", 14, 15)
row("6b matrix -> registers, env 1", 15, 7)
row("6c Gauss-Seidel, both envs", 7, 8)
row("   setup + warm start", 7, 11)
row("   first sweep", 11, 12)
row("7a+7 impulses, tree passes", 8, 9)
row("8 integrate (joint)", 9, 10)
row("substep total (pair)", 0, 10)
