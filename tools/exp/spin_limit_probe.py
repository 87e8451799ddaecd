"""Why the link angular speed is capped at 0.4 rad / substep (48 rad/s) rather than at the asset's 100 rad/s.

Free-floating humanoid, drives on, no gravity, spun about the vertical at w0; prints (control step, |w_root|, max link |w|,
|L| about the origin, kinetic energy).  `max_ang_vel` = 48 reproduces the shipped cap.  Result recorded in DESIGN.md section 3:
with the cap, |L| and T settle (101.7 -> 92.7, T 2368 -> 948 as the limbs extend); the variant that kept the velocities and only
scaled the velocity-product terms by (48 / w)^2 let |L| grow 4x in one second at 80 rad/s (192 -> 806) -- scaling Coriolis /
centrifugal terms breaks angular-momentum conservation, so it was not adopted."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle
from emloco_amd.model import pack_models, smpl_humanoid
from test_oracle_physics import _momenta
m = smpl_humanoid()
for w0, mav in ((80.0, 100.0), (150.0, 100.0), (30.0, 100.0)):
    s = oracle.Sim(pack_models([m]), oracle.default_params(gravity_z=0.0, ang_damping=0.0, max_ang_vel=mav))
    s.root_state[0, :3] = [52, 55, 50]
    s.root_state[0, 10:13] = [0, 0, w0]
    out = []
    for t in range(31):
        s.step()
        P, L, T = _momenta(m, s.rb_state[0])
        wl = np.linalg.norm(s.rb_state[0][:, 10:13], axis=1)
        if t % 5 == 0:
            out.append((t, round(float(wl[0]), 1), round(float(wl.max()), 1), round(float(np.linalg.norm(L)), 1), round(float(T))))
    print(w0, mav, out)
