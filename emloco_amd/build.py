"""Build libemloco_hip.so (gfx950 only) in-tree with hipcc.

    python -m emloco_amd.build            # rebuild if sources are newer than the library

The shared object lands in emloco_amd/lib/ so it travels with the repo snapshot to the GPU box
(built artefacts are git-ignored, not gpurun-ignored).  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libemloco_hip.so")

# translation units: (source, extra flags).  The rollout kernels are built without fused-multiply-add
# contraction so that fp32 body states track the CPU oracle's plain C arithmetic (tests/test_gpu_sim.py).
UNITS = [
    ("sim_capi.hip", ["-ffp-contract=off"]),
    ("task_capi.hip", ["-ffp-contract=off"]),
    ("predictor_capi.hip", []),
]
# EMLOCO_HIPCC_EXTRA="-DFOO=1 ..." appends flags to every unit (A/B experiments)
EXTRA = os.environ.get("EMLOCO_HIPCC_EXTRA", "").split()
COMMON = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libemloco_hip.so cannot be built")
    return exe


def _deps():
    out = []
    for root, _, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files]
    for h in ("emloco_sim.h", "emloco_task.h", "emloco_predictor.h"):
        out.append(os.path.join(os.path.dirname(HERE), "include", h))
    return [p for p in out if os.path.exists(p)]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for src, extra in UNITS:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc()] + COMMON + extra + ["-I", CSRC, "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out:
            print(out.decode())
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
