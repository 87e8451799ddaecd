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
# -fno-slp-vectorize on the rigid-body unit: the SLP vectoriser pairs the kernel's scalar fp32 arithmetic into v_pk_fma / v_pk_mul /
# v_pk_add_f32, and the register-pair shuffling that feeds them costs more than the pairs save in this kernel (static count: 943 -> 417
# v_mov_b32, 48 -> 0 bytes of scratch at the 168-register cap) -- measured 0.392 -> 0.351 ms per launch at 4096 envs, same bytes
# (tests/test_gpu_sim.py); eleven further scheduler / vectoriser switches were within +-1 % or worse (tools/exp/run_flag_variants.sh).
UNITS = [
    ("sim_capi.hip", ["-ffp-contract=off", "-fno-slp-vectorize"]),
    ("task_capi.hip", ["-ffp-contract=off"]),
    ("predictor_capi.hip", ["-fno-slp-vectorize"]),      # split-mode GEMMs: 183.6 -> 176.5 ms per fp32-class train step, policy 0.341 -> 0.311 ms
    ("attention_capi.hip", []),                          # the fused attention keeps the SLP vectoriser (bf16 kernels 8-18 % slower without)
    ("ffn_capi.hip", []),                                # the chained feed-forward kernels (round 5)
    ("ppo_capi.hip", []),                                # the PPO learner's loss heads (round 5)
]
# EMLOCO_HIPCC_EXTRA="-DFOO=1 ..." appends flags to every unit, EMLOCO_HIPCC_EXTRA_SIM / _TASK / _PREDICTOR to one (A/B experiments)
EXTRA = os.environ.get("EMLOCO_HIPCC_EXTRA", "").split()
UNIT_EXTRA = {u: os.environ.get("EMLOCO_HIPCC_EXTRA_" + u.split("_")[0].upper(), "").split() for u in ("sim_capi.hip", "task_capi.hip", "predictor_capi.hip", "attention_capi.hip", "ffn_capi.hip", "ppo_capi.hip")}
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
        cmd = [hipcc()] + COMMON + extra + UNIT_EXTRA.get(src, []) + ["-I", CSRC, "-c", path, "-o", obj]
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
