"""emloco_amd: MI355X-native hot path of EmLoco."""
import os as _os
import sys as _sys

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  This package runs several streams side by
# side (the LocoVal fit beside the rollout chain, the discriminator beside the next step, the parallel arms of the captured PPO step):
# when two of them land on one hardware queue they serialise -- the PPO step as a graph with four arms takes 9.4 ms instead of 4.7
# (profiles/r04_ppo_hw_queues.txt).  The runtime reads the variable ONCE, when it initialises.
#
# `hw_queues()` is what the package believes the runtime runs with (None: unknown -- the runtime was up before the package could see
# or set the variable); schedules that depend on the queue count key off it, not off the environment of the moment.
#
# Who sets it: the package's ENTRY POINTS (bench.py, emloco_amd.run, the train / evaluate mains) call `configure_runtime()` ahead of
# their first GPU call.  A plain `import emloco_amd` inside a host application does the same unless EMLOCO_KEEP_HW_QUEUES=1 is set
# (opt-out: the library then leaves the process environment alone and runs on whatever the host chose), and never touches a value the
# caller has exported.
_HW_QUEUES = None


def _runtime_up():
    t = _sys.modules.get("torch")
    return t is not None and t.cuda.is_initialized()


def configure_runtime(hw_queues=16, force=False):
    """Ask the HIP runtime for `hw_queues` hardware queues.  Effective only before the runtime initialises; a value already in the
    environment wins unless `force`.  Returns what the runtime will run with (None when it was already up with an unknown value)."""
    global _HW_QUEUES
    if _runtime_up():
        return _HW_QUEUES                                        # too late to change; report what was recorded (None: never seen)
    if force or "GPU_MAX_HW_QUEUES" not in _os.environ:
        _os.environ["GPU_MAX_HW_QUEUES"] = str(int(hw_queues))
    try:
        _HW_QUEUES = int(_os.environ["GPU_MAX_HW_QUEUES"])
    except ValueError:
        _HW_QUEUES = None
    return _HW_QUEUES


def hw_queues():
    """Hardware queues the HIP runtime was (or will be) initialised with, as far as this package can know; None = unknown."""
    return _HW_QUEUES


if _runtime_up():
    pass                                                         # the host initialised the GPU first: nothing is changed, nothing assumed
elif _os.environ.get("EMLOCO_KEEP_HW_QUEUES", "0") == "1":
    try:                                                         # opt-out: record the host's choice (the runtime default is 4)
        _HW_QUEUES = int(_os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        _HW_QUEUES = None
else:
    configure_runtime()
