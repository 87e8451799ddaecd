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
# Who sets it (round 6: opt-IN): the ENTRY POINTS -- bench.py, emloco_amd.run, the train / evaluate mains, __graft_entry__ -- call
# `configure_runtime()` ahead of their first GPU call.  A plain `import emloco_amd` inside a host application changes nothing in the
# process environment: it records what the runtime will be (or was) initialised with -- the caller's GPU_MAX_HW_QUEUES, else the
# runtime's default of 4 -- and the schedules that depend on the queue count (learning/amp_agent.py, learning/locoval_rollout.py) fall
# back to their 4-queue forms.  A host that wants the 16-queue schedules calls `emloco_amd.configure_runtime()` itself before its
# first GPU call, or exports the variable.
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


if not _runtime_up():
    try:                                                         # record, never set: the caller's value, else the runtime's default
        _HW_QUEUES = int(_os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        _HW_QUEUES = None
