"""emloco_amd: MI355X-native hot path of EmLoco."""
import os as _os
import sys as _sys

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  This package runs several streams side by
# side (the LocoVal fit beside the rollout chain, the discriminator beside the next step, the parallel arms of the captured PPO step):
# when two of them land on one hardware queue they serialise -- the PPO step as a graph with four arms takes 9.4 ms instead of 4.7
# (profiles/r04_ppo_hw_queues.txt).  Raised to 16 here unless the caller has chosen a value; the runtime reads it when it initialises,
# so this only helps a process that imports the package before its first GPU call (every entry point of the package does).
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _t = _sys.modules.get("torch")
    if _t is None or not _t.cuda.is_initialized():
        _os.environ["GPU_MAX_HW_QUEUES"] = "16"
