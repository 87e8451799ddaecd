"""gymutil: command-line plumbing with the reference's flag names (isaacgym/python/isaacgym/gymutil.py:298-368)."""
import argparse

from . import gymapi


def parse_device_str(device_str):
    device, device_id = "cpu", 0
    if device_str in ("cpu", "cuda"):
        device = device_str
    else:
        parts = device_str.split(":")
        assert len(parts) == 2 and parts[0] == "cuda", f'Invalid device string "{device_str}"'
        device, device_id = parts[0], int(parts[1])
    return device, device_id


def parse_arguments(description="emloco", headless=False, no_graphics=False, custom_parameters=(), argv=None):
    parser = argparse.ArgumentParser(description=description)
    if headless:
        parser.add_argument("--headless", action="store_true", help="Run headless without creating a viewer window")
    if no_graphics:
        parser.add_argument("--nographics", action="store_true", help="Disable graphics context creation")
    parser.add_argument("--sim_device", type=str, default="cuda:0", help="Physics device in PyTorch-like syntax")
    parser.add_argument("--pipeline", type=str, default="gpu", help="Tensor API pipeline (cpu/gpu)")
    parser.add_argument("--graphics_device_id", type=int, default=0, help="Graphics Device ID")
    ptype = parser.add_mutually_exclusive_group()
    ptype.add_argument("--flex", action="store_true", help="Use FleX for physics")
    ptype.add_argument("--physx", action="store_true", help="Use PhysX for physics")
    parser.add_argument("--num_threads", type=int, default=0, help="Number of cores used by PhysX")
    parser.add_argument("--subscenes", type=int, default=0, help="Number of PhysX subscenes to simulate in parallel")
    parser.add_argument("--slices", type=int, help="Number of client threads that process env slices")
    for arg in custom_parameters:
        if "name" not in arg:
            continue
        kw = {k: v for k, v in arg.items() if k != "name"}
        parser.add_argument(arg["name"], **kw)
    args = parser.parse_args(argv)
    args.sim_device_type, args.compute_device_id = parse_device_str(args.sim_device)
    pipeline = args.pipeline.lower()
    assert pipeline in ("cpu", "gpu"), f"Invalid pipeline '{args.pipeline}'. Should be either cpu or gpu."
    args.use_gpu_pipeline = pipeline in ("gpu", "cuda")
    if args.sim_device_type != "cuda" or not args.use_gpu_pipeline:
        # the reference can run PhysX on the host (--pipeline=cpu); this build has no CPU pipeline
        raise SystemExit("emloco: only --sim_device cuda:N --pipeline gpu is available (no CPU physics path)")
    args.physics_engine = gymapi.SIM_PHYSX
    args.use_gpu = True
    if args.slices is None:
        args.slices = args.subscenes
    return args


def parse_sim_config(sim_cfg, sim_options):
    """gymutil.parse_sim_config: copy a yaml `sim:` block onto SimParams (config.py:165-166)."""
    for opt in ("dt", "substeps", "use_gpu_pipeline", "num_client_threads"):
        if opt in sim_cfg:
            setattr(sim_options, opt, sim_cfg[opt])
    if "gravity" in sim_cfg:
        sim_options.gravity = gymapi.Vec3(*sim_cfg["gravity"])
    for block in ("physx", "flex"):
        if block in sim_cfg:
            for k, v in sim_cfg[block].items():
                if hasattr(getattr(sim_options, block), k):
                    setattr(getattr(sim_options, block), k, v)
