"""Global mutable flag bag, as in the reference (pacer/pacer/utils/flags.py; filled by run.py:263-331)."""


class Flags(object):
    def __init__(self, items):
        for key, val in items.items():
            setattr(self, key, val)


# defaults = what run.py assigns when no option is given
flags = Flags(dict(
    test=False, debug=False, follow=False, fixed=True, divide_group=False, no_collision_check=False,
    fixed_path=False, real_path=False, jta_path=False, jrdb_path=False, pred_path=False, small_terrain=False,
    show_traj=False, server_mode=False, slow=False, height_debug=False, random_heading=False,
    no_virtual_display=True, render=False, init_heading=False, heading_inversion=False, adjust_root_vel=False,
    input_init_pose=False, add_noise=False, vru=False, add_proj=False,
))
