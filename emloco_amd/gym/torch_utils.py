"""Host-side torch helpers with the reference's names and conventions (quaternions xyzw).

Used only off the per-step hot path (reset, trajectory generation, motion library); the per-step
versions live in csrc/dev_math.h.  Formulas follow isaacgym/python/isaacgym/torch_utils.py and
pacer/pacer/utils/torch_utils.py (cited per function) so host and device agree.
"""
import numpy as np
import torch


def to_torch(x, dtype=torch.float, device="cuda:0", requires_grad=False):
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)


def normalize(x, eps: float = 1e-9):                     # isaacgym torch_utils.py:44-46
    return x / x.norm(p=2, dim=-1).clamp(min=eps).unsqueeze(-1)


def quat_mul(a, b):                                       # isaacgym torch_utils.py:19-41 (8-multiply form)
    shape = a.shape
    a, b = a.reshape(-1, 4), b.reshape(-1, 4)
    x1, y1, z1, w1 = a.unbind(-1)
    x2, y2, z2, w2 = b.unbind(-1)
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return torch.stack([x, y, z, w], dim=-1).view(shape)


def quat_conjugate(a):
    return torch.cat((-a[..., :3], a[..., 3:]), dim=-1)


def quat_apply(a, b):                                     # isaacgym torch_utils.py:49-56
    shape = b.shape
    a, b = a.reshape(-1, 4), b.reshape(-1, 3)
    xyz = a[:, :3]
    t = torch.cross(xyz, b, dim=-1) * 2
    return (b + a[:, 3:] * t + torch.cross(xyz, t, dim=-1)).view(shape)


def quat_rotate(q, v):                                    # pacer/utils/torch_utils.py:14-24 (my_quat_rotate)
    q_w = q[:, -1]
    q_vec = q[:, :3]
    a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * (q_vec * v).sum(-1, keepdim=True) * 2.0
    return a + b + c


my_quat_rotate = quat_rotate


def quat_from_angle_axis(angle, axis):                    # isaacgym torch_utils.py:96-101
    theta = (angle / 2).unsqueeze(-1)
    xyz = normalize(axis) * theta.sin()
    return normalize(torch.cat([xyz, theta.cos()], dim=-1))


def normalize_angle(x):
    return torch.atan2(torch.sin(x), torch.cos(x))


def calc_heading(q):                                      # pacer/utils/torch_utils.py:137-149
    ref = torch.zeros_like(q[..., 0:3])
    ref[..., 0] = 1
    rot = quat_rotate(q.reshape(-1, 4), ref.reshape(-1, 3)).view(ref.shape)
    return torch.atan2(rot[..., 1], rot[..., 0])


def calc_heading_quat(q):
    axis = torch.zeros_like(q[..., 0:3])
    axis[..., 2] = 1
    return quat_from_angle_axis(calc_heading(q), axis)


def calc_heading_quat_inv(q):
    axis = torch.zeros_like(q[..., 0:3])
    axis[..., 2] = 1
    return quat_from_angle_axis(-calc_heading(q), axis)


def quat_to_angle_axis(q):                                # pacer/utils/torch_utils.py:26-47
    sin_theta = torch.sqrt(1 - q[..., 3] * q[..., 3])
    angle = normalize_angle(2 * torch.acos(q[..., 3]))
    axis = q[..., 0:3] / sin_theta.unsqueeze(-1)
    mask = torch.abs(sin_theta) > 1e-5
    default_axis = torch.zeros_like(axis)
    default_axis[..., -1] = 1
    angle = torch.where(mask, angle, torch.zeros_like(angle))
    axis = torch.where(mask.unsqueeze(-1), axis, default_axis)
    return angle, axis


def quat_to_exp_map(q):
    angle, axis = quat_to_angle_axis(q)
    return angle.unsqueeze(-1) * axis


def exp_map_to_quat(exp_map):                             # pacer/utils/torch_utils.py:88-111
    angle = torch.norm(exp_map, dim=-1)
    axis = exp_map / angle.unsqueeze(-1)
    angle = normalize_angle(angle)
    default_axis = torch.zeros_like(exp_map)
    default_axis[..., -1] = 1
    mask = torch.abs(angle) > 1e-5
    angle = torch.where(mask, angle, torch.zeros_like(angle))
    axis = torch.where(mask.unsqueeze(-1), axis, default_axis)
    return quat_from_angle_axis(angle, axis)


def slerp(q0, q1, t):                                     # pacer/utils/torch_utils.py:113-135
    cos_half = torch.sum(q0 * q1, dim=-1)
    q1 = torch.where((cos_half < 0).unsqueeze(-1), -q1, q1)
    cos_half = torch.abs(cos_half).unsqueeze(-1)
    half = torch.acos(cos_half)
    sin_half = torch.sqrt(1.0 - cos_half * cos_half)
    ra = torch.sin((1 - t) * half) / sin_half
    rb = torch.sin(t * half) / sin_half
    out = ra * q0 + rb * q1
    out = torch.where(torch.abs(sin_half) < 0.001, 0.5 * q0 + 0.5 * q1, out)
    return torch.where(torch.abs(cos_half) >= 1, q0, out)


def torch_rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


def get_axis_params(value, axis_idx, x_value=0.0, dtype=float, n_dims=3):
    zs = np.zeros((n_dims,))
    zs[axis_idx] = 1.0
    params = np.where(zs == 1.0, value, zs)
    params[0] = x_value
    return list(params.astype(dtype))


def quat_apply_yaw(quat, vec):                            # humanoid_pedestrain_terrain.py:1533-1538
    quat_yaw = quat.clone().view(-1, 4)
    quat_yaw[:, :2] = 0.0
    return quat_apply(normalize(quat_yaw), vec)
