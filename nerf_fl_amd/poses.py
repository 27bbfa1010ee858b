"""Learnable camera poses and ray generation for `--refine_pose` training (the callers either side of
render_rays when poses are optimised: reference models/poses.py, utils/lie_group_helper.py:50-84,
datasets/ray_utils.py:5-55, train.py:86-98).

Plain PyTorch on purpose: this is O(cameras) / O(rays) 3x3 algebra whose only job is to turn the
gradient w.r.t. rays, which the HIP backward produces, into gradients of (r, t) through autograd.
Unlike the fork's forward (a Python loop over rays with an int() sync each, train.py:92-98) everything
is batched over the rays of the step and stays on the device.
"""
import torch
from torch import nn

__all__ = ["so3_exp", "make_c2w", "LearnPose", "get_ray_directions", "get_rays"]


def _skew(v):
    z = torch.zeros_like(v[..., 0])
    return torch.stack([torch.stack([z, -v[..., 2], v[..., 1]], -1),
                        torch.stack([v[..., 2], z, -v[..., 0]], -1),
                        torch.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def so3_exp(r):
    """Rodrigues: axis-angle (..., 3) -> rotation (..., 3, 3), with the reference's +1e-15 in the norm
    (lie_group_helper.py:63-72)."""
    K = _skew(r)
    n = r.norm(dim=-1) + 1e-15
    a = (torch.sin(n) / n)[..., None, None]
    b = ((1 - torch.cos(n)) / n ** 2)[..., None, None]
    eye = torch.eye(3, dtype=r.dtype, device=r.device).expand(K.shape)
    return eye + a * K + b * (K @ K)


def make_c2w(r, t):
    """(..., 3), (..., 3) -> (..., 4, 4) (lie_group_helper.py:75-84)."""
    top = torch.cat([so3_exp(r), t[..., None]], -1)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=r.dtype, device=r.device).expand(*top.shape[:-2], 1, 4)
    return torch.cat([top, bottom], -2)


class LearnPose(nn.Module):
    """Per-camera (r, t) delta composed with the initial pose (reference models/poses.py:9-34);
    `forward(cam_id)` accepts a scalar id (reference behaviour) or a tensor of ids (batched)."""

    def __init__(self, num_cams, learn_R, learn_t, init_c2w=None):
        super().__init__()
        self.num_cams = num_cams
        self.init_c2w = nn.Parameter(init_c2w, requires_grad=False) if init_c2w is not None else None
        self.r = nn.Parameter(torch.zeros(num_cams, 3), requires_grad=learn_R)
        self.t = nn.Parameter(torch.zeros(num_cams, 3), requires_grad=learn_t)

    def forward(self, cam_id):
        c2w = make_c2w(self.r[cam_id], self.t[cam_id])
        if self.init_c2w is not None:
            c2w = c2w @ self.init_c2w[cam_id]
        return c2w


def get_ray_directions(H, W, K, device=None):
    """(H, W, 3) camera-frame directions, no half-pixel centring (ray_utils.py:5-26)."""
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=device),
                          torch.arange(W, dtype=torch.float32, device=device), indexing="ij")
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    return torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)


def get_rays(directions, c2w):
    """directions (B, 3) camera frame, c2w (B, 3|4, 4) or (3|4, 4) -> unit world directions and origins
    (ray_utils.py:29-55: rotate, normalise, origin = translation column)."""
    if c2w.dim() < 3:
        c2w = c2w[None]
    directions = directions.reshape(-1, 3)
    R, t = c2w[:, :3, :3], c2w[:, :3, 3]
    rays_d = (directions[:, None, :] @ R.transpose(1, 2))[:, 0, :]
    rays_d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    rays_o = t.expand(rays_d.shape)
    return rays_o, rays_d
