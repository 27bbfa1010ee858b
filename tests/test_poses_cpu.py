"""Pose / ray helpers around render_rays for --refine_pose.  When /root/reference is present (build
container) the SO(3) helper is pinned against the reference's own utils/lie_group_helper.py, loaded by
file path (it needs only torch/numpy/scipy); everywhere it is checked against closed-form properties."""
import importlib.util
import os

import pytest
import torch

from nerf_fl_amd.poses import LearnPose, get_ray_directions, get_rays, make_c2w, so3_exp

REF = "/root/reference/utils/lie_group_helper.py"


def test_so3_exp_properties():
    torch.manual_seed(0)
    r = torch.randn(16, 3)
    R = so3_exp(r)
    eye = torch.eye(3).expand(16, 3, 3)
    assert torch.allclose(R @ R.transpose(1, 2), eye, atol=1e-5)
    assert torch.allclose(torch.linalg.det(R), torch.ones(16), atol=1e-5)
    assert torch.allclose((R @ r[..., None])[..., 0], r, atol=1e-5)           # the axis is fixed
    assert torch.allclose(so3_exp(torch.zeros(3)), torch.eye(3))


@pytest.mark.skipif(not os.path.exists(REF), reason="reference not present on this machine")
def test_make_c2w_matches_reference():
    spec = importlib.util.spec_from_file_location("ref_lie", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    torch.manual_seed(1)
    for _ in range(8):
        r, t = torch.randn(3) * 0.7, torch.randn(3)
        assert torch.allclose(make_c2w(r, t), ref.make_c2w(r, t), atol=1e-6)
    r, t = torch.randn(5, 3), torch.randn(5, 3)                                # batched == per camera
    batched = make_c2w(r, t)
    for k in range(5):
        assert torch.allclose(batched[k], ref.make_c2w(r[k], t[k]), atol=1e-6)


def test_rays_and_learnable_pose_gradients():
    H, W = 6, 8
    K = torch.tensor([[10.0, 0, 4.0], [0, 10.0, 3.0], [0, 0, 1]])
    d = get_ray_directions(H, W, K)
    assert d.shape == (H, W, 3)
    assert torch.allclose(d[3, 4], torch.tensor([0.0, 0.0, -1.0]))             # principal point, no half-pixel shift
    assert torch.allclose(d[0, 0], torch.tensor([-0.4, 0.3, -1.0]))
    init = torch.eye(4).repeat(3, 1, 1)
    init[:, :3, 3] = torch.tensor([[0.0, 0, 4], [1, 0, 4], [0, 1, 4]])
    pose = LearnPose(3, True, True, init_c2w=init)
    cam = torch.tensor([0, 2, 2, 1])
    c2w = pose(cam)
    assert c2w.shape == (4, 4, 4) and torch.allclose(c2w, init[cam])           # zero delta = initial pose
    rays_o, rays_d = get_rays(d.reshape(-1, 3)[:4], c2w)
    assert torch.allclose(rays_d.norm(dim=-1), torch.ones(4), atol=1e-6)
    assert torch.allclose(rays_o, init[cam][:, :3, 3])
    (rays_o.sum() + (rays_d * torch.arange(3.0)).sum()).backward()             # gradients reach (r, t) of the used cameras
    assert pose.t.grad[0].abs().sum() > 0 and pose.r.grad[2].abs().sum() > 0


def test_dolly_path_restates_the_reference_video_path():
    """nerf_fl_amd.eval.dolly_path against the literal loop of the reference (eval.py:171-183), and the test intrinsics
    (eval.py:164-168)."""
    import math

    import numpy as np
    from nerf_fl_amd.eval import dolly_path, fov60_intrinsics
    pose = torch.tensor([[1.0, 0.0, 0.0, 0.2], [0.0, 1.0, 0.0, -0.3], [0.0, 0.0, 1.0, 1.5]])
    n = 120
    dx, dy, dz = np.linspace(0, 0.03, n), np.linspace(0, -0.1, n), np.linspace(0, 0.5, n)
    ref = np.tile(pose.numpy(), (n, 1, 1))
    for i in range(n):
        ref[i, 0, 3] += dx[i]
        ref[i, 1, 3] += dy[i]
        ref[i, 2, 3] += dz[i]
    got = dolly_path(pose, n)
    assert got.shape == (n, 3, 4)
    assert np.abs(got.numpy() - ref).max() <= 1e-6
    K = fov60_intrinsics(400, 300)
    assert abs(K[0, 0].item() - 400 / 2 / math.tan(math.pi / 6)) < 1e-3 and K[0, 2] == 200 and K[1, 2] == 150
