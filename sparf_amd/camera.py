"""Ray generation: the boundary feeder of the hot path.

Stays in PyTorch on purpose (north_star: pose SE(3) machinery and autograd to the pose
parameters live in PyTorch); mirrors /root/reference/source/utils/camera.py:296-416.
Unlike the reference, rays are built only for the requested pixels instead of all H*W
pixels of every image followed by an index (renderer.py:273-291) -- same values, ~100x
less work.
"""
import torch


def to_hom(X):
    return torch.cat([X, torch.ones_like(X[..., :1])], dim=-1)


def invert_pose(pose):
    """[...,3,4] rigid transform -> its inverse (camera.py Pose.invert)."""
    R, t = pose[..., :3], pose[..., 3:]
    R_inv = R.transpose(-1, -2)
    return torch.cat([R_inv, -R_inv @ t], dim=-1)


def _intr_inverse(cam_intr):
    """K^-1 of [...,3,3] intrinsics.  torch.linalg.inv synchronises with the host (LAPACK-style info check) and cannot be
    captured in a hipGraph, so a non-differentiable K is inverted in closed form -- the adjugate over the determinant, in
    float64 like the fused ray-generation kernel (csrc/ray_ops.hip) -- from the tensor's CURRENT values on every call.
    (Round 3 cached the inverse by (data_ptr, _version, shape): a freed and re-allocated intrinsics tensor at the same
    address with other values returned a stale inverse.)"""
    if cam_intr.requires_grad:
        return cam_intr.inverse()
    K = cam_intr.double()
    r0, r1, r2 = K[..., 0, :], K[..., 1, :], K[..., 2, :]
    c0, c1, c2 = torch.linalg.cross(r1, r2), torch.linalg.cross(r2, r0), torch.linalg.cross(r0, r1)
    det = (r0 * c0).sum(-1, keepdim=True)
    return (torch.stack([c0, c1, c2], dim=-1) / det[..., None]).to(cam_intr.dtype)


def img2cam(X, cam_intr):
    return X @ _intr_inverse(cam_intr).transpose(-1, -2)


def cam2world(X_cam, pose_w2c):
    return to_hom(X_cam) @ invert_pose(pose_w2c).transpose(-1, -2)


def get_center_and_ray_at_pixels(pose_w2c, pixels, intr):
    """pixels [N,2] or [B,N,2] (x,y) used as given -- no +0.5 (camera.py:384-416).
    Returns center, ray [B,N,3]; ray = R_c2w K^-1 [x,y,1] is NOT normalised."""
    B = len(pose_w2c)
    xy = pixels.unsqueeze(0).repeat(B, 1, 1) if pixels.dim() == 2 else pixels
    grid_3D = img2cam(to_hom(xy.to(pose_w2c.dtype)), intr)
    center_3D = cam2world(torch.zeros_like(grid_3D), pose_w2c)
    grid_3D = cam2world(grid_3D, pose_w2c)
    return center_3D, grid_3D - center_3D


def pixel_centers(ray_idx, W, dtype):
    """flat pixel index -> (x+0.5, y+0.5), the grid of camera.py:365-368."""
    ray_idx = ray_idx.long()
    x = (ray_idx % W).to(dtype) + 0.5
    y = torch.div(ray_idx, W, rounding_mode="floor").to(dtype) + 0.5
    return torch.stack([x, y], dim=-1)


def get_center_and_ray(pose_w2c, H, W, intr, ray_idx=None):
    """All H*W pixel centres (ray_idx None) or the selected flat indices: [N] shared by
    every image or [B,N] per image (renderer.py:277-291)."""
    B = len(pose_w2c)
    if ray_idx is None:
        ray_idx = torch.arange(H * W, device=pose_w2c.device)
    xy = pixel_centers(ray_idx, W, pose_w2c.dtype)
    if xy.dim() == 3 and xy.shape[0] != B:
        raise ValueError("per-image ray_idx must have one row per pose")
    return get_center_and_ray_at_pixels(pose_w2c, xy, intr)


def get_3D_points_from_depth(center, ray, depth, multi_samples=False):
    """camera.py:418-437 (kept for API completeness; the HIP kernel fuses it)."""
    if multi_samples:
        center, ray = center[:, :, None], ray[:, :, None]
    return center + ray * depth
