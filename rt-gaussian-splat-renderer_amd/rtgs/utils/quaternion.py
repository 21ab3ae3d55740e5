"""Quaternion utils, scalar-last (x, y, z, w) -- the device mirror of the reference's
src/rtgs/utils/quaternion.py.

In the reference these are Taichi @ti.func (device) functions.  Here mul / conj / rot_vec3 /
as_rotation_mat3 run the HIP kernels behind ``rtgs_quat_op`` (include/rtgs_hip.h) on batches:
inputs are array-likes or torch tensors of shape (..., 4) / (..., 3); results are torch
tensors on the HIP device.  inv / from_axis_angle / as_axis_angle / as_rotation_mat4 are off
the render path and are composed from those device ops with torch.
"""

from __future__ import annotations

import math

import torch

from .. import _native as nat


def _dev(x, last):
    t = torch.as_tensor(x, dtype=torch.float32)
    if t.shape[-1] != last:
        raise ValueError(f"expected last dimension {last}, got {tuple(t.shape)}")
    return t.to(nat.device()).contiguous()


def _pad4(v):
    return torch.cat([v, torch.zeros_like(v[..., :1])], dim=-1).contiguous()


def _op(op, p, q, out_cols):
    shape = p.shape[:-1]
    n = math.prod(shape)
    if n == 0:
        return torch.empty((0, out_cols), dtype=torch.float32, device=p.device)
    p2 = p.reshape(n, 4).contiguous()
    q2 = None if q is None else q.expand(p.shape).reshape(n, 4).contiguous()
    out = torch.empty((n, out_cols), dtype=torch.float32, device=p.device)
    nat.check(nat.lib().rtgs_quat_op(op, n, nat.ptr(p2), nat.ptr(q2), nat.ptr(out), nat.stream()),
              "rtgs_quat_op")
    return out


def mul(p, q):
    """Quaternion product pq (utils/quaternion.py:8-23)."""
    p, q = _dev(p, 4), _dev(q, 4)
    p, q = torch.broadcast_tensors(p, q)
    p = p.contiguous()
    return _op(0, p, q.contiguous(), 4).reshape(p.shape)


def conj(q):
    """Complex conjugate (utils/quaternion.py:26-35)."""
    q = _dev(q, 4)
    return _op(1, q, None, 4).reshape(q.shape)


def inv(q):
    """conj(q) / |q| -- literal (utils/quaternion.py:38-47); an inverse only for unit q."""
    q = _dev(q, 4)
    return conj(q) / torch.linalg.vector_norm(q, dim=-1, keepdim=True)


def from_axis_angle(v):
    """utils/quaternion.py:50-65."""
    v = _dev(v, 3)
    theta = torch.linalg.vector_norm(v, dim=-1, keepdim=True)
    axis = torch.where(theta > 0, v / torch.where(theta > 0, theta, torch.ones_like(theta))
                       * torch.sin(theta / 2), v)
    return torch.cat([axis, torch.cos(theta / 2)], dim=-1)


def as_axis_angle(q):
    """utils/quaternion.py:68-81."""
    q = _dev(q, 4)
    theta = torch.acos(q[..., 3:4]) * 2
    norm = torch.linalg.vector_norm(q[..., :3], dim=-1, keepdim=True)
    safe = torch.where(norm > 0, norm, torch.ones_like(norm))
    return torch.where(norm > 0, q[..., :3] / safe * theta, torch.zeros_like(q[..., :3]))


def rot_vec3(q, v):
    """Rotate v by q as q v q* (utils/quaternion.py:84-96)."""
    q, v = _dev(q, 4), _dev(v, 3)
    shape = torch.broadcast_shapes(q.shape[:-1], v.shape[:-1])
    q = q.expand(shape + (4,)).contiguous()
    v4 = _pad4(v.expand(shape + (3,)).contiguous())
    return _op(2, q, v4, 3).reshape(shape + (3,))


def as_rotation_mat3(q):
    """3x3 matrix whose column k is q e_k q* (utils/quaternion.py:99-121)."""
    q = _dev(q, 4)
    return _op(3, q, None, 9).reshape(q.shape[:-1] + (3, 3))


def as_rotation_mat4(q):
    """The same rotation in homogeneous coordinates (utils/quaternion.py:124-147)."""
    m3 = as_rotation_mat3(q)
    m = torch.zeros(m3.shape[:-2] + (4, 4), dtype=m3.dtype, device=m3.device)
    m[..., :3, :3] = m3
    m[..., 3, 3] = 1
    return m
