"""Camera-frame geometry of the KITTI labels: box corners, projection with a 3x4 camera matrix.

Host-side (numpy) helpers behind ``utils.data_classes``; same names, argument meaning and corner order as the
reference's ``utils/geometry_ops.py`` (points_cam2img :47-93, corners_nd :96-124, rotation_3d_in_axis :127-166,
center_to_corner_box3d :169-193, view_points :196-211), pinned against it by tests/golden/kitti_objects.npz.
"""
from typing import Sequence, Union

import numpy as np

# unit-cube corner pattern in the reference's order (x0y0z0, x0y0z1, x0y1z1, x0y1z0, x1y0z0, x1y0z1, x1y1z1, x1y1z0)
_CUBE = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 0, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0]], dtype=np.float64)


def _as_4x4(mat: np.ndarray) -> np.ndarray:
    m = np.eye(4, dtype=np.asarray(mat).dtype if np.asarray(mat).dtype.kind == 'f' else np.float64)
    a = np.asarray(mat)
    if a.ndim != 2 or a.shape[0] > 4 or a.shape[1] > 4:
        raise ValueError("projection matrix must be at most 4x4, got %s" % (a.shape,))
    m[:a.shape[0], :a.shape[1]] = a
    return m


def points_cam2img(points_3d: np.ndarray, proj_mat: np.ndarray, with_depth: bool = False, get_as_tensor: bool = False):
    """(N,3) camera-frame points -> (N,2) pixel coordinates (or (N,3) with the depth appended)."""
    pts = np.asarray(points_3d)
    P = _as_4x4(proj_mat)
    hom = np.concatenate([pts, np.ones(pts.shape[:-1] + (1,))], axis=-1) @ P.T
    uv = hom[..., :2] / hom[..., 2:3]
    out = np.concatenate([uv, hom[..., 2:3]], axis=-1) if with_depth else uv
    if get_as_tensor:
        import torch
        return torch.from_numpy(out)
    return out


def corners_nd(dims: np.ndarray, origin: Union[float, Sequence[float]] = 0.5) -> np.ndarray:
    """(N,3) box sizes -> (N,8,3) corner offsets relative to ``origin`` (fractions of the size)."""
    dims = np.asarray(dims)
    if dims.shape[1] != 3:
        raise NotImplementedError("only 3-D boxes are used by the MonoCon labels")
    pattern = _CUBE.astype(dims.dtype) - np.asarray(origin, dtype=dims.dtype)
    return dims[:, None, :] * pattern[None, :, :]


def rotation_3d_in_axis(points: np.ndarray, angles: np.ndarray, axis: int = 0, get_as_tensor: bool = False):
    """rotate (N,P,3) points by (N,) angles about the given axis (1 = the camera's vertical axis)."""
    s, c = np.sin(angles)[:, None], np.cos(angles)[:, None]
    x, y, z = points[..., 0], points[..., 1], points[..., 2]
    if axis == 1:
        out = np.stack([x * c + z * s, y + 0 * c, -x * s + z * c], axis=-1)
    elif axis in (2, -1):
        out = np.stack([x * c + y * s, -x * s + y * c, z + 0 * c], axis=-1)
    elif axis == 0:
        out = np.stack([z + 0 * c, x * c + y * s, -x * s + y * c], axis=-1)
    else:
        raise ValueError('axis should in range')
    if get_as_tensor:
        import torch
        return torch.from_numpy(out)
    return out


def center_to_corner_box3d(centers: np.ndarray, dims: np.ndarray, angles: np.ndarray = None, origin=(0.5, 1.0, 0.5),
                           axis: int = 1) -> np.ndarray:
    """KITTI (location, (l,h,w), rotation_y) -> (N,8,3) corners in the camera frame."""
    corners = corners_nd(dims, origin=origin)
    if angles is not None:
        corners = rotation_3d_in_axis(corners, np.asarray(angles), axis=axis)
    return corners + np.asarray(centers).reshape(-1, 1, 3)


def view_points(points: np.ndarray, view: np.ndarray, normalize: bool) -> np.ndarray:
    """(3,N) points through a (<=4 x <=4) view matrix; ``normalize`` divides by the third row."""
    pts = np.asarray(points)
    if pts.shape[0] != 3:
        raise ValueError("points must be (3, N)")
    V = _as_4x4(np.asarray(view, dtype=np.float64))
    out = (V @ np.concatenate([pts, np.ones((1, pts.shape[1]))]))[:3]
    return out / out[2:3] if normalize else out
