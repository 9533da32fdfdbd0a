"""Camera helpers (host side, float32 NumPy): the public names and conventions of the reference's keras_nerf/data/utils.py
(focal length from the field of view at :5-16, the spherical camera pose at :52-63) on a small homogeneous-matrix toolkit."""
from __future__ import annotations

import numpy as np

_F = np.float32


def get_focal_from_fov(field_of_view: float, width: int) -> float:
    """focal = (width / 2) / tan(fov / 2), evaluated in float32 as the reference does"""
    half_width, half_fov = _F(0.5) * _F(width), _F(0.5) * _F(field_of_view)
    return float(half_width / np.tan(half_fov))


def _homogeneous(rotation=None, translation=(0.0, 0.0, 0.0)) -> np.ndarray:
    m = np.eye(4, dtype=_F)
    if rotation is not None:
        m[:3, :3] = rotation
    m[:3, 3] = translation
    return m


def _axis_rotation(axis: int, angle) -> np.ndarray:
    """3x3 rotation that leaves coordinate `axis` alone; on the two remaining coordinates (in increasing order) it is
    [[c, -s], [s, c]] -- the convention of both of the reference's rotation helpers"""
    c, s = np.cos(_F(angle)), np.sin(_F(angle))
    i, j = [k for k in range(3) if k != axis]
    r = np.eye(3, dtype=_F)
    r[i, i] = r[j, j] = c
    r[i, j], r[j, i] = -s, s
    return r


def get_translation_t(t):
    """camera pushed back by t along +z"""
    return _homogeneous(translation=(0.0, 0.0, t))


def get_rotation_phi(phi):
    """elevation: rotation about x (rows y,z = [c -s; s c])"""
    return _homogeneous(_axis_rotation(0, phi))


def get_rotation_theta(theta):
    """azimuth: rotation about y with the reference's sign (rows x,z = [c -s; s c])"""
    return _homogeneous(_axis_rotation(1, theta))


# world-from-blender axis swap applied last: x -> -x, y <-> z
_SWAP = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], _F)


def pose_spherical(theta, phi, t):
    """camera-to-world matrix of a camera at radius t looking at the origin, azimuth theta and elevation phi in degrees"""
    pose = get_translation_t(t)
    for rot, deg in ((get_rotation_phi, phi), (get_rotation_theta, theta)):
        pose = rot(deg / 180.0 * np.pi) @ pose
    return (_SWAP @ pose).astype(_F)
