"""Camera helpers -- counterpart of reference keras_nerf/data/utils.py:5-63 (host side, float32 NumPy)."""
from __future__ import annotations

import numpy as np


def get_focal_from_fov(field_of_view: float, width: int) -> float:
    """data/utils.py:5-16: 0.5 * width / tan(0.5 * fov), in float32 like the reference"""
    f = np.float32
    return float(f(0.5) * f(width) / np.tan(f(0.5) * f(field_of_view)))


def get_translation_t(t):
    return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]], np.float32)


def get_rotation_phi(phi):
    c, s = np.cos(np.float32(phi)), np.sin(np.float32(phi))
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], np.float32)


def get_rotation_theta(theta):
    c, s = np.cos(np.float32(theta)), np.sin(np.float32(theta))
    return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], np.float32)


def pose_spherical(theta, phi, t):
    """data/utils.py:52-63: camera-to-world matrix for (theta deg, phi deg, radius t)"""
    c2w = get_translation_t(t)
    c2w = get_rotation_phi(phi / 180.0 * np.pi) @ c2w
    c2w = get_rotation_theta(theta / 180.0 * np.pi) @ c2w
    return (np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32) @ c2w).astype(np.float32)
