"""Shared helpers for the parity tests."""
import numpy as np

from covins_amd import mapdata


def truth_map(m):
    t = m.copy()
    t.kf_pose = m.truth["kf_pose"].copy()
    t.kf_pose_vio = m.truth["kf_pose"].copy()
    t.kf_velocity = m.truth["kf_velocity"].copy()
    t.kf_bias_a = m.truth["kf_bias_a"].copy()
    t.kf_bias_g = m.truth["kf_bias_g"].copy()
    t.lm_pos = m.truth["lm_pos"].copy()
    return t


def rot_angle(qa, qb):
    """Angle between unit quaternions [N,4] (x,y,z,w), radians."""
    d = np.abs(np.sum(qa * qb, axis=1)).clip(0, 1)
    return 2 * np.arccos(d)


def rel_err(a, b, floor=1e-300):
    a, b = np.asarray(a, float), np.asarray(b, float)
    scale = max(np.abs(b).max(), floor)
    return np.abs(a - b).max() / scale
