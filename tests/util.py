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


def landmark_parity(sol_lm, ref_prob, ref_lm=None, idx=None):
    """Landmark criterion of the parity tests. `ref_prob` holds the ORACLE's solution.
    Every landmark whose 3x3 Hessian H_ll (oracle, at the oracle's solution) has cond < 1e6 must agree within 1e-6 m;
    the others (near-degenerate parallax: H_ll almost singular along the viewing ray, so two correct solvers that round
    differently land far apart along that ray at identical cost; or H_ll = 0: a landmark that a Gauss-Newton step threw
    behind its cameras, where the residual is defined as zero) within 1e-4 in whitened units
    sqrt(d^T H_ll d), i.e. no reprojection moves by 1e-4 sigma. The cut is 1e6, not the 1e8 first proposed: the oracle
    against ITSELF with a different linear solver (dense Cholesky vs SuperLU, poses equal to 2e-11 m) differs by 3.6e-8 m
    at most below cond 1e6 but by 1.3e-6 m between 1e6 and 1e8 (tests/test_oracle.py, `small` map). Returns (number of ill-conditioned landmarks, max
    well-conditioned difference, max whitened difference). `idx`: compare only these landmarks (strided goldens)."""
    from oracle import covo
    H = covo.landmark_hessians(ref_prob, covo.default_options())
    ref = ref_prob.lm_pos if ref_lm is None else ref_lm
    if idx is not None:
        H = H[idx]
    d = np.asarray(sol_lm) - np.asarray(ref)
    cond = np.linalg.cond(H)
    good = cond < 1e6
    dist = np.linalg.norm(d, axis=1)
    white = np.sqrt(np.einsum("li,lij,lj->l", d, H, d).clip(0))
    if not good.any() and len(cond):  # diagnostic for an unexplained all-ill result seen once on a GPU box
        print(f"landmark_parity: no well-conditioned landmark: |H|max={np.abs(H).max():.3e} cond min={np.nanmin(cond):.3e} "
              f"nan={int(np.isnan(cond).sum())} |d|max={dist.max():.3e} L={len(cond)}")
    return int((~good).sum()), float(dist[good].max() if good.any() else 0.0), float(white.max() if len(white) else 0.0)
