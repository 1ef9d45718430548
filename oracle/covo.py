"""Python loader for the CPU oracle (oracle/libcovo.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never from covins_amd/. PARITY UNPINNED — see oracle/covo_residuals.hpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from covins_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libcovo.so")
    srcs = [os.path.join(_HERE, f) for f in ("covo_solver.cpp", "covo_residuals.hpp", "covo_math.hpp")]
    srcs.append(os.path.join(_HERE, "..", "include", "covgpu.h"))
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcovo.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        capi.declare(_LIB, "covo_")
        dp = capi._dp
        OP, PP = C.POINTER(capi.Options), C.POINTER(capi.ProblemStruct)
        L = _LIB
        L.covo_num_threads.restype = C.c_int
        L.covo_set_num_threads.argtypes = [C.c_int]
        L.covo_schur.argtypes = [OP, PP, C.c_int, C.c_double, dp, dp, dp]
        L.covo_dense_step.argtypes = [OP, PP, C.c_double, dp, dp]
        L.covo_schur_step.argtypes = [OP, PP, C.c_double, dp, dp]
        L.covo_solve_reduced.argtypes = [C.c_int, dp, dp, dp]
        L.covo_cost.argtypes = [OP, PP, C.c_int]
        L.covo_cost.restype = C.c_double
        L.covo_pose_plus.argtypes = [dp, dp, dp]
        L.covo_reproj_residual.argtypes = [dp, dp, dp, dp, dp, C.c_int, dp, C.c_double, dp, dp, dp]
        L.covo_between_residual.argtypes = [dp, dp, dp, dp, dp, dp, dp]
        L.covo_imu_residual.argtypes = [dp, dp, C.c_int, dp, dp, dp, dp, dp, dp, dp, C.c_int, dp, dp, dp]
        for n in ("covo_pose_plus", "covo_reproj_residual", "covo_between_residual", "covo_imu_residual"):
            getattr(L, n).restype = None
    return _LIB


def default_options(**kw) -> capi.Options:
    o = capi.Options()
    lib().covo_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _d(a):
    return capi.dptr(a)


def gba_solve(prob: capi.FlatProblem, opt: capi.Options, pgo: bool = False):
    """Solves on a copy; returns (solved FlatProblem, Result)."""
    q = prob.copy()
    s = q.as_struct()
    r = capi.Result()
    fn = lib().covo_pgo_solve if pgo else lib().covo_gba_solve
    rc = fn(C.byref(opt), C.byref(s), C.byref(r))
    assert rc == 0, rc
    return q, r


def linearize_reprojection(prob, opt):
    O = prob.O
    r, Jp, Jl, c = np.zeros((O, 2)), np.zeros((O, 12)), np.zeros((O, 6)), np.zeros(O)
    s = prob.as_struct()
    assert lib().covo_linearize_reprojection(C.byref(opt), C.byref(s), _d(r), _d(Jp), _d(Jl), _d(c)) == 0
    return r, Jp, Jl, c


def preintegrate(prob, opt):
    I = prob.I
    d, J, P = np.zeros((I, 11)), np.zeros((I, 225)), np.zeros((I, 225))
    s = prob.as_struct()
    assert lib().covo_preintegrate(C.byref(opt), C.byref(s), _d(d), _d(J), _d(P)) == 0
    return d, J, P


def linearize_imu(prob, opt):
    I = prob.I
    r, J = np.zeros((I, 15)), np.zeros((I, 450))
    s = prob.as_struct()
    assert lib().covo_linearize_imu(C.byref(opt), C.byref(s), _d(r), _d(J)) == 0
    return r, J


def linearize_between(prob, opt):
    E = prob.E
    r, J, c = np.zeros((E, 6)), np.zeros((E, 72)), np.zeros(E)
    s = prob.as_struct()
    assert lib().covo_linearize_between(C.byref(opt), C.byref(s), _d(r), _d(J), _d(c)) == 0
    return r, J, c


def schur(prob, opt, mu, pgo=False):
    n = (6 if (pgo or opt.visual_only) else 15) * prob.K
    S, b, c = np.zeros((n, n)), np.zeros(n), np.zeros(1)
    s = prob.as_struct()
    assert lib().covo_schur(C.byref(opt), C.byref(s), int(pgo), float(mu), _d(S), _d(b), _d(c)) == 0
    return S, b, float(c[0])


def step(prob, opt, mu, dense: bool):
    n = (6 if opt.visual_only else 15) * prob.K
    dp_, dl = np.zeros(n), np.zeros((prob.L, 3))
    s = prob.as_struct()
    fn = lib().covo_dense_step if dense else lib().covo_schur_step
    assert fn(C.byref(opt), C.byref(s), float(mu), _d(dp_), _d(dl)) == 0
    return dp_, dl


def solve_reduced(S, b):
    n = b.shape[0]
    S = np.ascontiguousarray(S, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(n)
    rc = lib().covo_solve_reduced(n, _d(S), _d(b), _d(x))
    return rc, x


def cost(prob, opt, pgo=False):
    s = prob.as_struct()
    return lib().covo_cost(C.byref(opt), C.byref(s), int(pgo))


def residual_norms(prob, opt):
    out = np.zeros(prob.O)
    s = prob.as_struct()
    assert lib().covo_reprojection_residual_norms(C.byref(opt), C.byref(s), _d(out)) == 0
    return out


def pgo_reanchor(pose_old, pose_new, velocity, ref_kf, lm_pos):
    pose_old = np.ascontiguousarray(pose_old, dtype=np.float64)
    pose_new = np.ascontiguousarray(pose_new, dtype=np.float64)
    vel = None if velocity is None else np.array(velocity, dtype=np.float64, order="C")
    lm = np.array(lm_pos, dtype=np.float64, order="C")
    ref = np.ascontiguousarray(ref_kf, dtype=np.int32)
    lib().covo_pgo_reanchor(pose_old.shape[0], _d(pose_old), _d(pose_new), _d(vel), lm.shape[0], capi.iptr(ref), _d(lm))
    return vel, lm
