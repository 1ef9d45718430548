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
_SOLVER_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libcovo.so")
    srcs = [os.path.join(_HERE, f) for f in ("covo_solver.cpp", "covo_relpose.cpp", "covo_residuals.hpp", "covo_math.hpp")]
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
        L.covo_set_sparse_solver.argtypes = [_SOLVER_FN, C.c_int]
        L.covo_set_sparse_solver.restype = None
        ip = capi._ip
        L.covo_schur_sparse.argtypes = [OP, PP, C.c_int, C.c_double, ip, ip, dp, dp, dp]
        L.covo_landmark_hessians.argtypes = [OP, PP, dp]
        L.covo_relpose.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp, dp, C.c_int, dp, C.c_int, C.c_double, C.c_int, dp, C.POINTER(C.c_ubyte)]
        L.covo_relpose_residual.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_double, dp, C.c_int, dp, C.c_int, dp, dp]
        L.covo_relpose_residual.restype = None
    return _LIB


# ---------------------------------------------------------------- sparse linear solver for BASELINE-size problems
# The oracle's own dense Cholesky handles the small test problems. For the reduced systems of the BASELINE
# configurations (15 K up to 33k unknowns) the block-CSR system is handed to scipy's SuperLU: a library solver that
# shares no code with either Cholesky written in this repository. No pivoting + symmetric mode makes the U diagonal
# the pivots of L D L^T, so "all pivots positive" is the positive-definiteness test Ceres' CHOLMOD path relies on.
_solver_keepalive = None
solver_stats = {"calls": 0, "seconds": 0.0, "nnz_factor": 0}


def _superlu_solve(n, D, K, ptr, col, blocks, rhs, x):
    import time
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    t0 = time.perf_counter()
    try:
        ptr_a = np.ctypeslib.as_array(ptr, (K + 1,))
        nnzb = int(ptr_a[K])
        col_a = np.ctypeslib.as_array(col, (nnzb,))
        blk = np.ctypeslib.as_array(blocks, (nnzb, D, D))
        A = sp.bsr_matrix((blk, col_a, ptr_a), shape=(n, n)).tocsc()
        A.eliminate_zeros()
        lu = spla.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
        if not (lu.U.diagonal() > 0.0).all():
            return 1
        b = np.ctypeslib.as_array(rhs, (n,))
        np.ctypeslib.as_array(x, (n,))[:] = lu.solve(b)
        solver_stats["nnz_factor"] = int(lu.L.nnz)
        return 0
    except Exception as e:  # singular matrix etc.: treated like a failed Cholesky
        print("covo sparse solver:", e)
        return 2
    finally:
        solver_stats["calls"] += 1
        solver_stats["seconds"] += time.perf_counter() - t0


def use_sparse_solver(min_n: int = 3000, enable: bool = True, kind: str = "superlu") -> None:
    """Route reduced systems with >= min_n unknowns through scipy's SuperLU (smaller ones keep the dense Cholesky).
    kind="multifrontal": the threaded CPU port of the product's multifrontal solve (oracle/covo_mf.py) — the cpu_baseline
    leg of bench.py; call covo_mf.set_problem(prob, opt, threads) before the solve."""
    global _solver_keepalive
    if enable and kind == "multifrontal":
        from oracle import covo_mf
        _solver_keepalive = _SOLVER_FN(covo_mf.solve)
        lib().covo_set_sparse_solver(_solver_keepalive, int(min_n))
    elif enable:
        _solver_keepalive = _SOLVER_FN(_superlu_solve)
        lib().covo_set_sparse_solver(_solver_keepalive, int(min_n))
    else:
        lib().covo_set_sparse_solver(C.cast(None, _SOLVER_FN), 0)
        _solver_keepalive = None


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


def schur_sparse(prob, opt, mu, pgo=False):
    """Block-CSR reduced system (ptr, col, blocks[nnzb, D, D]), right-hand side and cost."""
    D = 6 if (pgo or opt.visual_only) else 15
    s = prob.as_struct()
    nnzb = lib().covo_schur_sparse(C.byref(opt), C.byref(s), int(pgo), float(mu), None, None, None, None, None)
    ptr, col = np.zeros(prob.K + 1, np.int32), np.zeros(nnzb, np.int32)
    blocks, b, c = np.zeros((nnzb, D, D)), np.zeros(D * prob.K), np.zeros(1)
    lib().covo_schur_sparse(C.byref(opt), C.byref(s), int(pgo), float(mu), capi.iptr(ptr), capi.iptr(col), _d(blocks), _d(b), _d(c))
    return ptr, col, blocks, b, float(c[0])


def landmark_hessians(prob, opt):
    H = np.zeros((prob.L, 9))
    s = prob.as_struct()
    assert lib().covo_landmark_hessians(C.byref(opt), C.byref(s), _d(H)) == 0
    return H.reshape(-1, 3, 3)


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


def relpose(pB, pA, kpA, kpB, sigA, sigB, camA, distA, camB, distB, T_ab, th_outlier=2.0, min_inliers=12):
    """Optimization::OptimizeRelativePose for ONE keyframe pair (optimization_be.cpp:620-831): returns (inliers, T_ab, outlier flags)."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    pB, pA, kpA, kpB, sigA, sigB, camA, camB = map(f, (pB, pA, kpA, kpB, sigA, sigB, camA, camB))
    T = np.array(T_ab, dtype=np.float64)
    out = np.zeros(len(sigA), np.uint8)
    n = lib().covo_relpose(len(sigA), _d(pB), _d(pA), _d(kpA), _d(kpB), _d(sigA), _d(sigB), _d(camA), int(distA), _d(camB), int(distB),
                           float(th_outlier), int(min_inliers), _d(T), out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return int(n), T, out.astype(bool)


def relpose_residual(T_ab, pB, pA, kpA, kpB, sigA, sigB, camA, distA, camB, distB, jac=True):
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    r, J = np.zeros(4), np.zeros((4, 6))
    lib().covo_relpose_residual(_d(f(T_ab)), _d(f(pB)), _d(f(pA)), _d(f(kpA)), _d(f(kpB)), float(sigA), float(sigB), _d(f(camA)), int(distA),
                                _d(f(camB)), int(distB), _d(r), _d(J) if jac else None)
    return r, J
