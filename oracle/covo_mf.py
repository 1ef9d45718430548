"""Threaded CPU solve of the oracle's reduced camera system — TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests).

SURVEY.md §8(d) asks for a CPU baseline that uses the box's host cores the way the reference does (Ceres SPARSE_SCHUR ->
CHOLMOD with opt threads, optimization_be.cpp:258-262, 560-565). scipy's SuperLU — the oracle's solver for the parity
goldens, kept there because it shares no code with this repository — is serial: 75 % of the round-2 baseline was one
library call on one core. This module is the CPU PORT of the product's linear solve instead: the multifrontal Cholesky
over the nested-dissection tree of covgpu_nd_plan_* (host-only plan, include/covgpu.h), fronts of one level factorised
concurrently on a thread pool with LAPACK (dpotrf / dtrsm / dsyrk through scipy, which release the GIL), extend-add in
child order. It is exact for any valid tree and is cross-checked against SuperLU in tests/test_oracle.py.
"""
from __future__ import annotations

import ctypes as C
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
from threadpoolctl import threadpool_limits

stats = {"calls": 0, "seconds": 0.0, "threads": 0, "fronts": 0, "levels": 0}
_plan_cache = {}
_problem = None   # (FlatProblem, Options) of the solve in flight: the plan needs the covisibility structure
_pool = None


def set_problem(prob, opt, threads: int):
    global _problem, _pool
    _problem = (prob, opt)
    if _pool is None or stats["threads"] != threads:
        _pool = ThreadPoolExecutor(max_workers=threads)
        stats["threads"] = threads


def _plan(D, nrow, K, ptr_a, col_a):
    prob, opt = _problem
    key = (id(prob), D, int(ptr_a[K]))
    if key in _plan_cache:
        return _plan_cache[key]
    from covins_amd import backend
    lib = backend.lib()
    h = C.c_void_p()
    s = prob.as_struct()
    rc = lib.covgpu_nd_plan_create(C.byref(opt), C.byref(s), 0, C.byref(h))
    assert rc == 0, lib.covgpu_last_error()
    info = (C.c_int64 * 16)()
    lib.covgpu_nd_plan_info(h, info)
    nn = info[0]
    parent = np.zeros(nn, np.int32); level = np.zeros(nn, np.int32)
    optr = np.zeros(nn + 1, np.int32); sptr = np.zeros(nn + 1, np.int32)
    ov = np.zeros(max(info[3], 1), np.int32); sv = np.zeros(max(info[4], 1), np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    lib.covgpu_nd_plan_arrays(h, ip(parent), ip(level), ip(optr), ip(ov), ip(sptr), ip(sv))
    lib.covgpu_nd_plan_destroy(h)

    def rows(vs):
        out = []
        for v in vs:
            kf, t = int(v) >> 1, int(v) & 1
            out.extend(range(D * kf + (6 if t else 0), D * kf + (15 if t else 6)))
        return np.array(out, np.int64)
    own = [rows(ov[optr[n]:optr[n + 1]]) for n in range(nn)]
    st = [rows(sv[sptr[n]:sptr[n + 1]]) for n in range(nn)]
    loc = [None] * nn   # position of the node's border rows inside the parent's front
    for n in range(nn):
        p = parent[n]
        if p >= 0:
            where = {int(g): i for i, g in enumerate(np.r_[own[p], st[p]])}
            loc[n] = np.array([where[int(g)] for g in st[n]], np.int64)
    levels = [np.nonzero(level == l)[0] for l in range(int(level.max()) + 1)]
    children = [[] for _ in range(nn)]
    for n in range(nn):
        if parent[n] >= 0:
            children[parent[n]].append(n)
    # where every front entry comes from in the block-CSR value array (the sparsity is the same in every iteration): a CSR of
    # 1-based source positions, sliced once per front
    nnzb = int(ptr_a[K])
    idx = sp.bsr_matrix((np.arange(1, nnzb * D * D + 1, dtype=np.float64).reshape(nnzb, D, D), col_a.copy(), ptr_a.copy()), shape=(nrow, nrow)).tocsr()
    gather = []
    for k in range(nn):
        o, s = own[k], st[k]
        m, q = len(o), len(s)
        Io = idx[o]
        blk = np.zeros((m + q, m + q), order="F")
        blk[:m, :m] = Io[:, o].toarray()
        if q:
            blk[m:, :m] = Io[:, s].toarray().T
        flat = blk.ravel(order="F")
        dst = np.nonzero(flat)[0]
        gather.append((dst, flat[dst].astype(np.int64) - 1, m + q))
    plan = dict(parent=parent, own=own, st=st, loc=loc, levels=levels, children=children, gather=gather)
    _plan_cache.clear(); _plan_cache[key] = plan
    stats["fronts"], stats["levels"] = nn, len(levels)
    return plan


def solve(n, D, K, ptr, col, blocks, rhs, x):
    """Callback of covo_set_sparse_solver: block-CSR SPD system -> x. Returns 0, or 1 if a pivot is not positive."""
    t0 = time.perf_counter()
    try:
        ptr_a = np.ctypeslib.as_array(ptr, (K + 1,))
        nnzb = int(ptr_a[K])
        vals = np.ctypeslib.as_array(blocks, (nnzb * D * D,))
        b = np.ctypeslib.as_array(rhs, (n,)).copy()
        pl = _plan(D, n, K, ptr_a, np.ctypeslib.as_array(col, (nnzb,)))
        nn = len(pl["own"])
        F = [None] * nn; R = [None] * nn; fac = [None] * nn
        bad = []

        def assemble(k):
            o = pl["own"][k]
            dst, src, f = pl["gather"][k]
            Fk = np.zeros(f * f)
            Fk[dst] = vals[src]
            F[k] = Fk.reshape(f, f, order="F")
            R[k] = np.r_[b[o], np.zeros(f - len(o))]

        def factor(k):
            Fk, r = F[k], R[k]
            m = len(pl["own"][k])
            for c in pl["children"][k]:           # extend-add in child order
                lc = pl["loc"][c]
                U, rb = fac[c][3], fac[c][4]
                Fk[np.ix_(lc, lc)] += U
                r[lc] += rb
                fac[c] = fac[c][:3]
            try:
                L11 = sla.cholesky(Fk[:m, :m], lower=True, overwrite_a=False, check_finite=False)
            except sla.LinAlgError:
                bad.append(k); return
            y1 = sla.solve_triangular(L11, r[:m], lower=True, check_finite=False)
            if Fk.shape[0] > m:
                L21 = sla.solve_triangular(L11, Fk[m:, :m].T, lower=True, check_finite=False).T
                U = Fk[m:, m:] - L21 @ L21.T
                fac[k] = (L11, L21, y1, U, r[m:] - L21 @ y1)
            else:
                fac[k] = (L11, None, y1)
            F[k] = None

        xs = np.ctypeslib.as_array(x, (n,))

        def back(k):
            L11, L21, y1 = fac[k][:3]
            o, s = pl["own"][k], pl["st"][k]
            t = y1 if L21 is None else y1 - L21.T @ xs[s]
            xs[o] = sla.solve_triangular(L11, t, lower=True, trans="T", check_finite=False)

        def run(fn, items):
            # many fronts: one per pool thread with single-threaded LAPACK; few (the top of the tree): in turn, LAPACK on all threads
            if len(items) >= 4:
                with threadpool_limits(limits=1):
                    list(_pool.map(fn, items))
            else:
                with threadpool_limits(limits=stats["threads"]):
                    for k in items:
                        fn(k)

        run(assemble, list(range(nn)))
        for lev in pl["levels"]:
            run(factor, list(lev))
            if bad:
                return 1
        for lev in pl["levels"][::-1]:
            run(back, list(lev))
        return 0
    except Exception as e:
        print("covo multifrontal solver:", repr(e))
        return 2
    finally:
        stats["calls"] += 1
        stats["seconds"] += time.perf_counter() - t0
