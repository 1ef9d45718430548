"""ctypes binding of the product library covins_amd/libcovgpu.so (C ABI: include/covgpu.h).

There is NO CPU fallback: if the HIP extension is missing or no MI355X is visible, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

from . import capi
from .capi import FlatProblem, Options, Result, dptr, iptr

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcovgpu.so")
if os.environ.get("COVGPU_LIBRARY"):   # dev aid: another build of the library (compile-time A/B, tools/gpu_ab.sh)
    _SO = os.environ["COVGPU_LIBRARY"]
_LIB: Optional[C.CDLL] = None


class CovGpuError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compiles the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return _SO


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise CovGpuError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(libcovgpu has no CPU fallback)")
        _LIB = C.CDLL(_SO)
        capi.declare(_LIB, "covgpu_")
        OP, PP = C.POINTER(Options), C.POINTER(capi.ProblemStruct)
        _LIB.covgpu_upload_pgo.argtypes = [C.c_void_p, OP, PP]
        _LIB.covgpu_schur_pgo.argtypes = [C.c_void_p, OP, PP, C.c_double, capi._dp, capi._dp, capi._dp]
        _LIB.covgpu_set_profiling.argtypes = [C.c_void_p, C.c_int]
        _LIB.covgpu_set_profiling.restype = None
        _LIB.covgpu_get_profile.argtypes = [C.c_void_p, capi._dp]
        _LIB.covgpu_get_profile.restype = None
        _LIB.covgpu_gba_two_round.argtypes = [C.c_void_p, OP, PP, C.POINTER(capi.TwoRound), capi._bp, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                              C.POINTER(capi.Result), C.POINTER(capi.Result)]
        _LIB.covgpu_get_profile2.argtypes = [C.c_void_p, capi._dp]
        _LIB.covgpu_get_profile2.restype = None
        _LIB.covgpu_get_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        _LIB.covgpu_get_layout.restype = None
    return _LIB


def pgo_partition(num_kf: int, edge_i, edge_j):
    """Host-only: block index per keyframe (-1 = border) and the number of blocks (0 = solved densely) of the
    block-arrow pose-graph solve (covgpu_pgo_partition, include/covgpu.h)."""
    import numpy as np
    ei = np.ascontiguousarray(edge_i, np.int32); ej = np.ascontiguousarray(edge_j, np.int32)
    out = np.empty(num_kf, np.int32)
    n = lib().covgpu_pgo_partition(num_kf, len(ei), ei.ctypes.data_as(capi._ip), ej.ctypes.data_as(capi._ip), out.ctypes.data_as(capi._ip))
    return out, int(n)


def default_options(**kw) -> Options:
    o = Options()
    lib().covgpu_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def gba_two_round_multi(prob: FlatProblem, opt: Options, threshold: float, devices, round1_iterations: int = 5, use_loops_round2: bool = True,
                        loop_loss_round2: float = 1.0, kf_fixed_round2=None):
    """Both rounds of GlobalBundleAdjustment on len(devices) ranks behind one call (covgpu_gba_two_round_multi; ranks that share a device are virtual
    ranks on the library's in-process group). Returns what Context.gba_two_round returns."""
    q = prob.copy(); s = q.as_struct(); r1 = Result(); r2 = Result()
    erase = np.zeros(max(prob.O, 1), np.uint8); left = np.zeros(max(prob.L, 1), np.int32); cnt = (C.c_int64 * 2)()
    tr = capi.TwoRound()
    tr.outlier_threshold = float(threshold); tr.round1_iterations = int(round1_iterations); tr.use_loops_round2 = int(bool(use_loops_round2))
    tr.loop_loss_round2 = float(loop_loss_round2)
    fx = None
    if kf_fixed_round2 is not None:
        fx = np.ascontiguousarray(kf_fixed_round2, np.uint8)
        tr.kf_fixed_round2 = fx.ctypes.data_as(C.POINTER(C.c_uint8))
    dev = np.ascontiguousarray(devices, np.int32)
    rc = lib().covgpu_gba_two_round_multi(C.byref(opt), C.byref(s), C.byref(tr), len(dev), iptr(dev), erase.ctypes.data_as(capi._bp), iptr(left), cnt,
                                          C.byref(r1), C.byref(r2))
    if rc != 0:
        raise CovGpuError(f"covgpu error {rc}: {lib().covgpu_last_error().decode()}")
    return q, r1, r2, erase[:prob.O].astype(bool), left[:prob.L], (int(cnt[0]), int(cnt[1]))


class Context:
    """One solver context = one HIP stream + HBM workspace (covgpu_create / covgpu_destroy)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        o = default_options(device=device)
        self._check(lib().covgpu_create(C.byref(o), C.byref(self._h)))
        self._keep = None

    def _check(self, rc: int):
        if rc != 0:
            raise CovGpuError(f"covgpu error {rc}: {lib().covgpu_last_error().decode()}")

    def close(self):
        if self._h:
            lib().covgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- full solves (upload + solve + download)
    def gba_solve(self, prob: FlatProblem, opt: Options) -> Tuple[FlatProblem, Result]:
        q = prob.copy(); s = q.as_struct(); r = Result()
        self._check(lib().covgpu_gba_solve(self._h, C.byref(opt), C.byref(s), C.byref(r)))
        return q, r

    def gba_two_round(self, prob: FlatProblem, opt: Options, threshold: float, round1_iterations: int = 5, use_loops_round2: bool = True,
                      loop_loss_round2: float = 1.0, kf_fixed_round2=None):
        """Both rounds of GlobalBundleAdjustment behind one call (covgpu_gba_two_round): `prob` is the FIRST round's problem. Returns
        (solution of round 2 in round-1 indexing, round-1 result, round-2 result, erase flags [O] bool, lm_left [L], (erased, short))."""
        q = prob.copy(); s = q.as_struct(); r1 = Result(); r2 = Result()
        erase = np.zeros(max(prob.O, 1), np.uint8); left = np.zeros(max(prob.L, 1), np.int32); cnt = (C.c_int64 * 2)()
        tr = capi.TwoRound()
        tr.outlier_threshold = float(threshold); tr.round1_iterations = int(round1_iterations); tr.use_loops_round2 = int(bool(use_loops_round2))
        tr.loop_loss_round2 = float(loop_loss_round2)
        fx = None
        if kf_fixed_round2 is not None:
            fx = np.ascontiguousarray(kf_fixed_round2, np.uint8)
            tr.kf_fixed_round2 = fx.ctypes.data_as(C.POINTER(C.c_uint8))
        self._check(lib().covgpu_gba_two_round(self._h, C.byref(opt), C.byref(s), C.byref(tr), erase.ctypes.data_as(capi._bp), iptr(left), cnt,
                                               C.byref(r1), C.byref(r2)))
        return q, r1, r2, erase[:prob.O].astype(bool), left[:prob.L], (int(cnt[0]), int(cnt[1]))

    def pgo_solve(self, prob: FlatProblem, opt: Options) -> Tuple[FlatProblem, Result]:
        q = prob.copy(); s = q.as_struct(); r = Result()
        self._check(lib().covgpu_pgo_solve(self._h, C.byref(opt), C.byref(s), C.byref(r)))
        return q, r

    # ---- split form: inputs resident in HBM before the timed region
    def upload(self, prob: FlatProblem, opt: Options, pgo: bool = False):
        self._keep = prob
        s = prob.as_struct()
        fn = lib().covgpu_upload_pgo if pgo else lib().covgpu_upload
        self._check(fn(self._h, C.byref(opt), C.byref(s)))

    def solve_resident(self, opt: Options) -> Result:
        r = Result()
        self._check(lib().covgpu_solve_resident(self._h, C.byref(opt), C.byref(r)))
        return r

    def download(self) -> FlatProblem:
        q = self._keep.copy(); s = q.as_struct()
        self._check(lib().covgpu_download(self._h, C.byref(s)))
        return q

    def set_shard_group(self, plan, rank: int, group):
        """Agent-sharded solve among host threads of THIS process (virtual ranks): `plan` = distrib.ShardPlan, `group` = distrib.Group."""
        self._check(lib().covgpu_set_shard_group(self._h, plan.handle, int(rank), group.handle))
        self._shard_keep = (plan, group)

    def set_shard_rccl(self, plan, rank: int, world: int, unique_id: bytes):
        """Agent-sharded solve, one process per GPU: RCCL communicator from rank 0's covgpu_rccl_unique_id."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(lib().covgpu_set_shard_rccl(self._h, plan.handle, int(rank), int(world), buf))
        self._shard_keep = (plan,)

    def set_shard_none(self):
        self._check(lib().covgpu_set_shard_none(self._h))
        self._shard_keep = None

    def allreduce_host(self, a: np.ndarray, op: int = 0) -> np.ndarray:
        """Sum (0) / max (1) of a small host vector over the ranks of the context's collective (identity without one)."""
        a = np.ascontiguousarray(a, dtype=np.float64).copy()
        self._check(lib().covgpu_allreduce_host(self._h, dptr(a), a.size, int(op)))
        return a

    def shard_stats(self) -> dict:
        out = (C.c_int64 * 4)()
        lib().covgpu_shard_stats(self._h, out)
        return {"collectives": int(out[0]), "bytes": int(out[1]), "rank": int(out[2]), "world": int(out[3])}

    def outlier_pass(self, n_obs: int, n_lm: int, threshold: float):
        """Outlier flags and per-landmark remaining-observation counts at the resident estimate (covgpu_outlier_pass)."""
        erase = np.zeros(max(n_obs, 1), np.uint8); left = np.zeros(max(n_lm, 1), np.int32); cnt = (C.c_int64 * 2)()
        self._check(lib().covgpu_outlier_pass(self._h, float(threshold), erase.ctypes.data_as(capi._bp), iptr(left), cnt))
        return erase[:n_obs].astype(bool), left[:n_lm], (int(cnt[0]), int(cnt[1]))

    def covisibility(self, threshold: int):
        """Covisibility recount on the resident problem (covgpu_covisibility): (kf_i, kf_j, weight) with kf_i > kf_j, weight >= threshold."""
        cap = 1 << 16
        while True:
            ki = np.zeros(cap, np.int32); kj = np.zeros(cap, np.int32); w = np.zeros(cap, np.int32); n = C.c_int64()
            self._check(lib().covgpu_covisibility(self._h, int(threshold), cap, iptr(ki), iptr(kj), iptr(w), C.byref(n)))
            if n.value <= cap:
                return ki[:n.value], kj[:n.value], w[:n.value]
            cap = int(n.value)

    def relpose_batch(self, bt: dict, th_outlier: float = 1.3, min_inliers: int = 12):
        """Batched Optimization::OptimizeRelativePose (covgpu_relpose_batch). `bt`: dict with ptr, pA, pB, kpA, kpB, sigA, sigB,
        camA, camB, distA, distB, T0 (see include/covgpu.h). Returns (T_ab [B,7], outlier flags [C], inliers [B])."""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        keep = dict(ptr=i(bt["ptr"]), pB=f(bt["pB"]), pA=f(bt["pA"]), kA=f(bt["kpA"]), kB=f(bt["kpB"]), sA=f(bt["sigA"]), sB=f(bt["sigB"]),
                    cA=f(bt["camA"]), cB=f(bt["camB"]), dA=i(bt["distA"]), dB=i(bt["distB"]), T=np.array(bt["T0"], dtype=np.float64, order="C"))
        B = len(keep["ptr"]) - 1
        out = np.zeros(max(int(keep["ptr"][-1]), 1), np.uint8); inl = np.zeros(max(B, 1), np.int32)
        s = capi.RelposeBatch(B, iptr(keep["ptr"]), dptr(keep["pB"]), dptr(keep["pA"]), dptr(keep["kA"]), dptr(keep["kB"]), dptr(keep["sA"]),
                              dptr(keep["sB"]), dptr(keep["cA"]), dptr(keep["cB"]), iptr(keep["dA"]), iptr(keep["dB"]), dptr(keep["T"]),
                              out.ctypes.data_as(capi._bp), iptr(inl))
        self._check(lib().covgpu_relpose_batch(self._h, C.byref(s), float(th_outlier), int(min_inliers)))
        return keep["T"], out[:int(keep["ptr"][-1])].astype(bool), inl[:B]

    def set_profiling(self, on: bool):
        lib().covgpu_set_profiling(self._h, int(on))

    def profile(self) -> dict:
        out = np.zeros(16)
        lib().covgpu_get_profile2(self._h, dptr(out))
        return dict(build_ms=out[0], n_build=int(out[1]), factor_ms=out[2], n_factor=int(out[3]), syrk_ms=out[4],
                    n_syrk=int(out[5]), syrk_flops=out[6], offdiag_blocks=int(out[7]),
                    potrf_ms=out[8], n_potrf=int(out[9]), potrf_flops=out[10], plan_flops=out[11])

    def layout(self) -> dict:
        out = (C.c_int64 * 16)()
        lib().covgpu_get_layout(self._h, out)
        keys = ("shard_world", "shard_rank", "top_unknowns", "top_levels", "allreduce_kib", "stream_ordering", "dense_order", "covisible_pairs",
                "edge_pairs", "chains", "device_mib", "nd_fronts", "nd_levels", "nd_serial_panels", "nd_root_order", "nd_front_mib")
        return {k: int(out[i]) for i, k in enumerate(keys)}

    # ---- per-kernel entry points (tests)
    def residual_norms(self, prob, opt):
        out = np.zeros(prob.O); s = prob.as_struct()
        self._check(lib().covgpu_reprojection_residual_norms(self._h, C.byref(opt), C.byref(s), dptr(out)))
        return out

    def linearize_reprojection(self, prob, opt):
        O = prob.O
        r, Jp, Jl, c = np.zeros((O, 2)), np.zeros((O, 12)), np.zeros((O, 6)), np.zeros(O)
        s = prob.as_struct()
        self._check(lib().covgpu_linearize_reprojection(self._h, C.byref(opt), C.byref(s), dptr(r), dptr(Jp), dptr(Jl), dptr(c)))
        return r, Jp, Jl, c

    def preintegrate(self, prob, opt):
        I = prob.I
        d, J, P = np.zeros((I, 11)), np.zeros((I, 225)), np.zeros((I, 225))
        s = prob.as_struct()
        self._check(lib().covgpu_preintegrate(self._h, C.byref(opt), C.byref(s), dptr(d), dptr(J), dptr(P)))
        return d, J, P

    def linearize_imu(self, prob, opt):
        I = prob.I
        r, J = np.zeros((I, 15)), np.zeros((I, 450))
        s = prob.as_struct()
        self._check(lib().covgpu_linearize_imu(self._h, C.byref(opt), C.byref(s), dptr(r), dptr(J)))
        return r, J

    def linearize_between(self, prob, opt):
        E = prob.E
        r, J, c = np.zeros((E, 6)), np.zeros((E, 72)), np.zeros(E)
        s = prob.as_struct()
        self._check(lib().covgpu_linearize_between(self._h, C.byref(opt), C.byref(s), dptr(r), dptr(J), dptr(c)))
        return r, J, c

    def schur(self, prob, opt, mu, pgo=False):
        n = (6 if (pgo or opt.visual_only) else 15) * prob.K
        S, b, c = np.zeros((n, n)), np.zeros(n), np.zeros(1)
        s = prob.as_struct()
        fn = lib().covgpu_schur_pgo if pgo else lib().covgpu_schur
        self._check(fn(self._h, C.byref(opt), C.byref(s), float(mu), dptr(S), dptr(b), dptr(c)))
        return S, b, float(c[0])

    def gn_step(self, prob, opt, mu):
        """One damped Gauss-Newton step through the product solve path: (dx[n], dl[L,3], cost)."""
        n = (6 if opt.visual_only else 15) * prob.K
        dx, dl, c = np.zeros(n), np.zeros((prob.L, 3)), np.zeros(1)
        s = prob.as_struct()
        self._check(lib().covgpu_gn_step(self._h, C.byref(opt), C.byref(s), float(mu), dptr(dx), dptr(dl), dptr(c)))
        return dx, dl, float(c[0])

    def solve_reduced(self, S, b):
        S = np.ascontiguousarray(S, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros(b.shape[0])
        rc = lib().covgpu_solve_reduced(self._h, b.shape[0], dptr(S), dptr(b), dptr(x))
        return rc, x

    def pgo_reanchor(self, pose_old, pose_new, velocity, ref_kf, lm_pos):
        po = np.ascontiguousarray(pose_old, dtype=np.float64); pn = np.ascontiguousarray(pose_new, dtype=np.float64)
        vel = None if velocity is None else np.array(velocity, dtype=np.float64, order="C")
        lm = np.array(lm_pos, dtype=np.float64, order="C").reshape(-1, 3)
        ref = np.ascontiguousarray(ref_kf, dtype=np.int32)
        self._check(lib().covgpu_pgo_reanchor(self._h, po.shape[0], dptr(po), dptr(pn), dptr(vel), lm.shape[0], iptr(ref), dptr(lm)))
        return vel, lm
