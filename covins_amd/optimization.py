"""Host-side mirror of `covins::Optimization` (covins_backend/include/covins/covins_backend/
optimization_be.hpp:34-53) over `SlamMap`: same entry points, argument meaning and side effects as
optimization_be.cpp:56-618 (GlobalBundleAdjustment) and :833-1086 (PoseGraphOptimization); every
ceres::Solve is replaced by a call through the C ABI of libcovgpu (include/covgpu.h).
The C++ counterpart for the real COVINS classes is include/covins_gpu/optimization_gpu.hpp.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

from . import backend, capi, mapdata
from .mapdata import PgoParams, SlamMap


@dataclass
class OptParams:
    """covins_params::opt keys read on this path (config/config_backend.yaml:115-140)."""
    gba_use_map_loop_constraints: bool = True
    th_gba_outlier_global: float = 0.92
    gba_fix_poses_loaded_maps: bool = False
    strategy: int = capi.COVGPU_DOGLEG       # the reference runs DOGLEG; COVGPU_LM is the north-star mode
    pgo: PgoParams = None

    def __post_init__(self):
        if self.pgo is None:
            self.pgo = PgoParams()


class Optimization:
    """Static-function facade like the reference class (constructor deleted there, optimization_be.hpp:36)."""

    @staticmethod
    def GlobalBundleAdjustment(map_: SlamMap, interations_limit: int, time_limit: float = -1.0, visual_only: bool = False,
                               outlier_removal: bool = True, estimate_bias: bool = False, *, params: Optional[OptParams] = None,
                               ctx: Optional[backend.Context] = None, device_second_round: bool = True) -> Dict[str, object]:
        """`time_limit` and `estimate_bias` are accepted and ignored, exactly like the reference (never read).
        device_second_round: the second round's problem is derived on the device from the resident first round (covgpu_gba_two_round:
        one flatten and one upload per call); False: the reference's literal sequence — flatten, solve, erase, flatten again, solve."""
        import time
        prm = params or OptParams()
        own = ctx is None
        ctx = ctx or backend.Context()
        info: Dict[str, object] = {}
        st: Dict[str, float] = {}   # stage seconds of this call (bench.py e2e_call)
        info["stages_s"] = st

        def add(key, t0):
            st[key] = st.get(key, 0.0) + time.perf_counter() - t0

        def solve(prob, opt):
            sol, res = ctx.gba_solve(prob, opt)
            st["upload (host plan + H2D)"] = st.get("upload (host plan + H2D)", 0.0) + res.t_upload_s
            st["solve"] = st.get("solve", 0.0) + res.t_solve_s
            st["download"] = st.get("download", 0.0) + res.t_download_s
            return sol, res
        try:
            if outlier_removal and device_second_round:
                # ONE flatten (the first round's problem, :80-254) and ONE upload; rounds, erase decisions and the rebuilt second-round
                # problem (:296-557) on the device. The map is brought to the state the reference leaves: observations erased
                # (:281-289), estimate of the second round written back (:572-609), Map::Clean (:614).
                t0 = time.perf_counter()
                prob, idx = mapdata.flatten_gba(map_, visual_only, loop_loss=False, use_loops=True)
                fixed2 = None
                if prm.gba_fix_poses_loaded_maps:
                    fixed2 = prob.kf_fixed.copy(); fixed2[map_.kf_loaded[idx.kf_rows]] = 1
                add("flatten", t0)
                opt = backend.default_options(strategy=prm.strategy, max_iterations=int(interations_limit), visual_only=int(visual_only))
                sol, res1, res2, bad, lm_left, (n_bad, n_short) = ctx.gba_two_round(
                    prob, opt, prm.th_gba_outlier_global, 5, prm.gba_use_map_loop_constraints, 1.0, fixed2)
                st["upload (host plan + H2D)"] = res1.t_upload_s
                st["second round rebuilt on the device"] = res2.t_upload_s
                st["solve"] = res1.t_solve_s + res2.t_solve_s
                st["download"] = res2.t_download_s
                t0 = time.perf_counter()
                mask = np.zeros(map_.O, bool)
                mask[idx.obs_rows[bad]] = True
                map_.erase_observations(mask)
                add("erase observations", t0)
                info["outliers_removed"] = int(bad.sum()); info["round1"] = res1; info["round2"] = res2
                info["landmarks_left_short"] = n_short
                keep = lm_left >= 2   # landmarks of the second round (:428-440)
                info["problem"] = (prob.K, int(keep.sum()), int(prob.O - bad.sum() - lm_left[~keep].sum()), prob.I, prob.E)
                t0 = time.perf_counter()
                q = sol.kf_pose[:, :4] / np.linalg.norm(sol.kf_pose[:, :4], axis=1, keepdims=True)
                map_.kf_pose[idx.kf_rows, :4] = q
                map_.kf_pose[idx.kf_rows, 4:] = sol.kf_pose[:, 4:]
                if not visual_only:
                    map_.kf_velocity[idx.kf_rows] = sol.kf_speed_bias[:, 0:3]
                    map_.kf_bias_a[idx.kf_rows] = sol.kf_speed_bias[:, 3:6]
                    map_.kf_bias_g[idx.kf_rows] = sol.kf_speed_bias[:, 6:9]
                map_.kf_gba_optimized[idx.kf_rows] = True
                map_.lm_pos[idx.lm_rows[keep]] = sol.lm_pos[keep]
                map_.lm_gba_optimized[idx.lm_rows[keep]] = True
                add("write-back", t0)
                t0 = time.perf_counter()
                info["cleaned"] = map_.clean()  # map->Clean() (opt_be.cpp:614)
                add("Map::Clean", t0)
                return info
            if outlier_removal:  # first round (opt_be.cpp:62-293): 5 iterations, loop edges without loss, then erase outliers
                t0 = time.perf_counter()
                prob, idx = mapdata.flatten_gba(map_, visual_only, loop_loss=False, use_loops=True)
                add("flatten", t0)
                opt = backend.default_options(strategy=prm.strategy, max_iterations=5, visual_only=int(visual_only))
                sol, res = solve(prob, opt)
                # problem.Evaluate applies the loss (opt_be.cpp:270-274); decisions taken on the device at the resident estimate
                t0 = time.perf_counter()
                bad, lm_left, (n_bad, n_short) = ctx.outlier_pass(prob.O, prob.L, prm.th_gba_outlier_global)
                add("outlier pass", t0)
                info["landmarks_left_short"] = n_short
                t0 = time.perf_counter()
                mask = np.zeros(map_.O, bool)
                mask[idx.obs_rows[bad]] = True
                map_.erase_observations(mask)
                add("erase observations", t0)
                info["outliers_removed"] = int(bad.sum()); info["round1"] = res
                # NB: the outlier round's estimate is discarded — round 2 restarts from the map state (opt_be.cpp:327,454)
            t0 = time.perf_counter()
            prob, idx = mapdata.flatten_gba(map_, visual_only, loop_loss=True, use_loops=prm.gba_use_map_loop_constraints,
                                            fix_loaded=prm.gba_fix_poses_loaded_maps)
            add("flatten", t0)
            opt = backend.default_options(strategy=prm.strategy, max_iterations=int(interations_limit), visual_only=int(visual_only))
            sol, res = solve(prob, opt)
            t0 = time.perf_counter()
            info["round2"] = res; info["problem"] = (prob.K, prob.L, prob.O, prob.I, prob.E)
            # write-back (opt_be.cpp:572-609); Ceres2Transform normalises q (utils_base.cpp:38-40)
            q = sol.kf_pose[:, :4] / np.linalg.norm(sol.kf_pose[:, :4], axis=1, keepdims=True)
            map_.kf_pose[idx.kf_rows, :4] = q
            map_.kf_pose[idx.kf_rows, 4:] = sol.kf_pose[:, 4:]
            if not visual_only:
                map_.kf_velocity[idx.kf_rows] = sol.kf_speed_bias[:, 0:3]
                map_.kf_bias_a[idx.kf_rows] = sol.kf_speed_bias[:, 3:6]
                map_.kf_bias_g[idx.kf_rows] = sol.kf_speed_bias[:, 6:9]
            map_.kf_gba_optimized[idx.kf_rows] = True
            map_.lm_pos[idx.lm_rows] = sol.lm_pos
            map_.lm_gba_optimized[idx.lm_rows] = True
            add("write-back", t0)
            t0 = time.perf_counter()
            info["cleaned"] = map_.clean()  # map->Clean() (opt_be.cpp:614)
            add("Map::Clean", t0)
        finally:
            if own:
                ctx.close()
        return info

    @staticmethod
    def PoseGraphOptimization(map_: SlamMap, corrected_poses: Dict[int, np.ndarray], *, params: Optional[OptParams] = None,
                              ctx: Optional[backend.Context] = None) -> Dict[str, object]:
        prm = params or OptParams()
        own = ctx is None
        ctx = ctx or backend.Context()
        try:
            prob, idx = mapdata.flatten_pgo(map_, corrected_poses, prm.pgo)
            opt = backend.default_options(strategy=prm.strategy, max_iterations=prm.pgo.pgo_iteration_limit)
            sol, res = ctx.pgo_solve(prob, opt)
            pose_old = map_.kf_pose.copy()                      # non_corrected_poses (opt_be.cpp:1042-1043)
            q = sol.kf_pose[:, :4] / np.linalg.norm(sol.kf_pose[:, :4], axis=1, keepdims=True)
            map_.kf_pose[idx.kf_rows, :4] = q
            map_.kf_pose[idx.kf_rows, 4:] = sol.kf_pose[:, 4:]
            # landmarks without a (valid) reference keyframe are erased (opt_be.cpp:1059-1072)
            ref = map_.lm_ref_kf.copy()
            has_obs = np.diff(map_.lm_obs_ptr) > 0
            no_ref = (ref < 0) | map_.kf_invalid[np.maximum(ref, 0)]
            erase = (~map_.lm_invalid) & no_ref & (has_obs | (ref >= 0))
            map_.lm_invalid |= erase
            ref_eff = np.where(map_.lm_invalid | no_ref, -1, ref).astype(np.int32)
            vel, lm = ctx.pgo_reanchor(pose_old, map_.kf_pose, map_.kf_velocity, ref_eff, map_.lm_pos)
            map_.kf_velocity[~map_.kf_invalid] = vel[~map_.kf_invalid]
            map_.lm_pos = lm
            return dict(result=res, erased_landmarks=int(erase.sum()), problem=(prob.K, prob.E))
        finally:
            if own:
                ctx.close()
