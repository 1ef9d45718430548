"""Struct-of-arrays stand-in for the COVINS back-end map, and its flattening into the solver IR.

The reference keeps `Map` / `Keyframe` / `Landmark` as shared_ptr graphs (covins_backend/include/covins/
covins_backend/map_be.hpp:63-209, covins_base/keyframe_base.hpp:51-238, landmark_base.hpp:43-119). Only the
fields that `Optimization::GlobalBundleAdjustment` / `PoseGraphOptimization` read or write cross our
boundary; `SlamMap` holds exactly those, one numpy array per field. The flattening
functions walk it in the order optimization_be.cpp does and apply the same gating rules.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import capi


@dataclass
class LoopConstraint:
    """typedefs_base.hpp:264-277: kf1, kf2 (indices into the map), T_s1_s2 as [q(4), t(3)], 6x6 covariance."""
    kf1: int
    kf2: int
    T_s1_s2: np.ndarray
    cov: np.ndarray = field(default_factory=lambda: np.eye(6))


@dataclass
class SlamMap:
    id_map: int
    # keyframes, stored sorted by (kf_id, client_id) like the reference's std::map (typedefs_base.hpp:178)
    kf_id: np.ndarray            # [K] id_.first
    kf_client: np.ndarray        # [K] id_.second (agent)
    kf_time: np.ndarray          # [K] seconds
    kf_invalid: np.ndarray       # [K] bool
    kf_loaded: np.ndarray        # [K] bool   is_loaded_
    kf_gba_optimized: np.ndarray # [K] bool   is_gba_optimized_
    kf_pose: np.ndarray          # [K,7] T_w_s
    kf_pose_vio: np.ndarray      # [K,7] T_w_s_vio (GetPoseTws_vio)
    kf_velocity: np.ndarray      # [K,3]
    kf_bias_a: np.ndarray        # [K,3]
    kf_bias_g: np.ndarray        # [K,3]
    kf_pred: np.ndarray          # [K] index of predecessor or -1
    kf_succ: np.ndarray          # [K] index of successor or -1
    kf_cam: np.ndarray           # [K] camera index
    cam_extr: np.ndarray         # [A,7] T_s_c
    cam_intr: np.ndarray         # [A,4]
    cam_dist: np.ndarray         # [A,4]
    cam_dist_type: np.ndarray    # [A]
    cam_imu_calib: np.ndarray    # [A,5] VICalibration of the agent's keyframes: sigma_a_c sigma_g_c sigma_aw_c sigma_gw_c g
                                 #       (typedefs_base.hpp:333-340; every keyframe's preintegrator uses its own, keyframe_be.cpp:187-195)
    # per-KF raw IMU between predecessor and this KF (preintegrated_imu_): CSR over keyframes
    imu_ptr: np.ndarray          # [K+1]
    imu_samples: np.ndarray      # [S,7] dt, acc, gyr
    imu_first: np.ndarray        # [K,6] reading at the predecessor (lin_acc_init / ang_vel_init)
    # landmarks
    lm_pos: np.ndarray           # [L,3]
    lm_invalid: np.ndarray       # [L] bool
    lm_ref_kf: np.ndarray        # [L] reference keyframe index or -1
    lm_gba_optimized: np.ndarray # [L] bool
    # observations, CSR by landmark: (kf index, keypoint uv float32, octave)
    lm_obs_ptr: np.ndarray       # [L+1]
    obs_kf: np.ndarray           # [O]
    obs_uv: np.ndarray           # [O,2] float32 (keypoints_distorted_)
    obs_octave: np.ndarray       # [O]   keypoints_aors_[.][1]
    loops: List[LoopConstraint] = field(default_factory=list)
    # ground truth (synthetic maps only; used for ATE, never by the optimiser)
    truth: Dict[str, np.ndarray] = field(default_factory=dict)

    @property
    def K(self): return self.kf_pose.shape[0]
    @property
    def L(self): return self.lm_pos.shape[0]
    @property
    def O(self): return self.obs_kf.shape[0]

    def copy(self) -> "SlamMap":
        d = {}
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                d[k] = v.copy()
            elif k == "loops":
                d[k] = [LoopConstraint(l.kf1, l.kf2, l.T_s1_s2.copy(), l.cov.copy()) for l in v]
            elif k == "truth":
                d[k] = {a: b.copy() for a, b in v.items()}
            else:
                d[k] = v
        return SlamMap(**d)

    def gauge_kf(self) -> int:
        """Index of KF (0, id_map) whose pose block is held constant (opt_be.cpp:329-331, 870-871)."""
        m = np.nonzero((self.kf_id == 0) & (self.kf_client == self.id_map))[0]
        return int(m[0]) if len(m) else -1

    def erase_observations(self, obs_mask: np.ndarray) -> None:
        """kf->EraseLandmark + lm->EraseObservation for every flagged observation (opt_be.cpp:285-286)."""
        keep = ~obs_mask
        counts = np.add.reduceat(keep.astype(np.int64), self.lm_obs_ptr[:-1]) if self.O else np.zeros(self.L, np.int64)
        counts[np.diff(self.lm_obs_ptr) == 0] = 0
        rows = np.flatnonzero(keep)   # (one index list, three takes: a boolean mask is re-scanned by every indexing)
        self.obs_kf, self.obs_uv, self.obs_octave = self.obs_kf.take(rows), self.obs_uv.take(rows, axis=0), self.obs_octave.take(rows)
        self.lm_obs_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)

    def clean(self) -> int:
        """Map::Clean (map_be.cpp:448-454, 698-743): invalidate landmarks left with < 2 observations."""
        n = np.diff(self.lm_obs_ptr)
        bad = (n < 2) & ~self.lm_invalid
        self.lm_invalid = self.lm_invalid | bad
        return int(bad.sum())


@dataclass
class FlatIndex:
    """Maps IR rows back to map rows for the write-back passes."""
    kf_rows: np.ndarray   # [K_ir] map keyframe index
    lm_rows: np.ndarray   # [L_ir] map landmark index
    obs_rows: np.ndarray  # [O_ir] map observation index


GBA_LOOP_SQRT_INFO = np.diag([100.0] * 3 + [1e4] * 3)  # opt_be.cpp:238-240, 534-536


def flatten_gba(m: SlamMap, visual_only: bool, loop_loss: bool, use_loops: bool = True,
                fix_loaded: bool = False) -> Tuple[capi.FlatProblem, FlatIndex]:
    """Map -> IR for one GBA round, following optimization_be.cpp:320-557 (round 2) / :80-254 (round 1;
    `loop_loss=False`: the outlier round adds loop edges without a loss, :253).

    Gating: invalid KFs skipped (:322); landmarks need >= 2 observations from valid KFs (:438-453);
    IMU factor per KF with a valid predecessor and > 0 samples (:369-385); loop edges whose two KFs are
    both in the problem (:546-549). A KF with id != 0 and no predecessor is fatal in the reference (:371-373).
    """
    valid = ~m.kf_invalid
    kf_rows = np.nonzero(valid)[0].astype(np.int32)
    remap = -np.ones(m.K, np.int32)
    remap[kf_rows] = np.arange(len(kf_rows), dtype=np.int32)
    fixed = np.zeros(len(kf_rows), np.uint8)
    g = m.gauge_kf()
    if g >= 0 and remap[g] >= 0:
        fixed[remap[g]] = 1
    if fix_loaded:  # opt.gba_fix_poses_loaded_maps (opt_be.cpp:338-341)
        fixed[m.kf_loaded[kf_rows]] = 1

    # landmarks / observations
    obs_valid = valid[m.obs_kf] if m.O else np.zeros(0, bool)
    n_all = np.diff(m.lm_obs_ptr)
    if m.O:
        csum = np.concatenate([[0], np.cumsum(obs_valid)])
        n_ok = csum[m.lm_obs_ptr[1:]] - csum[m.lm_obs_ptr[:-1]]
    else:
        n_ok = np.zeros(m.L, np.int64)
    lm_in = (~m.lm_invalid) & (n_all >= 2) & (n_ok >= 2)
    lm_rows = np.nonzero(lm_in)[0].astype(np.int32)
    obs_lm = np.repeat(np.arange(m.L), n_all)
    obs_keep = obs_valid & lm_in[obs_lm] if m.O else np.zeros(0, bool)
    obs_rows = np.nonzero(obs_keep)[0].astype(np.int32)
    cnt = np.bincount(obs_lm[obs_rows], minlength=m.L) if len(obs_rows) else np.zeros(m.L, np.int64)
    lm_obs_ptr = np.concatenate([[0], np.cumsum(cnt[lm_rows])]).astype(np.int32)

    # IMU factors
    imu_i, imu_j, ptr, first, chunks, noise = [], [], [0], [], [], []
    if not visual_only:
        for k in kf_rows:
            p = m.kf_pred[k]
            if p < 0 or m.kf_invalid[p]:
                if m.kf_id[k] != 0:
                    raise RuntimeError(f"KF {int(m.kf_id[k])}|{int(m.kf_client[k])}: no predecessor")  # exit(-1), :371-373
                continue
            s0, s1 = m.imu_ptr[k], m.imu_ptr[k + 1]
            if s1 == s0:
                continue  # "0 IMU measurements - skip IMU factor" (:382-385)
            imu_i.append(remap[p]); imu_j.append(remap[k])
            chunks.append(m.imu_samples[s0:s1]); ptr.append(ptr[-1] + (s1 - s0)); first.append(m.imu_first[k])
            noise.append(m.cam_imu_calib[m.kf_cam[k]])
    samples = np.concatenate(chunks) if chunks else np.zeros((0, 7))

    # loop edges
    ei, ej, meas = [], [], []
    if use_loops:
        for lc in m.loops:
            a, b = remap[lc.kf1], remap[lc.kf2]
            if a < 0 or b < 0:
                continue  # "Loop KF missing -- skip loop" (:546-549)
            ei.append(a); ej.append(b); meas.append(lc.T_s1_s2)
    E = len(ei)
    all_obs = len(obs_rows) == m.O

    prob = capi.FlatProblem(
        kf_pose=m.kf_pose[kf_rows],
        kf_speed_bias=np.concatenate([m.kf_velocity[kf_rows], m.kf_bias_a[kf_rows], m.kf_bias_g[kf_rows]], axis=1),
        kf_fixed=fixed, kf_cam=m.kf_cam[kf_rows],
        cam_extr=m.cam_extr.copy(), cam_intr=m.cam_intr.copy(), cam_dist=m.cam_dist.copy(), cam_dist_type=m.cam_dist_type.copy(),  # (the IR owns its arrays)
        lm_pos=m.lm_pos[lm_rows], lm_obs_ptr=lm_obs_ptr,
        # (every observation kept — the usual case — : contiguous conversions instead of gathers through obs_rows)
        obs_kf=(remap[m.obs_kf] if all_obs else remap[m.obs_kf[obs_rows]]) if len(obs_rows) else np.zeros(0, np.int32),
        obs_uv=(m.obs_uv if all_obs else m.obs_uv[obs_rows]).astype(np.float64),                     # float -> double (opt_be.cpp:477)
        obs_sigma=((m.obs_octave if all_obs else m.obs_octave[obs_rows]).astype(np.float64) + 1.0) * 2.0,  # (opt_be.cpp:478)
        imu_kf_i=np.array(imu_i, np.int32), imu_kf_j=np.array(imu_j, np.int32),
        imu_sample_ptr=np.array(ptr, np.int32), imu_samples=samples,
        imu_first=np.array(first).reshape(-1, 6) if first else np.zeros((0, 6)),
        imu_noise=np.array(noise).reshape(-1, 5) if noise else np.zeros((0, 5)),
        edge_i=np.array(ei, np.int32), edge_j=np.array(ej, np.int32),
        edge_meas=np.array(meas).reshape(-1, 7) if E else np.zeros((0, 7)),
        edge_sqrt_info=np.tile(GBA_LOOP_SQRT_INFO.reshape(1, 36), (E, 1)),
        edge_loss_a=np.full(E, 1.0 if loop_loss else 0.0),
    )
    return prob, FlatIndex(kf_rows, lm_rows, obs_rows)


@dataclass
class PgoParams:
    """covins_params::opt / placerec keys read by PoseGraphOptimization (config_backend.yaml:121-140)."""
    wt_kf_r: float = 10.0
    wt_kf_t: float = 1.0
    wt_kf_n1: float = 10.0
    wt_kf_n23: float = 2.0
    wt_kf_n45: float = 3.0
    use_nbr_kfs: bool = True
    use_robust_loss: bool = True
    robust_loss_th: float = 0.5
    pgo_fix_kfs_after_gba: bool = True
    pgo_fix_poses_loaded_maps: bool = False
    placerec_type: str = "COVINS"
    pgo_iteration_limit: int = 10


def _pose_inv_mul(Ta: np.ndarray, Tb: np.ndarray) -> np.ndarray:
    """[q,t] of Ta^-1 * Tb for pose rows [qx,qy,qz,qw,px,py,pz]."""
    from scipy.spatial.transform import Rotation as R
    Ra, Rb = R.from_quat(Ta[:4]), R.from_quat(Tb[:4])
    Rab = Ra.inv() * Rb
    q = Rab.as_quat()
    if q[3] < 0:
        q = -q
    t = Ra.inv().apply(Tb[4:] - Ta[4:])
    return np.concatenate([q, t])


def flatten_pgo(m: SlamMap, corrected_poses: Dict[int, np.ndarray], prm: PgoParams) -> Tuple[capi.FlatProblem, FlatIndex]:
    """Map -> IR for PoseGraphOptimization, following optimization_be.cpp:846-1021.

    `corrected_poses`: map keyframe index -> pose row, the PoseMap of placerec_be.cpp:222-285.
    Edge order = loops (:912-944), successor edges from VIO poses (:947-972), then up to five
    previous-neighbour edges per KF with weights W, W/n23, W/n23, W/n45, W/n45 (:976-1021); duplicate
    (kf, other) pairs among odometry edges are suppressed (:907, 960-966, 1010-1015).
    """
    valid = ~m.kf_invalid
    kf_rows = np.nonzero(valid)[0].astype(np.int32)
    remap = -np.ones(m.K, np.int32)
    remap[kf_rows] = np.arange(len(kf_rows), dtype=np.int32)
    pose = m.kf_pose[kf_rows].copy()
    for k, T in corrected_poses.items():
        if remap[k] >= 0:
            pose[remap[k]] = T
    fixed = np.zeros(len(kf_rows), np.uint8)
    g = m.gauge_kf()
    if g >= 0 and remap[g] >= 0:
        fixed[remap[g]] = 1
    after_gba = m.kf_gba_optimized[kf_rows] & prm.pgo_fix_kfs_after_gba          # :875-877
    loaded = ~after_gba & m.kf_loaded[kf_rows] & prm.pgo_fix_poses_loaded_maps   # else-if, :878-881
    fixed[after_gba | loaded] = 1

    W1 = np.diag([prm.wt_kf_r] * 3 + [prm.wt_kf_t] * 3) * prm.wt_kf_n1
    W23, W45 = W1 / prm.wt_kf_n23, W1 / prm.wt_kf_n45
    ei, ej, meas, info, loss = [], [], [], [], []
    for lc in m.loops:
        if remap[lc.kf1] < 0 or remap[lc.kf2] < 0:
            continue  # invalidated loop keyframe: skipped with a warning in the C++ facade
        if prm.placerec_type == "COVINS":
            Sl = W1
        else:  # chol(cov^-1)^T: upper-triangular factor (:922-923)
            Sl = np.linalg.cholesky(np.linalg.inv(lc.cov)).T
        ei.append(remap[lc.kf1]); ej.append(remap[lc.kf2]); meas.append(lc.T_s1_s2); info.append(Sl)
        loss.append(prm.robust_loss_th if prm.use_robust_loss else 0.0)
    inserted = set()
    for k in kf_rows:
        s = m.kf_succ[k]
        if s < 0 or remap[s] < 0:
            continue
        if (k, s) in inserted:
            continue
        inserted.add((k, s))
        ei.append(remap[k]); ej.append(remap[s]); info.append(W1); loss.append(0.0)
        meas.append(_pose_inv_mul(m.kf_pose_vio[k], m.kf_pose_vio[s]))
    if prm.use_nbr_kfs:
        for k in kf_rows:
            t = k
            conns = []
            for j in range(1, 6):
                if int(m.kf_id[k]) - j > 0:
                    t = m.kf_pred[t]
                    if t < 0:
                        break  # chain shorter than kf_id suggests (culled keyframes): never index with -1
                    conns.append(t)
            for n, c in enumerate(conns, start=1):
                Wn = W1 if n <= 1 else (W23 if n <= 3 else W45)
                if remap[c] < 0:
                    continue
                if (k, c) in inserted:
                    continue
                inserted.add((k, c))
                ei.append(remap[k]); ej.append(remap[c]); info.append(Wn); loss.append(0.0)
                meas.append(_pose_inv_mul(m.kf_pose_vio[k], m.kf_pose_vio[c]))
    E = len(ei)
    prob = capi.FlatProblem(
        kf_pose=pose,
        kf_speed_bias=np.concatenate([m.kf_velocity[kf_rows], m.kf_bias_a[kf_rows], m.kf_bias_g[kf_rows]], axis=1),
        kf_fixed=fixed, kf_cam=m.kf_cam[kf_rows],
        cam_extr=m.cam_extr.copy(), cam_intr=m.cam_intr.copy(), cam_dist=m.cam_dist.copy(), cam_dist_type=m.cam_dist_type.copy(),
        edge_i=np.array(ei, np.int32), edge_j=np.array(ej, np.int32),
        edge_meas=np.array(meas).reshape(-1, 7), edge_sqrt_info=np.array(info).reshape(E, 36),
        edge_loss_a=np.array(loss, np.float64),
    )
    return prob, FlatIndex(kf_rows, np.zeros(0, np.int32), np.zeros(0, np.int32))


def write_tum(path: str, m: SlamMap, client: Optional[int] = None) -> None:
    """Trajectory CSV in the format of Map::WriteKFsToFile (map_be.cpp:1067-1070): t x y z qx qy qz qw."""
    with open(path, "w") as f:
        order = np.lexsort((m.kf_id, m.kf_client))
        for k in order:
            if m.kf_invalid[k] or (client is not None and m.kf_client[k] != client):
                continue
            T = m.kf_pose[k]
            f.write(f"{m.kf_time[k]:.9f} {T[4]:.9f} {T[5]:.9f} {T[6]:.9f} {T[0]:.9f} {T[1]:.9f} {T[2]:.9f} {T[3]:.9f}\n")
