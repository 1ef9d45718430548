"""EuRoC-shaped synthetic COVINS maps (SURVEY.md §8d "Concrete synthetic inputs").

No EuRoC images or saved COVINS maps exist offline, so the BASELINE.json configurations are synthesised:
true trajectories are the EuRoC Machine-Hall ground-truth paths shipped in the reference (fixture
covins_amd/data/euroc_mh_4hz.npz, derived by tools/make_euroc_fixture.py), calibration is the EuRoC
camera/IMU (orb_slam3/Examples/Monocular-Inertial/EuRoC.yaml:9-17,30-37,40-44), the IMU samples of agents
on MH03-05 are the RECORDED 200 Hz samples of orb_slam3/Examples/Monocular-Inertial/EuRoC_IMU/MH0{3,4,5}.txt aligned by
their absolute time stamps, those of MH01/02 (no recording in the tree) and of every resampled or re-posed
path are differentiated from a C2 spline through the keyframe poses (+ white noise and bias random walk), landmarks
are spawned on the hall surfaces and tracked over a window of neighbouring keyframes (SLAM-like track
lengths); landmark fusion happens where the reference does it — around loop closures: landmarks of the
keyframes near one loop keyframe are re-observed by the keyframes near the other (PlaceRecognition::ConnectLoop
fuses the matched keyframe's and its neighbours' landmarks, placerec_be.cpp:222-285) — measurements carry
1 px noise and are stored as float32 like `keypoint_precision_t`, the initial estimate carries VIO-like
random-walk drift, and loop constraints are noisy ground-truth relative poses.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
from scipy.interpolate import CubicSpline
from scipy.spatial.transform import Rotation as R, RotationSpline

from .mapdata import LoopConstraint, SlamMap

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "euroc_mh_4hz.npz")

# EuRoC.yaml:9-17
INTR = np.array([458.654, 457.296, 367.215, 248.375])
DIST = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
WIDTH, HEIGHT = 752, 480
# EuRoC.yaml:30-37  Tbc = T_s_c (camera -> body/IMU)
TBC = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
                [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
                [0.0, 0.0, 0.0, 1.0]])
IMU_RATE = 200.0
# EuRoC.yaml:40-44, discretised as orb_slam3/src/Tracking.cc:1203-1211
SIG_G, SIG_A = 1.7e-4 * np.sqrt(IMU_RATE), 2.0e-3 * np.sqrt(IMU_RATE)
SIG_GW, SIG_AW = 1.9393e-5 / np.sqrt(IMU_RATE), 3.0e-3 / np.sqrt(IMU_RATE)
GRAVITY = 9.81
# inflated machine-hall bounding box (x[-2.8,17.6] y[-5.8,11.8] z[-1.3,3.9])
HALL_MIN = np.array([-5.0, -8.0, -2.3])
HALL_MAX = np.array([20.0, 14.0, 6.0])


@dataclass
class SynthConfig:
    agents: Sequence[int] = (1,)          # EuRoC MH sequence per agent (1..5); >5 agents re-pose copies
    max_kf_per_agent: Optional[int] = None
    kf_start: int = 0
    kf_per_agent: Optional[int] = None    # resample every agent's path to this many keyframes (1667)
    kf_dt: float = 0.25                   # keyframe spacing [s] of the resampled paths (a multiple of 1/200 s)
    new_lm_per_kf: int = 40
    track_window: int = 10                # landmark visible at most +-window keyframes around its birth
    max_obs_per_kf: int = 400             # SURVEY.md §8d cap
    min_parallax_cos: float = 0.9998      # a front-end only triangulates rays at least ~1.15 deg apart (orb_slam3/src/LocalMapping.cc:674-675)
    p_fuse: float = 0.5                   # fraction of a loop zone's landmarks that the loop closure fuses
    fuse_window: int = 8                  # loop zone: +-window keyframes around either loop keyframe
    max_fused_obs: int = 6
    fuse_mode: str = "loops"              # "loops": fusion where loop closures are (placerec_be.cpp:222-285) | "random": map-wide
    px_noise: float = 1.0
    drift_trans: float = 0.01             # random-walk std per sqrt(m), metres
    drift_yaw_deg: float = 0.05           # random-walk std per sqrt(m), degrees
    lm_noise: float = 0.02                # triangulation noise on initial landmark positions [m]
    vel_noise: float = 0.02
    loops_per_100kf: float = 1.0
    loops_per_pair: int = 3
    loop_noise_t: float = 0.02
    loop_noise_deg: float = 0.2
    outlier_frac: float = 0.0             # gross outliers among observations (for the outlier round)
    imu_noise: bool = True
    recorded_imu: bool = True             # MH03-05 at their own keyframe times: the recorded 200 Hz samples (needs imu_noise: a noise-free map is all-synthetic)
    seed: int = 0


def _project(pc: np.ndarray):
    """Pinhole + radtan projection of camera-frame points [N,3] -> pixels [N,2], validity mask."""
    z = pc[:, 2]
    ok = z > 0.3
    zs = np.where(ok, z, 1.0)
    x, y = pc[:, 0] / zs, pc[:, 1] / zs
    r2 = x * x + y * y
    ok &= r2 < 1.2  # keep the radtan model in its monotone range
    k1, k2, p1, p2 = DIST
    rad = k1 * r2 + k2 * r2 * r2
    xd = x + x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y + y * rad + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
    u = INTR[0] * xd + INTR[2]
    v = INTR[1] * yd + INTR[3]
    ok &= (u > 5) & (u < WIDTH - 5) & (v > 5) & (v < HEIGHT - 5)
    return np.stack([u, v], 1), ok


def _agent_trajectory(seq: int, agent: int, cfg: SynthConfig):
    d = np.load(_DATA)
    base = ((seq - 1) % 5) + 1
    t, p_wc, q_wc = d[f"t_{base}"], d[f"p_{base}"], d[f"q_{base}"]
    if seq > 5:  # synthetic extra agents: time-reversed / re-posed copies
        rev = (seq // 5) % 2 == 1
        if rev:
            p_wc, q_wc = p_wc[::-1].copy(), q_wc[::-1].copy()
        yaw = R.from_euler("z", 0.7 * (seq - 5))
        c = 0.5 * (HALL_MIN + HALL_MAX)
        p_wc = yaw.apply(p_wc - c) * 0.9 + c
        q_wc = (yaw * R.from_quat(q_wc)).as_quat()
    if cfg.kf_per_agent:  # same path, denser keyframes on a new time base (IMU is differentiated from the splines below)
        u = np.linspace(t[0], t[-1], cfg.kf_per_agent)
        p_wc = CubicSpline(t, p_wc)(u)
        q_wc = RotationSpline(t, R.from_quat(q_wc))(u).as_quat()
        t = np.arange(cfg.kf_per_agent) * cfg.kf_dt
    s0 = cfg.kf_start
    s1 = len(t) if cfg.max_kf_per_agent is None else min(len(t), s0 + cfg.max_kf_per_agent)
    t, p_wc, q_wc = t[s0:s1] - t[s0], p_wc[s0:s1], q_wc[s0:s1]
    p_ws, R_ws = _body_from_camera(p_wc, q_wc)
    rec = None
    if cfg.recorded_imu and cfg.imu_noise and seq <= 5 and not cfg.kf_per_agent and f"imu_{base}" in d:
        rec = _recorded_imu(d, base, s0, s1)
    return t, p_ws, R_ws, rec


def _body_from_camera(p_wc, q_wc):
    """body pose T_w_s = T_w_c * T_s_c^-1 (q_wc: camera -> world, tools/make_euroc_fixture.py)"""
    R_ws = R.from_quat(q_wc) * R.from_matrix(TBC[:3, :3]).inv()
    return p_wc - R_ws.apply(TBC[:3, 3]), R_ws


def _recorded_imu(d, base: int, s0: int, s1: int):
    """The recorded samples between keyframes s0 .. s1-1 of sequence `base`, and what stands in for the unknown truth of the
    states only an IMU observes: velocity = derivative of a spline through the 20 Hz ground truth; biases = the sequence's mean of
    (recorded - predicted from that spline), one constant per sequence (EuRoC's own estimate for the machine hall is
    b_g ~ (-0.002, 0.021, 0.077) rad/s, b_a ~ (-0.02, 0.13, 0.06) m/s^2; this reproduces it)."""
    n_sub = int(round(IMU_RATE * 0.25))
    imu = d[f"imu_{base}"]
    t20 = d[f"t20_{base}"]
    p20, R20 = _body_from_camera(d[f"p20_{base}"], d[f"q20_{base}"])
    cs, rs = CubicSpline(t20, p20), RotationSpline(t20, R20)
    ti = np.arange(len(imu)) / IMU_RATE
    inner = (ti >= 1.0) & (ti <= ti[-1] - 1.0)
    bg = (imu[inner, 0:3] - rs(ti[inner], 1)).mean(0)
    ba = (imu[inner, 3:6] - rs(ti[inner]).inv().apply(cs(ti[inner], 2) + np.array([0.0, 0.0, GRAVITY]))).mean(0)
    tk = np.arange(s0, s1) * 0.25
    rows = (np.arange(s0, s1 - 1)[:, None] * n_sub + np.arange(0, n_sub + 1)[None, :])  # [n-1, n_sub+1]; column 0 = at keyframe i
    return dict(gyr=imu[rows, 0:3], acc=imu[rows, 3:6], vel=cs(tk, 1), ba=ba, bg=bg)


def make_map(cfg: SynthConfig) -> SlamMap:
    rng = np.random.default_rng(cfg.seed)
    A = len(cfg.agents)
    R_sc = R.from_matrix(TBC[:3, :3])
    p_sc = TBC[:3, 3]
    G = np.array([0.0, 0.0, GRAVITY])

    per_agent = []
    for a, seq in enumerate(cfg.agents):
        t, p_ws, R_ws, rec = _agent_trajectory(seq, a, cfg)
        n = len(t)
        if rec is not None and n >= 2:
            per_agent.append(dict(t=t, p=p_ws, R=R_ws, v=rec["vel"], n=n, n_sub=rec["acc"].shape[1] - 1, acc=rec["acc"], gyr=rec["gyr"],
                                  ba=np.tile(rec["ba"], (n, 1)), bg=np.tile(rec["bg"], (n, 1)), recorded=True))
            continue
        cs = CubicSpline(t, p_ws, bc_type="natural")
        rs = RotationSpline(t, R_ws)
        vel = cs(t, 1)
        # IMU on the 200 Hz grid, exact from the splines
        step = 1.0 / IMU_RATE
        n_sub = int(round((t[1] - t[0]) * IMU_RATE))
        tt = (t[:-1, None] + step * np.arange(0, n_sub + 1)[None, :])  # [n-1, n_sub+1]; column 0 = at KF i
        flat = tt.reshape(-1)
        Rt = rs(flat)
        acc = Rt.inv().apply(cs(flat, 2) + G)
        gyr = rs(flat, 1)
        # biases: random walk along the agent's whole 200 Hz timeline
        nsamp = (n - 1) * n_sub + 1
        if cfg.imu_noise:
            ba0, bg0 = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
            # per-sample increment std = sigma_w * dt: the factor's covariance model integrates the
            # (already discretised) walk noise with V[B,B] = I dt
            ba_t = ba0 + np.cumsum(rng.normal(0, SIG_AW * step, (nsamp, 3)), 0)
            bg_t = bg0 + np.cumsum(rng.normal(0, SIG_GW * step, (nsamp, 3)), 0)
        else:
            ba_t, bg_t = np.zeros((nsamp, 3)), np.zeros((nsamp, 3))
        gi = (np.arange(n - 1)[:, None] * n_sub + np.arange(0, n_sub + 1)[None, :]).reshape(-1)
        acc_m = acc + ba_t[gi]
        gyr_m = gyr + bg_t[gi]
        if cfg.imu_noise:
            # the same physical sample is shared by the end of one interval and the start of the next
            na = rng.normal(0, SIG_A, (nsamp, 3)); ng = rng.normal(0, SIG_G, (nsamp, 3))
            acc_m += na[gi]; gyr_m += ng[gi]
        acc_m = acc_m.reshape(n - 1, n_sub + 1, 3); gyr_m = gyr_m.reshape(n - 1, n_sub + 1, 3)
        per_agent.append(dict(t=t, p=p_ws, R=R_ws, v=vel, n=n, n_sub=n_sub, acc=acc_m, gyr=gyr_m,
                              ba=ba_t[::n_sub][:n], bg=bg_t[::n_sub][:n]))

    # ---- global keyframe table sorted by (kf_id, client) (typedefs_base.hpp:178: agents interleaved)
    ids = np.concatenate([np.arange(pa["n"]) for pa in per_agent])
    cl = np.concatenate([np.full(pa["n"], a) for a, pa in enumerate(per_agent)])
    order = np.lexsort((cl, ids))
    kf_id, kf_client = ids[order].astype(np.int32), cl[order].astype(np.int32)
    K = len(order)
    gidx = -np.ones((A, max(pa["n"] for pa in per_agent)), np.int64)  # (agent, local id) -> global row
    gidx[kf_client, kf_id] = np.arange(K)
    p_true = np.zeros((K, 3)); q_true = np.zeros((K, 4)); v_true = np.zeros((K, 3))
    ba_true = np.zeros((K, 3)); bg_true = np.zeros((K, 3)); kf_time = np.zeros(K)
    for a, pa in enumerate(per_agent):
        rows = gidx[a, :pa["n"]]
        p_true[rows] = pa["p"]; q_true[rows] = pa["R"].as_quat(); v_true[rows] = pa["v"]
        ba_true[rows] = pa["ba"]; bg_true[rows] = pa["bg"]; kf_time[rows] = pa["t"]
    q_true[q_true[:, 3] < 0] *= -1
    R_true = R.from_quat(q_true)
    pred = -np.ones(K, np.int32); succ = -np.ones(K, np.int32)
    for a, pa in enumerate(per_agent):
        rows = gidx[a, :pa["n"]]
        pred[rows[1:]] = rows[:-1]; succ[rows[:-1]] = rows[1:]
    # camera centres / rotations
    Rc_true = R_true * R_sc
    pc_true = p_true + R_true.apply(p_sc)

    # ---- IMU CSR per keyframe (samples between predecessor and this KF)
    imu_ptr = np.zeros(K + 1, np.int64)
    chunks = [None] * K
    imu_first = np.zeros((K, 6))
    for a, pa in enumerate(per_agent):
        rows = gidx[a, :pa["n"]]
        ns = pa["n_sub"]
        dtv = np.full((ns, 1), 1.0 / IMU_RATE)
        for i in range(1, pa["n"]):
            k = rows[i]
            chunks[k] = np.concatenate([dtv, pa["acc"][i - 1, 1:], pa["gyr"][i - 1, 1:]], axis=1)
            imu_first[k, :3] = pa["acc"][i - 1, 0]; imu_first[k, 3:] = pa["gyr"][i - 1, 0]
    for k in range(K):
        imu_ptr[k + 1] = imu_ptr[k] + (0 if chunks[k] is None else len(chunks[k]))
    imu_samples = np.concatenate([c for c in chunks if c is not None]) if K > A else np.zeros((0, 7))

    # ---- loop closures: pairs of spatially close keyframes that LOOK THE SAME WAY (place recognition needs visual
    #      overlap: optical axes within 40 degrees); measurement = noisy ground-truth relative pose
    loop_pairs = []
    axis = Rc_true.apply(np.array([0.0, 0.0, 1.0]))

    def pick(D, ok, n_want, sep):
        """greedy: closest admissible pairs first, at least `sep` keyframes from every pair already taken"""
        used = []
        if n_want <= 0:
            return used
        Dm = np.where(ok, D, np.inf)
        for c in np.argsort(Dm, axis=None):
            if len(used) >= n_want:
                break
            i, j = np.unravel_index(c, D.shape)
            if not np.isfinite(Dm[i, j]):
                break
            if all(abs(i - u[0]) > sep or abs(j - u[1]) > sep for u in used):
                used.append((int(i), int(j)))
        return used

    for a, pa in enumerate(per_agent):
        rows = gidx[a, :pa["n"]]
        n_loops = int(np.floor(pa["n"] / 100.0 * cfg.loops_per_100kf))
        if n_loops and pa["n"] > 60:
            D = np.linalg.norm(pa["p"][:, None, :] - pa["p"][None, :, :], axis=-1)
            ok = (axis[rows] @ axis[rows].T > np.cos(np.deg2rad(40.0))) & (np.arange(pa["n"])[None, :] - np.arange(pa["n"])[:, None] >= 50)
            for i, j in pick(D, ok, n_loops, 20):
                loop_pairs.append((rows[i], rows[j]))
    for a in range(A):
        for b in range(a + 1, A):
            ra, rb = gidx[a, :per_agent[a]["n"]], gidx[b, :per_agent[b]["n"]]
            D = np.linalg.norm(p_true[ra][:, None, :] - p_true[rb][None, :, :], axis=-1)
            ok = axis[ra] @ axis[rb].T > np.cos(np.deg2rad(40.0))
            for i, j in pick(D, ok, cfg.loops_per_pair, 15):
                loop_pairs.append((ra[i], rb[j]))

    # ---- landmarks: spawn per keyframe on the hall surfaces, track in a window, fuse around the loop closures
    nl = cfg.new_lm_per_kf
    u0 = rng.uniform(30, WIDTH - 30, (K, nl)); v0 = rng.uniform(30, HEIGHT - 30, (K, nl))
    xn = (u0 - INTR[2]) / INTR[0]; yn = (v0 - INTR[3]) / INTR[1]  # undistorted approximation is fine for spawning
    ray_c = np.stack([xn, yn, np.ones_like(xn)], -1)
    ray_c /= np.linalg.norm(ray_c, axis=-1, keepdims=True)
    ray_w = np.einsum("kij,knj->kni", Rc_true.as_matrix(), ray_c)
    o = np.clip(pc_true, HALL_MIN + 0.3, HALL_MAX - 0.3)[:, None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        tmax = np.where(ray_w > 0, (HALL_MAX - o) / ray_w, np.where(ray_w < 0, (HALL_MIN - o) / ray_w, np.inf))
    d_wall = np.min(tmax, axis=-1)
    clutter = rng.random((K, nl)) < 0.2
    depth = np.where(clutter, rng.uniform(0.8, 1.0, (K, nl)) * np.minimum(d_wall, 12.0) * rng.uniform(0.3, 1.0, (K, nl)),
                     np.minimum(d_wall, 12.0))
    depth = np.clip(depth, 0.6, 12.0)
    lm_all = (pc_true[:, None, :] + ray_w * depth[..., None]).reshape(-1, 3)
    lm_birth = np.repeat(np.arange(K), nl)
    M = lm_all.shape[0]

    def visible(lm_idx, kf_rows):
        pw = lm_all[lm_idx]
        pcam = Rc_true[kf_rows].inv().apply(pw - pc_true[kf_rows])
        uv, ok = _project(pcam)
        rng_ok = (np.linalg.norm(pcam, axis=1) < 12.5) & (pcam[:, 2] > 0.5)
        return uv, ok & rng_ok

    obs_l, obs_k, obs_uv = [], [], []
    birth_local = kf_id[lm_birth]; birth_agent = kf_client[lm_birth]
    n_agent = np.array([pa["n"] for pa in per_agent])
    for off in range(-cfg.track_window, cfg.track_window + 1):
        loc = birth_local + off
        m = (loc >= 0) & (loc < n_agent[birth_agent])
        li = np.nonzero(m)[0]
        rows = gidx[birth_agent[li], loc[li]]
        uv, ok = visible(li, rows)
        obs_l.append(li[ok]); obs_k.append(rows[ok]); obs_uv.append(uv[ok])
    def add_fused(li, rows):
        """re-observations of landmarks li[] by keyframes rows[] (pairs): visible ones, at most max_fused_obs per landmark"""
        far = (kf_client[rows] != birth_agent[li]) | (np.abs(kf_id[rows] - birth_local[li]) > cfg.track_window)
        li, rows = li[far], rows[far]
        uv, ok = visible(li, rows)
        li, rows, uv = li[ok], rows[ok], uv[ok]
        key = rng.random(len(li))
        o2 = np.lexsort((key, li))
        li, rows, uv = li[o2], rows[o2], uv[o2]
        first = np.concatenate([[True], li[1:] != li[:-1]]) if len(li) else np.zeros(0, bool)
        pos = np.arange(len(li)) - np.maximum.accumulate(np.where(first, np.arange(len(li)), 0))
        keep = pos < cfg.max_fused_obs
        obs_l.append(li[keep]); obs_k.append(rows[keep]); obs_uv.append(uv[keep])

    if cfg.p_fuse > 0 and cfg.fuse_mode == "random" and K > 2 * cfg.track_window:
        fused = np.nonzero(rng.random(M) < cfg.p_fuse)[0]
        for chunk in np.array_split(fused, max(1, len(fused) // 512)):
            if len(chunk):
                add_fused(np.repeat(chunk, K), np.tile(np.arange(K), len(chunk)))
    if cfg.p_fuse > 0 and cfg.fuse_mode == "loops":
        fuse_sel = rng.random(M) < cfg.p_fuse
        seen = set()
        for (k1, k2) in loop_pairs:
            for (ka, kb) in ((k1, k2), (k2, k1)):  # landmarks born around ka, re-observed by the keyframes around kb
                aa, ab = kf_client[ka], kf_client[kb]
                za = np.arange(max(0, kf_id[ka] - cfg.fuse_window), min(n_agent[aa], kf_id[ka] + cfg.fuse_window + 1))
                zb = np.arange(max(0, kf_id[kb] - cfg.fuse_window), min(n_agent[ab], kf_id[kb] + cfg.fuse_window + 1))
                born = np.nonzero(np.isin(lm_birth, gidx[aa, za]) & fuse_sel)[0]
                born = np.array([l for l in born if (l, int(ab)) not in seen], np.int64)  # one fusion per (landmark, agent)
                if len(born) == 0:
                    continue
                seen.update((int(l), int(ab)) for l in born)
                rows_b = gidx[ab, zb]
                add_fused(np.repeat(born, len(rows_b)), np.tile(rows_b, len(born)))
    obs_l = np.concatenate(obs_l); obs_k = np.concatenate(obs_k); obs_uv = np.concatenate(obs_uv)
    # cap observations per keyframe
    key = rng.random(len(obs_l))
    o2 = np.lexsort((key, obs_k))
    obs_l, obs_k, obs_uv = obs_l[o2], obs_k[o2], obs_uv[o2]
    first = np.concatenate([[True], obs_k[1:] != obs_k[:-1]])
    pos = np.arange(len(obs_k)) - np.maximum.accumulate(np.where(first, np.arange(len(obs_k)), 0))
    keep = pos < cfg.max_obs_per_kf
    obs_l, obs_k, obs_uv = obs_l[keep], obs_k[keep], obs_uv[keep]
    # parallax gate: ORB-SLAM3 creates a map point only from rays with cosParallaxRays < 0.9998 (LocalMapping.cc:674-675),
    # so a COVINS map holds no landmark whose observing keyframes all sit (almost) on one ray — e.g. born while the
    # vehicle stands still. Smallest pairwise ray cosine per landmark, in chunks.
    if cfg.min_parallax_cos < 1.0 and len(obs_l):
        o4 = np.lexsort((obs_k, obs_l))
        ol, ok_ = obs_l[o4], obs_k[o4]
        ray = lm_all[ol] - pc_true[ok_]
        ray /= np.linalg.norm(ray, axis=1, keepdims=True)
        cnt0 = np.bincount(ol, minlength=M)
        ptr0 = np.concatenate([[0], np.cumsum(cnt0)])
        within = np.arange(len(ol)) - ptr0[ol]
        maxn = int(cnt0.max())
        mindot = np.ones(M)
        ids = np.nonzero(cnt0 >= 2)[0]
        for c0 in range(0, len(ids), 20000):
            lid = ids[c0:c0 + 20000]
            slot = -np.ones(M, np.int64); slot[lid] = np.arange(len(lid))
            sel = slot[ol] >= 0
            Rp = np.zeros((len(lid), maxn, 3)); msk = np.zeros((len(lid), maxn), bool)
            Rp[slot[ol[sel]], within[sel]] = ray[sel]; msk[slot[ol[sel]], within[sel]] = True
            dots = np.einsum("lid,ljd->lij", Rp, Rp)
            dots[~(msk[:, :, None] & msk[:, None, :])] = 1.0
            mindot[lid] = dots.reshape(len(lid), -1).min(axis=1)
        flat = mindot >= cfg.min_parallax_cos
        keep = ~flat[obs_l]
        obs_l, obs_k, obs_uv = obs_l[keep], obs_k[keep], obs_uv[keep]
    # drop tracks < 2, compact landmark ids, sort by (landmark, kf) like std::map<KeyframePtr,...> iteration
    cnt = np.bincount(obs_l, minlength=M)
    good = cnt >= 2
    sel = good[obs_l]
    obs_l, obs_k, obs_uv = obs_l[sel], obs_k[sel], obs_uv[sel]
    new_id = -np.ones(M, np.int64); new_id[good] = np.arange(good.sum())
    obs_l = new_id[obs_l]
    o3 = np.lexsort((obs_k, obs_l))
    obs_l, obs_k, obs_uv = obs_l[o3], obs_k[o3], obs_uv[o3]
    L = int(good.sum())
    lm_true = lm_all[good]
    lm_ref = lm_birth[good].astype(np.int32)
    lm_obs_ptr = np.concatenate([[0], np.cumsum(np.bincount(obs_l, minlength=L))]).astype(np.int32)
    # measurement noise, gross outliers, float32 storage
    uv_meas = obs_uv + rng.normal(0, cfg.px_noise, obs_uv.shape)
    if cfg.outlier_frac > 0:
        bad = rng.random(len(uv_meas)) < cfg.outlier_frac
        uv_meas[bad] += rng.uniform(-1, 1, (int(bad.sum()), 2)) * 40.0
    uv_meas = uv_meas.astype(np.float32)

    # ---- initial estimate: per-agent VIO-like random-walk drift in yaw and translation
    p_est = p_true.copy(); q_est = q_true.copy(); v_est = v_true.copy()
    yaw_all = np.zeros(K); dp_all = np.zeros((K, 3))
    for a, pa in enumerate(per_agent):
        rows = gidx[a, :pa["n"]]
        ds = np.concatenate([[0], np.linalg.norm(np.diff(pa["p"], axis=0), axis=1)])
        yaw = np.cumsum(rng.normal(0, np.deg2rad(cfg.drift_yaw_deg), pa["n"]) * np.sqrt(ds))
        dpos = np.cumsum(rng.normal(0, cfg.drift_trans, (pa["n"], 3)) * np.sqrt(ds)[:, None], 0)
        yaw_all[rows] = yaw; dp_all[rows] = dpos
    Rz = R.from_euler("z", yaw_all)
    p_est = p_true + dp_all
    q_est = (Rz * R_true).as_quat()
    q_est[q_est[:, 3] < 0] *= -1
    v_est = Rz.apply(v_true) + rng.normal(0, cfg.vel_noise, (K, 3)) * (cfg.vel_noise > 0)
    lm_est = p_est[lm_ref] + Rz[lm_ref].apply(lm_true - p_true[lm_ref]) + rng.normal(0, cfg.lm_noise, (L, 3))
    # bias estimates of a VIO front-end vary as smoothly as the biases themselves (they are filtered against the very
    # random-walk prior the IMU factor encodes): a constant per-agent estimation offset, not white noise per keyframe —
    # white noise of this size would sit 1e4 sigma off the bias-walk prior and dominate the initial cost by 1e5.
    off_a = rng.normal(0, 0.005, (A, 3)) * cfg.imu_noise
    off_g = rng.normal(0, 0.0005, (A, 3)) * cfg.imu_noise
    ba_est = ba_true + off_a[kf_client]
    bg_est = bg_true + off_g[kf_client]

    # ---- loop constraints (typedefs_base.hpp:264-277): noisy ground-truth relative pose per loop closure
    loops: List[LoopConstraint] = []
    for (k1, k2) in loop_pairs:
        Ra, Rb = R_true[k1], R_true[k2]
        dR = R.from_rotvec(rng.normal(0, np.deg2rad(cfg.loop_noise_deg), 3))
        q = (Ra.inv() * Rb * dR).as_quat()
        if q[3] < 0:
            q = -q
        tt = Ra.inv().apply(p_true[k2] - p_true[k1]) + rng.normal(0, cfg.loop_noise_t, 3)
        loops.append(LoopConstraint(int(k1), int(k2), np.concatenate([q, tt])))

    def pose_rows(q, p):
        return np.concatenate([q, p], axis=1)

    extr = np.concatenate([R_sc.as_quat(), p_sc])[None, :].repeat(A, 0)
    m = SlamMap(
        id_map=0, kf_id=kf_id, kf_client=kf_client, kf_time=kf_time,
        kf_invalid=np.zeros(K, bool), kf_loaded=np.zeros(K, bool), kf_gba_optimized=np.zeros(K, bool),
        kf_pose=pose_rows(q_est, p_est), kf_pose_vio=pose_rows(q_est, p_est).copy(),
        kf_velocity=v_est, kf_bias_a=ba_est, kf_bias_g=bg_est, kf_pred=pred, kf_succ=succ,
        kf_cam=kf_client.copy(), cam_extr=extr, cam_intr=np.tile(INTR, (A, 1)), cam_dist=np.tile(DIST, (A, 1)),
        cam_dist_type=np.zeros(A, np.int32),
        cam_imu_calib=np.tile(np.array([SIG_A, SIG_G, SIG_AW, SIG_GW, GRAVITY]), (A, 1)),
        imu_ptr=imu_ptr.astype(np.int64), imu_samples=imu_samples, imu_first=imu_first,
        lm_pos=lm_est, lm_invalid=np.zeros(L, bool), lm_ref_kf=lm_ref, lm_gba_optimized=np.zeros(L, bool),
        lm_obs_ptr=lm_obs_ptr, obs_kf=obs_k.astype(np.int32), obs_uv=uv_meas, obs_octave=np.zeros(len(obs_k), np.int32),
        loops=loops,
        truth=dict(kf_pose=pose_rows(q_true, p_true), kf_velocity=v_true, kf_bias_a=ba_true, kf_bias_g=bg_true, lm_pos=lm_true),
    )
    return m


def ate_rmse(est_xyz: np.ndarray, gt_xyz: np.ndarray, with_scale: bool = False) -> float:
    """Absolute trajectory error after Horn/Umeyama alignment — the logic of
    orb_slam3/evaluation/evaluate_ate_scale.py:50-101 (what `evo_ape -va[s]` computes, docs/run_COVINS.md:110-114)."""
    mu_e, mu_g = est_xyz.mean(0), gt_xyz.mean(0)
    E, Gt = est_xyz - mu_e, gt_xyz - mu_g
    U, S, Vt = np.linalg.svd(Gt.T @ E / len(E))
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1
    Rot = U @ D @ Vt
    s = (S * np.diag(D)).sum() / (E ** 2).sum() * len(E) if with_scale else 1.0
    al = s * (Rot @ E.T).T + mu_g
    return float(np.sqrt(((al - gt_xyz) ** 2).sum(1).mean()))


# BASELINE.json configs -> generator settings
def config_named(name: str, seed: int = 0) -> SynthConfig:
    if name == "mh01":            # configs[0], configs[1]
        return SynthConfig(agents=(1,), seed=seed)
    if name == "mh123":           # configs[2]
        return SynthConfig(agents=(1, 2, 3), seed=seed)
    if name == "mh12345":         # configs[3] and the metric's 5-agent merged map
        return SynthConfig(agents=(1, 2, 3, 4, 5), seed=seed)
    if name == "a12x500":         # configs[4] at reduced scale: 12 agents (5 MH paths + 7 re-posed copies), <= 500 keyframes each
        return SynthConfig(agents=tuple(range(1, 13)), max_kf_per_agent=500, seed=seed)
    if name == "a12":             # configs[4] at its stated size: 12 agents x 1667 keyframes = 20k keyframes, ~2M landmarks, <= 530 obs / keyframe
        return SynthConfig(agents=tuple(range(1, 13)), kf_per_agent=1667, kf_dt=0.08, new_lm_per_kf=150, max_obs_per_kf=530, seed=seed)
    if name == "a12x1000":        # 12 agents x 1000 keyframes
        return SynthConfig(agents=tuple(range(1, 13)), kf_per_agent=1000, kf_dt=0.1, new_lm_per_kf=110, max_obs_per_kf=500, seed=seed)
    if name == "tiny":            # CPU tests
        return SynthConfig(agents=(1, 3), max_kf_per_agent=14, new_lm_per_kf=14, track_window=5, fuse_window=3,
                           loops_per_pair=1, seed=seed)
    if name == "micro":           # tests/golden/refmap: the saved map written by the reference's own cereal code (tools/make_ref_cereal_fixture.py)
        return SynthConfig(agents=(1, 3), max_kf_per_agent=6, new_lm_per_kf=10, track_window=3, fuse_window=2, loops_per_pair=1, seed=7)
    if name == "small":           # GPU parity tests / smoke
        return SynthConfig(agents=(1, 2, 3), max_kf_per_agent=60, new_lm_per_kf=30, track_window=8, seed=seed)
    raise KeyError(name)
