"""Reader (and, for fixtures, writer) of saved COVINS maps — SURVEY.md §8f rank 1.

`Map::SaveToFile` (covins_backend/src/covins_backend/map_be.cpp:813-922) writes one cereal *binary* archive per keyframe
(`keyframes/keyframesN.txt`), per landmark (`mappoints/mappointsN.txt`) and one `mapdata.txt` with the loop constraints;
`Map::LoadFromFile` (map_be.cpp:508-696) reads them back through `MsgKeyframe(true)` / `MsgLandmark(true)` / `MsgMap`.
This module decodes that byte stream WITHOUT cereal / OpenCV / Eigen / aslam and produces the `SlamMap` the optimiser's
host side flattens, so that GBA / PGO can run on a real saved map (BASELINE configs[0]: "GBA on loaded covins_backend map").

Byte layout (cereal::BinaryOutputArchive = raw little-endian, no tags; covins_backend/thirdparty/cereal):
  arithmetic T            sizeof(T) bytes (bool = 1, int = 4, size_t = 8, enum = underlying int = 4)
  std::vector<arith>      u64 count, then the elements back to back
  std::vector<other>      u64 count, then every element in turn
  std::pair<A,B>          A, B                      std::map<K,V>   u64 count, then K, V per entry (key order)
  Eigen::Matrix           i32 rows, i32 cols, rows*cols scalars in COLUMN-major order   (msg_keyframe.hpp:207-235)
  cv::Mat                 i32 rows, cols, type, bool continuous, then rows*cols*elemSize bytes   (msg_keyframe.hpp:237-283)
Field order: MsgKeyframe save_to_file branch (msg_keyframe.hpp:129-146), VICalibration::serialize (typedefs_base.hpp:376-380),
PreintegrationData::serialize (msg_keyframe.hpp:24-41), MsgLandmark save_to_file branch (msg_landmark.hpp:66-71),
MsgMap::serialize (map_be.hpp:126-136).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Tuple

import numpy as np

from .mapdata import LoopConstraint, SlamMap

DEFPAIR = (65535, 255)   # typedefs_base.hpp:51-56: defpair = (KFRANGE = uint16 max, MAPRANGE = uint8 max) marks "no keyframe"
_CV_ELEM = {0: 1, 1: 1, 2: 2, 3: 2, 4: 4, 5: 4, 6: 8}   # bytes per channel of CV_8U .. CV_64F


class Reader:
    def __init__(self, data: bytes):
        self.b, self.o = memoryview(data), 0

    def _take(self, n):
        if self.o + n > len(self.b):
            raise ValueError("truncated archive")
        v = self.b[self.o:self.o + n]; self.o += n
        return v

    def u64(self): return struct.unpack("<Q", self._take(8))[0]
    def i32(self): return struct.unpack("<i", self._take(4))[0]
    def f64(self): return struct.unpack("<d", self._take(8))[0]
    def boolean(self): return self._take(1)[0] != 0
    def idpair(self): return (self.u64(), self.u64())

    def eigen(self, dtype=np.float64) -> np.ndarray:
        r, c = self.i32(), self.i32()
        if r < 0 or c < 0 or r * c > 1 << 28:
            raise ValueError(f"implausible Eigen shape {r}x{c}")
        a = np.frombuffer(self._take(r * c * np.dtype(dtype).itemsize), dtype=dtype)
        return a.reshape(c, r).T.copy()   # column-major on disk

    def vec(self, dtype=np.float64) -> np.ndarray:
        n = self.u64()
        return np.frombuffer(self._take(n * np.dtype(dtype).itemsize), dtype=dtype).copy()

    def eigen_vec(self, dtype, rows) -> np.ndarray:
        """std::vector<Eigen::Matrix<dtype, rows, 1>>: count, then (i32 rows, i32 cols, data) per element."""
        n = self.u64()
        item = 8 + rows * np.dtype(dtype).itemsize
        raw = np.frombuffer(self._take(n * item), dtype=np.uint8).reshape(n, item)
        if n and (raw[:, :8].view("<i4") != [rows, 1]).any():
            raise ValueError("unexpected element shape in a vector of fixed-size Eigen vectors")
        return raw[:, 8:].copy().view(dtype).reshape(n, rows)

    def cvmat(self) -> np.ndarray:
        rows, cols, typ, cont = self.i32(), self.i32(), self.i32(), self.boolean()
        esz = _CV_ELEM.get(typ & 7, 1) * ((typ >> 3) + 1)
        data = self._take(rows * cols * esz)   # continuous or row by row: the same bytes either way
        return np.frombuffer(data, dtype=np.uint8).reshape(rows, cols * esz).copy() if rows * cols else np.zeros((0, 0), np.uint8)

    def done(self): return self.o == len(self.b)


def read_calibration(r: Reader) -> dict:
    c = dict(T_SC=r.eigen(), cam_model=r.i32(), dist_model=r.i32(), img_dims=r.eigen(), dist_coeffs=r.eigen().reshape(-1),
             intrinsics=r.eigen().reshape(-1), K=r.eigen())
    for k in ("a_max", "g_max", "sigma_a_c", "sigma_g_c", "sigma_ba", "sigma_bg", "sigma_aw_c", "sigma_gw_c", "tau", "g"):
        c[k] = r.f64()
    c["a0"] = r.eigen().reshape(-1); c["rate"] = r.i32(); c["delay_cam0_to_imu"] = r.f64(); c["delay_cam1_to_imu"] = r.f64()
    return c


def read_keyframe(data: bytes) -> dict:
    """MsgKeyframe, save_to_file branch (msg_keyframe.hpp:129-146)."""
    r = Reader(data)
    k = dict(timestamp=r.f64(), id=r.idpair(), calibration=read_calibration(r))
    k["img_dims"] = [r.i32() for _ in range(4)]
    k["keypoints_distorted"] = r.eigen_vec(np.float32, 2); k["keypoints_undistorted"] = r.eigen_vec(np.float32, 2)
    k["keypoints_aors"] = r.eigen_vec(np.float32, 4); k["descriptors"] = r.cvmat()
    k["keypoints_distorted_add"] = r.eigen_vec(np.float32, 2); k["keypoints_undistorted_add"] = r.eigen_vec(np.float32, 2)
    k["keypoints_aors_add"] = r.eigen_vec(np.float32, 4); k["descriptors_add"] = r.cvmat()
    for f in ("T_s_c", "T_w_s", "T_w_s_vio"):
        k[f] = r.eigen()
    for f in ("velocity", "bias_gyro", "bias_accel", "lin_acc", "ang_vel", "lin_acc_init", "ang_vel_init"):
        k[f] = r.eigen().reshape(-1)
    p = dict(acc=r.eigen().reshape(-1), gyr=r.eigen().reshape(-1), lin_bias_accel=r.eigen().reshape(-1), lin_bias_gyro=r.eigen().reshape(-1))
    for f in ("dt", "lin_acc_x", "lin_acc_y", "lin_acc_z", "ang_vel_x", "ang_vel_y", "ang_vel_z"):
        p[f] = r.vec()
    k["preintegration"] = p
    n = r.u64()
    k["landmarks"] = {r.i32(): r.idpair() for _ in range(n)}   # feature index -> landmark id
    k["id_predecessor"] = r.idpair(); k["id_successor"] = r.idpair()
    k["img"] = r.cvmat()
    if not r.done():
        raise ValueError(f"{len(r.b) - r.o} trailing bytes after MsgKeyframe")
    return k


def read_landmark(data: bytes) -> dict:
    """MsgLandmark, save_to_file branch (msg_landmark.hpp:66-71)."""
    r = Reader(data)
    lm = dict(id=r.idpair(), pos_w=r.eigen().reshape(-1))
    n = r.u64()
    lm["observations"] = [(r.idpair(), r.i32()) for _ in range(n)]   # keyframe id -> feature index, key order
    lm["id_reference"] = r.idpair()
    if not r.done():
        raise ValueError("trailing bytes after MsgLandmark")
    return lm


def read_mapdata(data: bytes) -> dict:
    """MsgMap (map_be.hpp:126-136): loop constraints of the map."""
    r = Reader(data)
    m = dict(id_map=r.u64())
    m["keyframes1"] = [r.idpair() for _ in range(r.u64())]
    m["keyframes2"] = [r.idpair() for _ in range(r.u64())]
    m["transforms12"] = [r.eigen() for _ in range(r.u64())]
    m["cov"] = [r.eigen() for _ in range(r.u64())]
    return m


def _pose_row(T: np.ndarray) -> np.ndarray:
    from scipy.spatial.transform import Rotation as R
    q = R.from_matrix(T[:3, :3]).as_quat()
    if q[3] < 0:
        q = -q
    return np.concatenate([q, T[:3, 3]])


def _pose_mat(p: np.ndarray) -> np.ndarray:
    from scipy.spatial.transform import Rotation as R
    T = np.eye(4); T[:3, :3] = R.from_quat(p[:4]).as_matrix(); T[:3, 3] = p[4:]
    return T


def load_map(path: str) -> SlamMap:
    """What Map::LoadFromFile builds (map_be.cpp:508-696), as the struct-of-arrays SlamMap: keyframes sorted by
    (id, client) (typedefs_base.hpp:178), predecessor / successor links, raw IMU samples per keyframe (samples with dt == 0
    skipped like keyframe_be.cpp:199-202), landmark observations with the keypoint each one refers to, reference keyframes,
    loop constraints; every keyframe is flagged is_loaded_ (map_be.cpp:583)."""
    def files(sub):
        d = os.path.join(path, sub)
        return sorted(os.path.join(d, f) for f in os.listdir(d)) if os.path.isdir(d) else []
    kfs = [read_keyframe(open(f, "rb").read()) for f in files("keyframes")]
    lms = [read_landmark(open(f, "rb").read()) for f in files("mappoints")]
    md = read_mapdata(open(os.path.join(path, "mapdata.txt"), "rb").read())
    kfs.sort(key=lambda k: k["id"])
    lms.sort(key=lambda l: l["id"])
    K = len(kfs)
    row = {k["id"]: i for i, k in enumerate(kfs)}
    # cameras: one row per distinct (client, calibration)
    cams: List[Tuple] = []
    cam_rows: Dict[Tuple, int] = {}
    kf_cam = np.zeros(K, np.int32)
    for i, k in enumerate(kfs):
        c = k["calibration"]
        # KeyframeBase's constructor exits on anything but pinhole / omni x radtan / equidistant (keyframe_base.cpp:58-82); omni
        # needs a 5-parameter unified-projection camera this back-end does not implement: refuse, do not
        # silently flatten such a map as pinhole + radtan
        if c["cam_model"] != 0:
            raise ValueError(f"keyframe {k['id']}: camera model {c['cam_model']} is not PINHOLE(0): unsupported")
        if c["dist_model"] not in (0, 1):
            raise ValueError(f"keyframe {k['id']}: distortion model {c['dist_model']} is neither RADTAN(0) nor EQUI(1): unsupported")
        key = (k["id"][1], c["dist_model"], tuple(c["intrinsics"]), tuple(c["dist_coeffs"][:4]), tuple(k["T_s_c"].reshape(-1)),
               c["sigma_a_c"], c["sigma_g_c"], c["sigma_aw_c"], c["sigma_gw_c"], c["g"])
        if key not in cam_rows:
            cam_rows[key] = len(cams); cams.append((k, c))
        kf_cam[i] = cam_rows[key]
    A = len(cams)
    imu_ptr = np.zeros(K + 1, np.int64)
    chunks = []
    for i, k in enumerate(kfs):
        p = k["preintegration"]
        s = np.stack([p["dt"], p["lin_acc_x"], p["lin_acc_y"], p["lin_acc_z"], p["ang_vel_x"], p["ang_vel_y"], p["ang_vel_z"]], 1) if len(p["dt"]) else np.zeros((0, 7))
        s = s[s[:, 0] != 0.0]
        chunks.append(s); imu_ptr[i + 1] = imu_ptr[i] + len(s)
    obs_kf, obs_uv, obs_oct, ptr = [], [], [], [0]
    for lm in lms:
        for kid, feat in lm["observations"]:
            i = row.get(kid)
            if i is None:
                continue   # "if(!kf) continue" (map_be.cpp:655-657)
            obs_kf.append(i); obs_uv.append(kfs[i]["keypoints_distorted"][feat]); obs_oct.append(int(kfs[i]["keypoints_aors"][feat][1]))
        ptr.append(len(obs_kf))
    loops = []
    for a, b, T, cov in zip(md["keyframes1"], md["keyframes2"], md["transforms12"], md["cov"]):
        if a in row and b in row:
            loops.append(LoopConstraint(row[a], row[b], _pose_row(T), cov))
    link = lambda ids: np.array([row.get(i, -1) if i != DEFPAIR else -1 for i in ids], np.int32)
    return SlamMap(
        id_map=int(md["id_map"]), kf_id=np.array([k["id"][0] for k in kfs], np.int32), kf_client=np.array([k["id"][1] for k in kfs], np.int32),
        kf_time=np.array([k["timestamp"] for k in kfs]), kf_invalid=np.zeros(K, bool), kf_loaded=np.ones(K, bool),
        kf_gba_optimized=np.zeros(K, bool), kf_pose=np.array([_pose_row(k["T_w_s"]) for k in kfs]).reshape(K, 7),
        kf_pose_vio=np.array([_pose_row(k["T_w_s_vio"]) for k in kfs]).reshape(K, 7),
        kf_velocity=np.array([k["velocity"] for k in kfs]).reshape(K, 3), kf_bias_a=np.array([k["bias_accel"] for k in kfs]).reshape(K, 3),
        kf_bias_g=np.array([k["bias_gyro"] for k in kfs]).reshape(K, 3),
        kf_pred=link([k["id_predecessor"] for k in kfs]), kf_succ=link([k["id_successor"] for k in kfs]), kf_cam=kf_cam,
        cam_extr=np.array([_pose_row(k["T_s_c"]) for k, _ in cams]).reshape(A, 7),
        cam_intr=np.array([c["intrinsics"][:4] for _, c in cams]).reshape(A, 4),
        cam_dist=np.array([np.pad(c["dist_coeffs"], (0, 4))[:4] for _, c in cams]).reshape(A, 4),
        cam_dist_type=np.array([c["dist_model"] for _, c in cams], np.int32),
        cam_imu_calib=np.array([[c["sigma_a_c"], c["sigma_g_c"], c["sigma_aw_c"], c["sigma_gw_c"], c["g"]] for _, c in cams]).reshape(A, 5),
        imu_ptr=imu_ptr, imu_samples=np.concatenate(chunks) if chunks else np.zeros((0, 7)),
        imu_first=np.array([np.concatenate([k["lin_acc_init"], k["ang_vel_init"]]) for k in kfs]).reshape(K, 6),
        lm_pos=np.array([l["pos_w"] for l in lms]).reshape(len(lms), 3), lm_invalid=np.zeros(len(lms), bool),
        lm_ref_kf=np.array([row.get(l["id_reference"], -1) for l in lms], np.int32), lm_gba_optimized=np.zeros(len(lms), bool),
        lm_obs_ptr=np.array(ptr, np.int32), obs_kf=np.array(obs_kf, np.int32), obs_uv=np.array(obs_uv, np.float32).reshape(-1, 2),
        obs_octave=np.array(obs_oct, np.int32), loops=loops)


# ------------------------------------------------------------------------------------------------ writer (fixtures / tests)
class Writer:
    def __init__(self): self.parts: List[bytes] = []
    def u64(self, v): self.parts.append(struct.pack("<Q", v))
    def i32(self, v): self.parts.append(struct.pack("<i", v))
    def f64(self, v): self.parts.append(struct.pack("<d", v))
    def boolean(self, v): self.parts.append(b"\x01" if v else b"\x00")
    def idpair(self, p): self.u64(p[0]); self.u64(p[1])

    def eigen(self, a, dtype=np.float64):
        a = np.atleast_2d(np.asarray(a, dtype=dtype))
        self.i32(a.shape[0]); self.i32(a.shape[1]); self.parts.append(np.asfortranarray(a).tobytes(order="F"))

    def colvec(self, v, dtype=np.float64): self.eigen(np.asarray(v, dtype=dtype).reshape(-1, 1), dtype)
    def vec(self, v, dtype=np.float64): v = np.asarray(v, dtype=dtype); self.u64(len(v)); self.parts.append(v.tobytes())

    def eigen_vec(self, a, dtype):
        a = np.asarray(a, dtype=dtype); self.u64(len(a))
        for row_ in a:
            self.colvec(row_, dtype)

    def cvmat(self, m):
        m = np.asarray(m, np.uint8).reshape(len(m), -1) if len(m) else np.zeros((0, 0), np.uint8)
        self.i32(m.shape[0]); self.i32(m.shape[1]); self.i32(0); self.boolean(True); self.parts.append(m.tobytes())

    def bytes(self): return b"".join(self.parts)


def save_map(path: str, m: SlamMap) -> None:
    """Writes `m` in the Map::SaveToFile layout (map_be.cpp:813-922). Used to make fixtures for the reader; every keyframe's
    keypoint list holds exactly its observations (feature index = order of appearance)."""
    os.makedirs(os.path.join(path, "keyframes")); os.makedirs(os.path.join(path, "mappoints"))
    ids = [(int(a), int(b)) for a, b in zip(m.kf_id, m.kf_client)]
    feats: List[List[int]] = [[] for _ in range(m.K)]          # observation rows per keyframe, in feature order
    obs_lm = np.repeat(np.arange(m.L), np.diff(m.lm_obs_ptr))
    feat_of_obs = np.zeros(m.O, np.int32)
    for o, k in enumerate(m.obs_kf):
        feat_of_obs[o] = len(feats[k]); feats[k].append(o)
    for i in range(m.K):
        w = Writer()
        w.f64(float(m.kf_time[i])); w.idpair(ids[i])
        cam = int(m.kf_cam[i]); ic = m.cam_imu_calib[cam]
        Tsc = _pose_mat(m.cam_extr[cam])
        w.eigen(Tsc); w.i32(0); w.i32(int(m.cam_dist_type[cam])); w.colvec([752.0, 480.0]); w.colvec(m.cam_dist[cam]); w.colvec(m.cam_intr[cam])
        fx, fy, cx, cy = m.cam_intr[cam]
        w.eigen(np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]]))
        for v in (176.0, 7.8, ic[0], ic[1], 0.0, 0.0, ic[2], ic[3], 3600.0, ic[4]):
            w.f64(float(v))
        w.colvec([0.0, 0.0, 0.0]); w.i32(200); w.f64(0.0); w.f64(0.0)
        for v in (0, 0, 752, 480):
            w.i32(v)
        uv = m.obs_uv[feats[i]] if feats[i] else np.zeros((0, 2), np.float32)
        aors = np.zeros((len(feats[i]), 4), np.float32)
        if feats[i]:
            aors[:, 1] = m.obs_octave[feats[i]]
        w.eigen_vec(uv, np.float32); w.eigen_vec(uv, np.float32); w.eigen_vec(aors, np.float32); w.cvmat(np.zeros((len(feats[i]), 32), np.uint8))
        w.eigen_vec(np.zeros((0, 2)), np.float32); w.eigen_vec(np.zeros((0, 2)), np.float32); w.eigen_vec(np.zeros((0, 4)), np.float32); w.cvmat([])
        w.eigen(Tsc); w.eigen(_pose_mat(m.kf_pose[i])); w.eigen(_pose_mat(m.kf_pose_vio[i]))
        s = m.imu_samples[m.imu_ptr[i]:m.imu_ptr[i + 1]]
        last = s[-1] if len(s) else np.zeros(7)
        for v in (m.kf_velocity[i], m.kf_bias_g[i], m.kf_bias_a[i], last[1:4], last[4:7], m.imu_first[i, :3], m.imu_first[i, 3:]):
            w.colvec(v)
        for v in (last[1:4], last[4:7], m.kf_bias_a[i], m.kf_bias_g[i]):
            w.colvec(v)
        for c in range(7):
            w.vec(s[:, c] if len(s) else [])
        w.u64(len(feats[i]))
        for f, o in enumerate(feats[i]):
            w.i32(f); w.idpair((int(obs_lm[o]), 0))
        w.idpair(ids[m.kf_pred[i]] if m.kf_pred[i] >= 0 else DEFPAIR); w.idpair(ids[m.kf_succ[i]] if m.kf_succ[i] >= 0 else DEFPAIR)
        w.cvmat([])
        open(os.path.join(path, "keyframes", f"keyframes{i}.txt"), "wb").write(w.bytes())
    for l in range(m.L):
        o0, o1 = m.lm_obs_ptr[l], m.lm_obs_ptr[l + 1]
        if o1 - o0 < 2 or m.lm_ref_kf[l] < 0:
            continue   # SaveToFile skips them (map_be.cpp:879-884)
        w = Writer()
        w.idpair((l, 0)); w.colvec(m.lm_pos[l])
        ent = sorted((ids[m.obs_kf[o]], int(feat_of_obs[o])) for o in range(o0, o1))   # std::map<idpair,int> key order
        w.u64(len(ent))
        for kid, f in ent:
            w.idpair(kid); w.i32(f)
        w.idpair(ids[m.lm_ref_kf[l]])
        open(os.path.join(path, "mappoints", f"mappoints{l}.txt"), "wb").write(w.bytes())
    w = Writer()
    w.u64(int(m.id_map))
    for sel in (lambda lc: ids[lc.kf1], lambda lc: ids[lc.kf2]):
        w.u64(len(m.loops))
        for lc in m.loops:
            w.idpair(sel(lc))
    w.u64(len(m.loops))
    for lc in m.loops:
        w.eigen(_pose_mat(lc.T_s1_s2))
    w.u64(len(m.loops))
    for lc in m.loops:
        w.eigen(lc.cov)
    open(os.path.join(path, "mapdata.txt"), "wb").write(w.bytes())
