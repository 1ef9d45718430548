"""ctypes mirror of include/covgpu.h — the C-ABI boundary of the GBA / PGO hot path.

`FlatProblem` is the Python-side owner of the flat problem IR: one numpy array per
`covgpu_problem` field, laid out exactly as the header documents. The C++ facade
(include/covins_gpu/optimization_gpu.hpp) fills the same struct from Map/Keyframe/Landmark as
covins_backend/src/covins_backend/optimization_be.cpp:320-557 does for ceres::Problem.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

COVGPU_OK = 0
COVGPU_DOGLEG, COVGPU_LM = 0, 1
COVGPU_DIST_RADTAN, COVGPU_DIST_EQUIDISTANT = 0, 1
COVGPU_MAX_TRACE = 64

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


class Options(C.Structure):
    _fields_ = [
        ("strategy", C.c_int32), ("max_iterations", C.c_int32), ("visual_only", C.c_int32), ("device", C.c_int32),
        ("reproj_loss_a", C.c_double), ("initial_radius", C.c_double), ("max_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("function_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("sigma_a", C.c_double), ("sigma_g", C.c_double), ("sigma_aw", C.c_double), ("sigma_gw", C.c_double),
        ("gravity", C.c_double), ("verbose", C.c_int32), ("reserved", C.c_int32),
    ]


class ProblemStruct(C.Structure):
    _fields_ = [
        ("num_kf", C.c_int32), ("num_cam", C.c_int32), ("num_lm", C.c_int32), ("num_obs", C.c_int32),
        ("num_imu", C.c_int32), ("num_edge", C.c_int32), ("num_imu_samples", C.c_int32), ("reserved", C.c_int32),
        ("kf_pose", _dp), ("kf_speed_bias", _dp), ("kf_fixed", _bp), ("kf_cam", _ip),
        ("cam_extr", _dp), ("cam_intr", _dp), ("cam_dist", _dp), ("cam_dist_type", _ip),
        ("lm_pos", _dp), ("lm_obs_ptr", _ip), ("obs_kf", _ip), ("obs_uv", _dp), ("obs_sigma", _dp),
        ("imu_kf_i", _ip), ("imu_kf_j", _ip), ("imu_sample_ptr", _ip), ("imu_samples", _dp), ("imu_first", _dp), ("imu_noise", _dp),
        ("edge_i", _ip), ("edge_j", _ip), ("edge_meas", _dp), ("edge_sqrt_info", _dp), ("edge_loss_a", _dp),
    ]


class RelposeBatch(C.Structure):
    _fields_ = [("num_pairs", C.c_int32), ("corr_ptr", _ip), ("p_b", _dp), ("p_a", _dp), ("kp_a", _dp), ("kp_b", _dp), ("sigma_a", _dp),
                ("sigma_b", _dp), ("cam_a", _dp), ("cam_b", _dp), ("dist_type_a", _ip), ("dist_type_b", _ip), ("T_ab", _dp), ("outlier", _bp),
                ("inliers", _ip)]


class Result(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("accepted", C.c_int32), ("termination", C.c_int32), ("reserved", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("t_upload_s", C.c_double),
        ("t_solve_s", C.c_double), ("t_download_s", C.c_double), ("t_linear_solve_s", C.c_double),
        ("cost_trace", C.c_double * COVGPU_MAX_TRACE), ("radius_trace", C.c_double * COVGPU_MAX_TRACE),
        ("accepted_trace", C.c_int32 * COVGPU_MAX_TRACE),
    ]


class TwoRound(C.Structure):
    """covgpu_two_round (include/covgpu.h): parameters of the second round of a GlobalBundleAdjustment call."""
    _fields_ = [("outlier_threshold", C.c_double), ("round1_iterations", C.c_int32), ("use_loops_round2", C.c_int32),
                ("loop_loss_round2", C.c_double), ("kf_fixed_round2", C.POINTER(C.c_uint8))]


def _f64(a, shape):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))
    return a


def _i32(a, shape):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(shape))


@dataclass
class FlatProblem:
    """Flat IR of one GBA / PGO call. Arrays are float64 / int32 / uint8, C-contiguous."""
    kf_pose: np.ndarray                 # [K,7]
    kf_speed_bias: np.ndarray           # [K,9]
    kf_fixed: np.ndarray                # [K] uint8
    kf_cam: np.ndarray                  # [K]
    cam_extr: np.ndarray                # [A,7]
    cam_intr: np.ndarray                # [A,4]
    cam_dist: np.ndarray                # [A,4]
    cam_dist_type: np.ndarray           # [A]
    lm_pos: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    lm_obs_ptr: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    obs_kf: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    obs_uv: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    obs_sigma: np.ndarray = field(default_factory=lambda: np.zeros(0))
    imu_kf_i: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    imu_kf_j: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    imu_sample_ptr: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    imu_samples: np.ndarray = field(default_factory=lambda: np.zeros((0, 7)))
    imu_first: np.ndarray = field(default_factory=lambda: np.zeros((0, 6)))
    imu_noise: Optional[np.ndarray] = None   # [I,5] sigma_a sigma_g sigma_aw sigma_gw gravity per factor; None -> options
    edge_i: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    edge_j: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    edge_meas: np.ndarray = field(default_factory=lambda: np.zeros((0, 7)))
    edge_sqrt_info: np.ndarray = field(default_factory=lambda: np.zeros((0, 36)))
    edge_loss_a: np.ndarray = field(default_factory=lambda: np.zeros(0))

    def __post_init__(self):
        K = np.asarray(self.kf_pose).reshape(-1, 7).shape[0]
        A = np.asarray(self.cam_extr).reshape(-1, 7).shape[0]
        self.kf_pose = _f64(self.kf_pose, (K, 7))
        self.kf_speed_bias = _f64(self.kf_speed_bias, (K, 9))
        self.kf_fixed = np.ascontiguousarray(np.asarray(self.kf_fixed, dtype=np.uint8).reshape(K))
        self.kf_cam = _i32(self.kf_cam, (K,))
        self.cam_extr = _f64(self.cam_extr, (A, 7))
        self.cam_intr = _f64(self.cam_intr, (A, 4))
        self.cam_dist = _f64(self.cam_dist, (A, 4))
        self.cam_dist_type = _i32(self.cam_dist_type, (A,))
        self.lm_pos = _f64(self.lm_pos, (-1, 3))
        self.lm_obs_ptr = _i32(self.lm_obs_ptr, (-1,))
        self.obs_kf = _i32(self.obs_kf, (-1,))
        self.obs_uv = _f64(self.obs_uv, (-1, 2))
        self.obs_sigma = _f64(self.obs_sigma, (-1,))
        self.imu_kf_i = _i32(self.imu_kf_i, (-1,))
        self.imu_kf_j = _i32(self.imu_kf_j, (-1,))
        self.imu_sample_ptr = _i32(self.imu_sample_ptr, (-1,))
        self.imu_samples = _f64(self.imu_samples, (-1, 7))
        self.imu_first = _f64(self.imu_first, (-1, 6))
        if self.imu_noise is not None:
            self.imu_noise = _f64(self.imu_noise, (-1, 5))
        self.edge_i = _i32(self.edge_i, (-1,))
        self.edge_j = _i32(self.edge_j, (-1,))
        self.edge_meas = _f64(self.edge_meas, (-1, 7))
        self.edge_sqrt_info = _f64(self.edge_sqrt_info, (-1, 36))
        self.edge_loss_a = _f64(self.edge_loss_a, (-1,))
        self.validate()

    # sizes
    @property
    def K(self): return self.kf_pose.shape[0]
    @property
    def A(self): return self.cam_extr.shape[0]
    @property
    def L(self): return self.lm_pos.shape[0]
    @property
    def O(self): return self.obs_kf.shape[0]
    @property
    def I(self): return self.imu_kf_i.shape[0]
    @property
    def E(self): return self.edge_i.shape[0]

    def validate(self):
        K, L, O, I, E = self.K, self.L, self.O, self.I, self.E
        assert self.lm_obs_ptr.shape[0] == L + 1 and self.lm_obs_ptr[0] == 0 and self.lm_obs_ptr[-1] == O
        assert np.all(np.diff(self.lm_obs_ptr) >= 0)
        assert self.obs_uv.shape[0] == O and self.obs_sigma.shape[0] == O
        if O: assert self.obs_kf.min() >= 0 and self.obs_kf.max() < K
        assert self.kf_cam.min() >= 0 and self.kf_cam.max() < self.A
        assert self.imu_kf_j.shape[0] == I and self.imu_sample_ptr.shape[0] == I + 1 and self.imu_first.shape[0] == I
        assert self.imu_sample_ptr[-1] == self.imu_samples.shape[0]
        assert self.imu_noise is None or self.imu_noise.shape[0] == I
        if I: assert min(self.imu_kf_i.min(), self.imu_kf_j.min()) >= 0 and max(self.imu_kf_i.max(), self.imu_kf_j.max()) < K
        assert self.edge_j.shape[0] == E and self.edge_meas.shape[0] == E and self.edge_sqrt_info.shape[0] == E
        assert self.edge_loss_a.shape[0] == E
        if E: assert min(self.edge_i.min(), self.edge_j.min()) >= 0 and max(self.edge_i.max(), self.edge_j.max()) < K

    def copy(self) -> "FlatProblem":
        return FlatProblem(**{k: (None if v is None else np.array(v, copy=True)) for k, v in self.__dict__.items()})

    def as_struct(self) -> ProblemStruct:
        """C view of the arrays (no copies; keep `self` alive while the struct is in use)."""
        s = ProblemStruct()
        s.num_kf, s.num_cam, s.num_lm, s.num_obs = self.K, self.A, self.L, self.O
        s.num_imu, s.num_edge, s.num_imu_samples = self.I, self.E, self.imu_samples.shape[0]
        for name, ctype in ProblemStruct._fields_:
            if name.startswith("num_") or name == "reserved":
                continue
            arr = getattr(self, name)
            setattr(s, name, None if arr is None else arr.ctypes.data_as(ctype))
        return s


def declare(lib: C.CDLL, prefix: str) -> None:
    """Attach argtypes/restype for the entry points shared by libcovgpu (prefix 'covgpu_', with a context
    argument) and — test side only — the oracle (prefix 'covo_', no context)."""
    ctx = [C.c_void_p] if prefix == "covgpu_" else []
    OP, PP, RP = C.POINTER(Options), C.POINTER(ProblemStruct), C.POINTER(Result)

    def d(name, args, res=C.c_int):
        fn = getattr(lib, prefix + name)
        fn.argtypes, fn.restype = args, res

    d("default_options", [OP], None)
    d("gba_solve", ctx + [OP, PP, RP])
    d("pgo_solve", ctx + [OP, PP, RP])
    d("reprojection_residual_norms", ctx + [OP, PP, _dp])
    d("linearize_reprojection", ctx + [OP, PP, _dp, _dp, _dp, _dp])
    d("preintegrate", ctx + [OP, PP, _dp, _dp, _dp])
    d("linearize_imu", ctx + [OP, PP, _dp, _dp])
    d("linearize_between", ctx + [OP, PP, _dp, _dp, _dp])
    d("pgo_reanchor", ctx + [C.c_int32, _dp, _dp, _dp, C.c_int32, _ip, _dp])
    d("reduced_dim", [OP, PP], C.c_int32)
    if prefix == "covgpu_":
        d("create", [OP, C.POINTER(C.c_void_p)])
        d("destroy", [C.c_void_p], None)
        d("last_error", [], C.c_char_p)
        d("upload", [C.c_void_p, OP, PP])
        d("solve_resident", [C.c_void_p, OP, RP])
        d("download", [C.c_void_p, PP])
        d("schur", [C.c_void_p, OP, PP, C.c_double, _dp, _dp, _dp])
        d("solve_reduced", [C.c_void_p, C.c_int32, _dp, _dp, _dp])
        d("pgo_partition", [C.c_int32, C.c_int32, _ip, _ip, _ip], C.c_int32)
        d("gn_step", [C.c_void_p, OP, PP, C.c_double, _dp, _dp, _dp])
        d("relpose_batch", [C.c_void_p, C.POINTER(RelposeBatch), C.c_double, C.c_int32])
        d("outlier_pass", [C.c_void_p, C.c_double, _bp, _ip, C.POINTER(C.c_int64)])
        d("covisibility", [C.c_void_p, C.c_int32, C.c_int64, _ip, _ip, _ip, C.POINTER(C.c_int64)])
        d("gba_solve_multi", [OP, PP, RP, C.c_int32, _ip, C.c_double, _bp, _ip, C.POINTER(C.c_int64)])
        d("gba_two_round_multi", [OP, PP, C.POINTER(TwoRound), C.c_int32, _ip, _bp, _ip, C.POINTER(C.c_int64), RP, RP])
        d("nd_plan_create", [OP, PP, C.c_int32, C.POINTER(C.c_void_p)])
        d("nd_plan_create_pgo", [OP, PP, C.c_int32, C.POINTER(C.c_void_p)])
        d("nd_plan_destroy", [C.c_void_p], None)
        d("nd_plan_info", [C.c_void_p, C.POINTER(C.c_int64)], None)
        d("nd_plan_arrays", [C.c_void_p, _ip, _ip, _ip, _ip, _ip, _ip], None)
        d("shard_plan", [OP, PP, C.c_int32, C.POINTER(C.c_void_p), _ip, _ip, _ip], C.c_int32)
        d("nd_plan_owner", [C.c_void_p, _ip, _ip], None)
        d("nd_plan_ranks", [C.c_void_p, _ip], None)
        d("group_create", [C.c_int32, C.POINTER(C.c_void_p)])
        d("group_destroy", [C.c_void_p], None)
        d("group_abort", [C.c_void_p], None)
        d("set_shard_group", [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p])
        d("rccl_unique_id", [_bp])
        d("set_shard_rccl", [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, _bp])
        d("set_shard_none", [C.c_void_p])
        d("allreduce_host", [C.c_void_p, _dp, C.c_int64, C.c_int32])
        d("shard_stats", [C.c_void_p, C.POINTER(C.c_int64)], None)


def dptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a: np.ndarray):
    return a.ctypes.data_as(_ip)
