"""Builds and drives tests/cpp/libfacade_shim.so: the C++ facade (include/covins_gpu/optimization_gpu.hpp)
instantiated on the stand-in Map/Keyframe/Landmark classes."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "cpp", "libfacade_shim.so")
        srcs = [os.path.join(HERE, "cpp", f) for f in ("facade_shim.cpp", "standin_map.hpp")] + \
               [os.path.join(ROOT, "include", "covins_gpu", "optimization_gpu.hpp"), os.path.join(ROOT, "include", "covgpu.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", srcs[0], "-o", so, "-L" + os.path.join(ROOT, "covins_amd"),
                                   "-lcovgpu", "-Wl,-rpath," + os.path.join(ROOT, "covins_amd")])
        _LIB = C.CDLL(so)
        _LIB.shim_build.restype = C.c_void_p
        _LIB.shim_free.argtypes = [C.c_void_p]
        _LIB.shim_gba.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        _LIB.shim_set_params.argtypes = [C.c_int, C.c_char_p]
        _LIB.shim_set_flatten_threads.argtypes = [C.c_int]
        _LIB.shim_set_gpus.argtypes = [C.c_int, C.c_int]
        _LIB.shim_set_invalid.argtypes = [C.c_void_p, C.c_int]
        _LIB.shim_set_device_second_round.argtypes = [C.c_int]
        _LIB.shim_set_device_clean.argtypes = [C.c_int]
        _LIB.shim_shutdown.argtypes = []
    return _LIB


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


class StandinMap:
    def __init__(self, m):
        self.m = m
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        NL = len(m.loops)
        self._keep = dict(
            kf_id=i32(m.kf_id), kf_client=i32(m.kf_client), inv=u8(m.kf_invalid), loaded=u8(m.kf_loaded), gba=u8(m.kf_gba_optimized),
            pose=f64(m.kf_pose), vio=f64(m.kf_pose_vio), vel=f64(m.kf_velocity), ba=f64(m.kf_bias_a), bg=f64(m.kf_bias_g),
            pred=i32(m.kf_pred), succ=i32(m.kf_succ), cam=i32(m.kf_cam), extr=f64(m.cam_extr), intr=f64(m.cam_intr), dist=f64(m.cam_dist),
            ctype=i32(m.cam_dist_type), cimu=f64(m.cam_imu_calib), iptr=np.ascontiguousarray(m.imu_ptr, dtype=np.int64), isamp=f64(m.imu_samples), ifirst=f64(m.imu_first),
            lm=f64(m.lm_pos), lminv=u8(m.lm_invalid), lmref=i32(m.lm_ref_kf), optr=i32(m.lm_obs_ptr), okf=i32(m.obs_kf),
            ouv=np.ascontiguousarray(m.obs_uv, dtype=np.float32), ooct=i32(m.obs_octave),
            l1=i32([l.kf1 for l in m.loops]), l2=i32([l.kf2 for l in m.loops]),
            lT=f64(np.array([l.T_s1_s2 for l in m.loops]).reshape(NL, 7)), lC=f64(np.array([l.cov for l in m.loops]).reshape(NL, 36)))
        k = self._keep
        self.h = C.c_void_p(lib().shim_build(
            C.c_int(m.K), _p(k["kf_id"], C.c_int), _p(k["kf_client"], C.c_int), _p(k["inv"], C.c_ubyte), _p(k["loaded"], C.c_ubyte),
            _p(k["gba"], C.c_ubyte), _p(k["pose"]), _p(k["vio"]), _p(k["vel"]), _p(k["ba"]), _p(k["bg"]), _p(k["pred"], C.c_int),
            _p(k["succ"], C.c_int), _p(k["cam"], C.c_int), _p(k["extr"]), _p(k["intr"]), _p(k["dist"]), _p(k["ctype"], C.c_int), _p(k["cimu"]),
            _p(k["iptr"], C.c_long), _p(k["isamp"]), _p(k["ifirst"]), C.c_int(m.L), _p(k["lm"]), _p(k["lminv"], C.c_ubyte),
            _p(k["lmref"], C.c_int), _p(k["optr"], C.c_int), _p(k["okf"], C.c_int), _p(k["ouv"], C.c_float), _p(k["ooct"], C.c_int),
            C.c_int(NL), _p(k["l1"], C.c_int), _p(k["l2"], C.c_int), _p(k["lT"]), _p(k["lC"]), C.c_int(m.id_map)))

    def close(self):
        if self.h:
            lib().shim_free(self.h)
            self.h = None

    def flatten_gba(self, visual_only, round2):
        sizes = np.zeros(6, np.int32)
        ncam = C.c_int(0)
        lib().shim_flatten_gba(self.h, int(visual_only), int(round2), _p(sizes, C.c_int), *([None] * 14), C.byref(ncam))
        K, L, O, I, E, S = [int(x) for x in sizes]
        out = dict(pose=np.zeros((K, 7)), fixed=np.zeros(K, np.uint8), lm=np.zeros((L, 3)), obs_ptr=np.zeros(L + 1, np.int32),
                   obs_kf=np.zeros(O, np.int32), uv=np.zeros((O, 2)), sigma=np.zeros(O), imu_i=np.zeros(I, np.int32), imu_j=np.zeros(I, np.int32),
                   ei=np.zeros(E, np.int32), ej=np.zeros(E, np.int32), loss=np.zeros(E), noise=np.zeros((I, 5)), kf_cam=np.zeros(K, np.int32))
        lib().shim_flatten_gba(self.h, int(visual_only), int(round2), _p(sizes, C.c_int), _p(out["pose"]), _p(out["fixed"], C.c_ubyte), _p(out["lm"]),
                               _p(out["obs_ptr"], C.c_int), _p(out["obs_kf"], C.c_int), _p(out["uv"]), _p(out["sigma"]), _p(out["imu_i"], C.c_int),
                               _p(out["imu_j"], C.c_int), _p(out["ei"], C.c_int), _p(out["ej"], C.c_int), _p(out["loss"]), _p(out["noise"]), _p(out["kf_cam"], C.c_int), None)
        out["sizes"] = (K, L, O, I, E, S)
        out["num_cam"] = int(ncam.value)
        return out

    def gba(self, iterations, visual_only=False, outlier_removal=True):
        lib().shim_gba(self.h, int(iterations), int(visual_only), int(outlier_removal))

    @staticmethod
    def last_stages():
        """{stage: ms} of this thread's last GlobalBundleAdjustment call through the facade."""
        names = C.create_string_buffer(64 * 16); ms = (C.c_double * 16)()
        lib().shim_last_stages.restype = C.c_int
        n = lib().shim_last_stages(names, ms, 16)
        out = {}
        for i in range(n):
            k = names.raw[64 * i:64 * i + 64].split(b"\0")[0].decode()
            out[k] = out.get(k, 0.0) + ms[i]
        return out

    def pgo(self, corrected):
        idx = np.ascontiguousarray(list(corrected.keys()), dtype=np.int32)
        poses = np.ascontiguousarray(np.array(list(corrected.values())).reshape(-1, 7), dtype=np.float64)
        lib().shim_pgo(self.h, len(idx), _p(idx, C.c_int), _p(poses))

    def relpose(self, kf1, kf2, T12):
        """Optimization::OptimizeRelativePose(kf1, kf2, matches1, T12, th2) through the C++ facade; matches = shared landmarks."""
        T = np.ascontiguousarray(T12, dtype=np.float64).copy()
        nfeat = int((self.m.obs_kf == kf1).sum())
        removed = np.zeros(max(nfeat, 1), np.uint8)
        lib().shim_relpose.restype = C.c_int
        n = lib().shim_relpose(self.h, int(kf1), int(kf2), _p(T), _p(removed, C.c_ubyte))
        return int(n), T, removed[:nfeat].astype(bool)

    def state(self):
        m = self.m
        out = dict(pose=np.zeros((m.K, 7)), vel=np.zeros((m.K, 3)), ba=np.zeros((m.K, 3)), bg=np.zeros((m.K, 3)), gba=np.zeros(m.K, np.uint8),
                   lm=np.zeros((m.L, 3)), lm_invalid=np.zeros(m.L, np.uint8), lm_nobs=np.zeros(m.L, np.int32))
        lib().shim_get_state(self.h, _p(out["pose"]), _p(out["vel"]), _p(out["ba"]), _p(out["bg"]), _p(out["gba"], C.c_ubyte), _p(out["lm"]),
                             _p(out["lm_invalid"], C.c_ubyte), _p(out["lm_nobs"], C.c_int))
        return out
