#!/usr/bin/env python3
"""Why the real EuRoC IMU recordings shipped in the reference (orb_slam3/Examples/Monocular-Inertial/EuRoC_IMU/MH0{3,4,5}.txt, 200 Hz)
are not what the synthetic maps' IMU factors are built from (SURVEY.md §8d asks for them on agents 3-5).

The maps' true motion is the ground-truth file of the same tree (orb_slam3/evaluation/Ground_truth/EuRoC_left_cam/MH0x_GT.txt),
turned into a body trajectory with Tbc of EuRoC.yaml and a world whose gravity is -z (covins_amd/synth.py). An IMU factor is only
as good as the agreement between its samples and that motion: sigma(delta p) of a 0.25 s factor is ~1e-4 m. This script measures the
agreement for the REAL samples: specific force and angular rate predicted from a C2 spline through the 4 Hz keyframe poses — under
both readings of the ground-truth file (camera pose, as the directory says; body pose, as its header says) — against the recorded
ones, low-passed to 2 Hz so that only the motion content is compared.

Container only (reads /root/reference). Output committed as profiles/r04_euroc_imu_check.txt.
"""
import sys

import numpy as np
from scipy.interpolate import CubicSpline
from scipy.spatial.transform import Rotation as R, RotationSpline

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from covins_amd import synth  # noqa: E402

REF = "/root/reference/orb_slam3"
SKIP = {3: 15.0, 4: 15.0, 5: 15.0}


def main():
    for s in (3, 4, 5):
        gt = np.loadtxt(f"{REF}/evaluation/Ground_truth/EuRoC_left_cam/MH0{s}_GT.txt", delimiter=",", comments="#")
        imu = np.loadtxt(f"{REF}/Examples/Monocular-Inertial/EuRoC_IMU/MH0{s}.txt", delimiter=",", comments="#")
        t = (gt[:, 0] - gt[0, 0]) * 1e-9
        g = gt[t >= SKIP[s]]
        tg = (g[:, 0] - g[0, 0]) * 1e-9
        q = np.stack([g[:, 5], g[:, 6], g[:, 7], g[:, 4]], 1)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        ti = (imu[:, 0] - g[0, 0]) * 1e-9
        m = (ti >= 0) & (ti <= tg[-1])
        ti, w, a = ti[m], imu[m, 1:4], imu[m, 4:7]
        k = np.ones(100) / 100
        lp = lambda x: np.stack([np.convolve(x[:, i], k, "same") for i in range(3)], 1)  # noqa: E731
        print(f"MH0{s}: {len(ti)} IMU samples over {tg[-1]:.1f} s; recorded specific force, mean = {np.round(a.mean(0), 2)} m/s^2 "
              f"(|mean| {np.linalg.norm(a.mean(0)):.2f})")
        for hyp in ("ground truth = camera pose (T_w_c; body = T_w_c Tbc^-1, as synth.py reads it)", "ground truth = body pose (header p_RS_R)"):
            Rw = R.from_quat(q)
            if hyp.startswith("ground truth = camera"):
                Rws = Rw * R.from_matrix(synth.TBC[:3, :3]).inv()
                pws = g[:, 1:4] - Rws.apply(synth.TBC[:3, 3])
            else:
                Rws, pws = Rw, g[:, 1:4]
            rs = RotationSpline(tg[::5], Rws[::5])
            cs = CubicSpline(tg[::5], pws[::5], bc_type="natural")
            ww = rs(ti, 1)
            aa = rs(ti).inv().apply(cs(ti, 2) + np.array([0.0, 0.0, synth.GRAVITY]))
            cw = [np.corrcoef(lp(w)[:, i], lp(ww)[:, i])[0, 1] for i in range(3)]
            ca = [np.corrcoef(lp(a)[:, i], lp(aa)[:, i])[0, 1] for i in range(3)]
            # best constant rotation between the two angular-rate signals (Kabsch on the low-passed gyro), and what is left after it
            H = lp(ww).T @ lp(w)
            U, _, Vt = np.linalg.svd(H)
            D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
            Rb = Vt.T @ D @ U.T
            res_w = np.sqrt(((lp(w) - lp(ww) @ Rb.T) ** 2).mean()) / np.sqrt((lp(w) ** 2).mean())
            res_a = np.sqrt(((lp(a) - lp(aa) @ Rb.T) ** 2).mean(0))
            print(f"  {hyp}:\n    predicted specific force, mean = {np.round(aa.mean(0), 2)}; per-axis correlation with the recording: gyro {np.round(cw, 2)}, "
                  f"accelerometer {np.round(ca, 2)}\n    after the best constant frame rotation between the gyro signals: relative gyro residual {res_w:.2f}, "
                  f"accelerometer residual per axis {np.round(res_a, 2)} m/s^2 (a 0.25 s factor tolerates ~{2 * 1e-4 / 0.25 ** 2:.4f} m/s^2)")


if __name__ == "__main__":
    main()
