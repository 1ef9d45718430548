#!/usr/bin/env python3
"""Do the EuRoC IMU recordings shipped in the reference (orb_slam3/Examples/Monocular-Inertial/EuRoC_IMU/MH0{3,4,5}.txt, 200 Hz)
describe the motion of the ground-truth files the maps are built on (orb_slam3/evaluation/Ground_truth/EuRoC_left_cam/MH0x_GT.txt)?
They do — once the file's quaternion is read as WORLD -> CAMERA (rounds 1-4 of this repository read it as camera -> world and
concluded the opposite; VERDICT r04 found the error). Evidence printed per sequence:
  * R(q) e_z, the world's up direction seen in the camera frame, is constant (a camera with y down, pitched forward);
  * specific force and angular rate predicted from a spline through the 20 Hz ground truth (body = T_w_c Tbc^-1, gravity -z)
    against the recorded ones, low-passed to 2 Hz: per-axis correlation, residual after removing the mean (= the sensor biases);
  * the same under the old (wrong) reading, for the record;
  * residual against a time offset between the two clocks (none: the minimum is at 0).
This is the check VERDICT r04 "Next 1" names: gyro correlation >= 0.99, accelerometer residual <= 0.2 m/s^2.

Container only (reads /root/reference). Output committed as profiles/r05_euroc_imu_check.txt.
"""
import sys

import numpy as np
from scipy.interpolate import CubicSpline
from scipy.spatial.transform import Rotation as R, RotationSpline

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from covins_amd import synth  # noqa: E402

REF = "/root/reference/orb_slam3"


def predict(tg, p_wc, R_wc, ti):
    R_ws = R_wc * R.from_matrix(synth.TBC[:3, :3]).inv()
    p_ws = p_wc - R_ws.apply(synth.TBC[:3, 3])
    rs, cs = RotationSpline(tg, R_ws), CubicSpline(tg, p_ws)
    return rs(ti, 1), rs(ti).inv().apply(cs(ti, 2) + np.array([0.0, 0.0, synth.GRAVITY]))


def main():
    k = np.ones(100) / 100
    lp = lambda x: np.stack([np.convolve(x[:, i], k, "same") for i in range(3)], 1)[100:-100]  # noqa: E731
    for s in (3, 4, 5):
        gt = np.loadtxt(f"{REF}/evaluation/Ground_truth/EuRoC_left_cam/MH0{s}_GT.txt", delimiter=",", comments="#")
        imu = np.loadtxt(f"{REF}/Examples/Monocular-Inertial/EuRoC_IMU/MH0{s}.txt", delimiter=",", comments="#")
        tg = (gt[:, 0] - gt[0, 0]) * 1e-9
        q = np.stack([gt[:, 5], gt[:, 6], gt[:, 7], gt[:, 4]], 1)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        Rq = R.from_quat(q)
        ti = (imu[:, 0] - gt[0, 0]) * 1e-9
        m = (ti >= 1.0) & (ti <= tg[-1] - 1.0)
        ti, w, a = ti[m], imu[m, 1:4], imu[m, 4:7]
        up_c, up_w = Rq.apply([0, 0, 1.0]), Rq.inv().apply([0, 0, 1.0])
        print(f"MH0{s}: {len(ti)} IMU samples over {tg[-1]:.1f} s; recorded specific force, mean = {np.round(a.mean(0), 2)} m/s^2 (|mean| {np.linalg.norm(a.mean(0)):.2f})")
        print(f"  R(q) e_z : mean {np.round(up_c.mean(0), 3)}, std {np.round(up_c.std(0), 3)}   |   R(q)^T e_z : mean {np.round(up_w.mean(0), 3)}, std {np.round(up_w.std(0), 3)}")
        for name, R_wc in (("R_wc = R(q)^T  (q is world -> camera: what synth.py uses from round 5 on)", Rq.inv()),
                           ("R_wc = R(q)    (the reading of rounds 1-4)", Rq)):
            ww, aa = predict(tg, gt[:, 1:4], R_wc, ti)
            cw = [np.corrcoef(lp(w)[:, i], lp(ww)[:, i])[0, 1] for i in range(3)]
            ca = [np.corrcoef(lp(a)[:, i], lp(aa)[:, i])[0, 1] for i in range(3)]
            bg, ba = (w - ww).mean(0), (a - aa).mean(0)
            rw = np.sqrt(((lp(w) - lp(ww) - bg) ** 2).mean(0))
            ra = np.sqrt(((lp(a) - lp(aa) - ba) ** 2).mean(0))
            print(f"  {name}:\n    predicted specific force, mean = {np.round(aa.mean(0), 2)}; correlation with the recording per axis: gyro {np.round(cw, 3)}, accelerometer {np.round(ca, 3)}\n"
                  f"    mean difference = bias: gyro {np.round(bg, 4)} rad/s, accelerometer {np.round(ba, 3)} m/s^2; residual after it per axis: gyro {np.round(rw, 4)} rad/s, accelerometer {np.round(ra, 3)} m/s^2")
        Rwc = Rq.inv()
        R_ws = Rwc * R.from_matrix(synth.TBC[:3, :3]).inv()
        rs = RotationSpline(tg, R_ws)
        offs = (-0.02, -0.01, -0.005, 0.0, 0.005, 0.01, 0.02)
        res = []
        for off in offs:
            ww = rs(ti + off, 1)
            res.append(np.sqrt(((lp(w) - lp(ww) - (w - ww).mean(0)) ** 2).mean()))
        print("  gyro residual against a clock offset [s]: " + ", ".join(f"{o:+.3f}: {r:.4f}" for o, r in zip(offs, res)))


if __name__ == "__main__":
    main()
