#!/usr/bin/env python3
"""Derives the trajectory / IMU fixture covins_amd/data/euroc_mh_4hz.npz from the EuRoC files shipped in the reference:
orb_slam3/evaluation/Ground_truth/EuRoC_left_cam/MH0x_GT.txt (left-camera ground truth, 20 Hz) and, for MH03-05,
orb_slam3/Examples/Monocular-Inertial/EuRoC_IMU/MH0x.txt (the recorded 200 Hz gyro / accelerometer samples; the blobs of
MH01/02 are not in the tree).

Run in the build container only (needs /root/reference); the .npz is committed so that tests and bench.py never read
/root/reference at run time. Per sequence: drop the initialisation segment the COVINS docs skip (docs/run_COVINS.md:123 ->
45/35/15/15/15 s), keep every 5th ground-truth sample = 4 Hz (one keyframe every 0.25 s).

Frame convention of the ground-truth files (round 5: rounds 1-4 read it the other way round): columns 5-8 are a Hamilton
quaternion (w, x, y, z) whose rotation matrix maps WORLD vectors into the LEFT-CAMERA frame — R(q) e_z is constant over a whole
sequence, (0.00, -0.94, -0.33): "up" seen by a camera whose y axis points down and which is pitched forward — so the camera -> world
rotation is its conjugate. Columns 2-4 are the camera centre in the world (z up, gravity-aligned). With that reading the recorded
specific force and angular rate of MH03-05 agree with the motion (tools/euroc_imu_check.py).

Output arrays per sequence s in 1..5: t_s [N] seconds from the first keyframe, p_s [N,3] camera centre in the world,
q_s [N,4] camera -> world (x,y,z,w), t0ns_s absolute time stamp of the first keyframe.
For s in 3..5 additionally: imu_s [(N-1)*50+1, 6] = (gyro xyz [rad/s], accelerometer xyz [m/s^2]) on the 200 Hz grid, row 50 i = the
sample AT keyframe i (ground-truth and IMU stamps share one 5 ms grid); t20_s / p20_s / q20_s the 20 Hz ground truth between the first and last
keyframe (for the velocity and bias "truth" of the recorded-IMU agents: covins_amd/synth.py).
"""
import os
import sys

import numpy as np

REF = "/root/reference/orb_slam3"
SKIP = {1: 45.0, 2: 35.0, 3: 15.0, 4: 15.0, 5: 15.0}
IMU_PER_KF = 50  # 200 Hz / 4 Hz


def cam_to_world_quat(a):
    """rows of MH0x_GT.txt -> unit quaternions (x,y,z,w) of the camera -> world rotation, sign-continuous"""
    q = np.stack([-a[:, 5], -a[:, 6], -a[:, 7], a[:, 4]], axis=1)   # conjugate of (w,x,y,z) = world -> camera
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    for k in range(1, len(q)):
        if np.dot(q[k], q[k - 1]) < 0:
            q[k] = -q[k]
    return q


def main():
    out = {}
    for s in range(1, 6):
        a = np.loadtxt(os.path.join(REF, "evaluation/Ground_truth/EuRoC_left_cam", f"MH0{s}_GT.txt"), delimiter=",", comments="#")
        assert np.all(np.abs(np.diff(a[:, 0]) - 50e6) < 1e3), "ground truth is not on a gap-free 20 Hz grid"
        t = (a[:, 0] - a[0, 0]) * 1e-9
        keep = t >= SKIP[s]
        t, a = t[keep], a[keep]
        idx = np.arange(0, len(t), 5)          # every 5th 20 Hz sample = 4 Hz
        q = cam_to_world_quat(a)
        out[f"t_{s}"] = (t[idx] - t[idx[0]]).astype(np.float64)
        out[f"p_{s}"] = a[idx, 1:4].astype(np.float64)
        out[f"q_{s}"] = q[idx].astype(np.float64)
        out[f"t0ns_{s}"] = np.array(a[0, 0])
        msg = f"MH0{s}: {len(idx)} keyframes, {out[f't_{s}'][-1]:.1f} s"
        f_imu = os.path.join(REF, "Examples/Monocular-Inertial/EuRoC_IMU", f"MH0{s}.txt")
        if os.path.exists(f_imu):
            imu = np.loadtxt(f_imu, delimiter=",", comments="#")
            assert np.all(np.abs(np.diff(imu[:, 0]) - 5e6) < 1e3), "IMU is not on a gap-free 200 Hz grid"
            i0 = int(round((a[idx[0], 0] - imu[0, 0]) / 5e6))
            assert abs(imu[i0, 0] - a[idx[0], 0]) < 1e3, "ground-truth and IMU stamps do not share a grid"
            n = (len(idx) - 1) * IMU_PER_KF + 1
            assert i0 >= 0 and i0 + n <= len(imu)
            out[f"imu_{s}"] = imu[i0:i0 + n, 1:7].astype(np.float64)
            last = idx[-1] + 1
            out[f"t20_{s}"] = (t[:last] - t[0]).astype(np.float64)
            out[f"p20_{s}"] = a[:last, 1:4].astype(np.float64)
            out[f"q20_{s}"] = q[:last].astype(np.float64)
            msg += f", {n} recorded IMU samples"
        print(msg, file=sys.stderr)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "covins_amd", "data", "euroc_mh_4hz.npz")
    np.savez_compressed(dst, **out)
    print("wrote", os.path.normpath(dst), os.path.getsize(dst), "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
