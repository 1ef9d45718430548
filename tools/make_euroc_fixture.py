#!/usr/bin/env python3
"""Derives the small trajectory fixture covins_amd/data/euroc_mh_4hz.npz from the EuRoC ground-truth
files shipped in the reference (orb_slam3/evaluation/Ground_truth/EuRoC_left_cam/MH0x_GT.txt, 20 Hz).

Run in the build container only (needs /root/reference); the .npz is committed so that tests and bench.py
never read /root/reference at run time. Per sequence: drop the initialisation segment the COVINS docs skip
(docs/run_COVINS.md:123 -> 45/35/15/15/15 s), resample to 4 Hz (one keyframe every 0.25 s).
Output arrays per sequence s in 1..5: t_s [N] seconds from sequence start, p_s [N,3], q_s [N,4] (x,y,z,w).
"""
import os
import sys

import numpy as np

REF = "/root/reference/orb_slam3/evaluation/Ground_truth/EuRoC_left_cam"
SKIP = {1: 45.0, 2: 35.0, 3: 15.0, 4: 15.0, 5: 15.0}


def main():
    out = {}
    for s in range(1, 6):
        a = np.loadtxt(os.path.join(REF, f"MH0{s}_GT.txt"), delimiter=",", comments="#")
        t = (a[:, 0] - a[0, 0]) * 1e-9
        keep = t >= SKIP[s]
        t, a = t[keep], a[keep]
        # every 5th 20 Hz sample = 4 Hz
        idx = np.arange(0, len(t), 5)
        q_wxyz = a[idx, 4:8]
        q = np.stack([q_wxyz[:, 1], q_wxyz[:, 2], q_wxyz[:, 3], q_wxyz[:, 0]], axis=1)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        # sign-continuous quaternions
        for k in range(1, len(q)):
            if np.dot(q[k], q[k - 1]) < 0:
                q[k] = -q[k]
        out[f"t_{s}"] = (t[idx] - t[idx[0]]).astype(np.float64)
        out[f"p_{s}"] = a[idx, 1:4].astype(np.float64)
        out[f"q_{s}"] = q.astype(np.float64)
        print(f"MH0{s}: {len(idx)} keyframes, {out[f't_{s}'][-1]:.1f} s", file=sys.stderr)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "covins_amd", "data", "euroc_mh_4hz.npz")
    np.savez_compressed(dst, **out)
    print("wrote", os.path.normpath(dst), os.path.getsize(dst), "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
