"""CPU ORACLE for row f3: key-frame policy, GOP pose chain, pose-file lines.

TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as oracle/dvo_oracle.h).  numpy restatement of

  src/SolveDVO.cpp:1970-2021   first frame: reference frame + key frame (reason 1)
  src/SolveDVO.cpp:2059-2241   every other frame with __NEW__REF_UPDATE (include/SolveDVO.h:91): align from the last
                               estimate; when (nFrame - lastRefFrame) == 5 and the n-1 frame is not already the
                               reference: re-reference on the n-1 frame, make the most recent GOP entry a key frame,
                               reset the estimate to identity, align again; push as ordinary frame
  src/GOP.cpp:138-196          global_T = key_T + key_R*cT, global_R = key_R*cR; updateMostRecentToKeyFrame
  src/GOP.cpp:103-115          Eigen::Quaternion(Matrix3) for the pose message
  src/SolveDVO.cpp:1341-1354   "qx qy qz qw tx ty tz" (default ostream precision: 6 significant digits)
  src/SolveDVO.cpp:2129-2152   the adaptive key-frame exits (Laplacian scale of the residues :1398-1481, visible ratio, < 50
                               points; constants :22-23) -- commented out in the reference, optional here

PARITY UNPINNED (no reference vectors exist for any of this; Eigen is not in the image).  The alignment itself is
delegated to a callable so that the policy can be checked with a stub and, with the C oracle, end to end.
"""
import numpy as np


def quaternion_from_matrix(R):
    """Eigen 3 QuaternionBase::operator=(MatrixBase) (Shoemake); returns (x, y, z, w)"""
    m = np.asarray(R, dtype=np.float64)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


class GOP:
    """src/GOP.cpp:118-196"""

    def __init__(self):
        self.elems = []                    # dict(frame, key, reason, R, t)
        self.key_R, self.key_t = np.eye(3), np.zeros(3)

    def _global(self, cR, cT):
        return self.key_R @ cR, self.key_t + self.key_R @ cT

    def push_ordinary(self, frame, cR, cT):
        R, t = self._global(cR, cT)
        self.elems.append(dict(frame=frame, key=False, reason=-1, R=R, t=t))

    def push_key(self, frame, reason, cR, cT):
        R, t = self._global(cR, cT)
        self.elems.append(dict(frame=frame, key=True, reason=reason, R=R, t=t))
        self.key_R, self.key_t = R, t

    def update_most_recent_to_key(self, reason):
        e = self.elems[-1]
        self.key_R, self.key_t = e["R"], e["t"]
        e["key"], e["reason"] = True, reason


def pose_line(R, t):
    """printPose to a file stream (:1348-1350): operator<< of doubles = %g with 6 significant digits"""
    q = quaternion_from_matrix(R)
    return " ".join("%g" % v for v in (q[0], q[1], q[2], q[3], t[0], t[1], t[2]))


def laplacian_b(residues):
    """processResidueHistogram (:1455-1462): mean of the residues, accumulated in float32 in the list's order"""
    r = np.asarray(residues, dtype=np.float32)
    if r.size == 0:
        return np.float32(0)
    return np.float32(np.cumsum(r, dtype=np.float32)[-1] / np.float32(r.size))


def track(n_frames, align, key_frame_every=5, adaptive=None):
    """align(ref_index, now_index, R0, t0) -> (R, t) or (R, t, info): the level schedule of :2097-2104 between two frames;
    info = dict(final_eps=..., visible_ratio=..., n=...) of the last level that ran (needed when `adaptive` is given).
    adaptive: None, or dict(laplacian_b=3.0, visible_ratio=0.8, min_points=50): the three exits of :2129-2152 (commented out in the
    reference), OR-ed with the live every-5-frames rule.  Returns (gop, lines) with one pose line per frame after the first."""
    gop = GOP()
    cR, cT = np.eye(3), np.zeros(3)
    last_ref = 0
    ref = 0
    gop.push_key(0, 1, cR, cT)                                   # :2014
    lines = []
    for n in range(1, n_frames):
        out = align(ref, n, cR, cT)                              # :2097-2104 (warm start)
        cR, cT = out[0], out[1]
        signal, reason = False, 0
        if adaptive is not None:
            info = out[2]
            if laplacian_b(info["final_eps"]) > np.float32(adaptive["laplacian_b"]):      # :2131
                signal, reason = True, 2
            if np.float32(info["visible_ratio"]) < np.float32(adaptive["visible_ratio"]):  # :2139
                signal, reason = True, 3
            if info["n"] < adaptive["min_points"]:                                          # :2146
                signal, reason = True, 4
        if (n - last_ref) == key_frame_every:                    # :2155-2160
            signal, reason = True, 5
        if signal and last_ref != n - 1:                         # :2198
            last_ref = n - 1
            ref = n - 1                                          # setPrevFrameAsRefFrame + preProcessRefFrame
            gop.update_most_recent_to_key(reason)                # :2207
            out = align(ref, n, np.eye(3), np.zeros(3))          # :2210-2227
            cR, cT = out[0], out[1]
        gop.push_ordinary(n, cR, cT)                             # :2232 / :2239
        lines.append(pose_line(gop.elems[-1]["R"], gop.elems[-1]["t"]))
    return gop, lines
