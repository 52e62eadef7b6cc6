"""MyoDM TrackEnv, batched on the HIP stepper: the env the reference runs on MJX
(/root/reference/myosuite/mjx/myodm_v0.py) -- MyoHand on a 6-dof arm base + a free-moving object with convex-hull contact meshes
(`envs/myo/assets/hand/myohand_object.xml`, OBJECT_NAME = airplane), reference-motion tracking reward.

Pieces: quaternion helpers (restating mjx/quat_math.py == utils/quat_math.py; pinned by tests/golden/ref_quat_math.npz), the
reference-motion lookup (mjx/reference_motion.py == logger/reference_motion.py; pinned by tests/golden/ref_motion.npz), body frames
from the kernel's exported link frames, and the env class."""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------------------------- quaternion helpers (w, x, y, z)
def mulQuat(qa, qb):
    qa, qb = np.asarray(qa, float), np.asarray(qb, float)
    return np.stack([qa[..., 0] * qb[..., 0] - qa[..., 1] * qb[..., 1] - qa[..., 2] * qb[..., 2] - qa[..., 3] * qb[..., 3],
                     qa[..., 0] * qb[..., 1] + qa[..., 1] * qb[..., 0] + qa[..., 2] * qb[..., 3] - qa[..., 3] * qb[..., 2],
                     qa[..., 0] * qb[..., 2] - qa[..., 1] * qb[..., 3] + qa[..., 2] * qb[..., 0] + qa[..., 3] * qb[..., 1],
                     qa[..., 0] * qb[..., 3] + qa[..., 1] * qb[..., 2] - qa[..., 2] * qb[..., 1] + qa[..., 3] * qb[..., 0]], -1)


def negQuat(q):
    q = np.asarray(q, float)
    return np.concatenate([q[..., :1], -q[..., 1:]], -1)


def quat2Vel(quat, dt=1.0):
    """(speed, axis) of the rotation `quat` over dt (quat_math.py:28-33; note the 1e-8 in the axis normalisation)."""
    quat = np.asarray(quat, float)
    axis = quat[..., 1:]
    s = np.sqrt(np.sum(axis ** 2, -1))
    return 2 * np.arctan2(s, quat[..., 0]) / dt, axis / (s[..., None] + 1e-8)


def diffQuat(q1, q2):
    return mulQuat(q2, negQuat(q1))


def quatDiff2Vel(q1, q2, dt):
    return quat2Vel(diffQuat(q1, q2), dt)


def quat2mat(quat):
    quat = np.asarray(quat, float)
    w, x, y, z = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    Nq = np.sum(quat * quat, -1)
    s = 2.0 / np.where(Nq > 0, Nq, 1.0)
    X, Y, Z = x * s, y * s, z * s
    m = np.empty(quat.shape[:-1] + (3, 3))
    m[..., 0, 0] = 1.0 - (y * Y + z * Z); m[..., 0, 1] = x * Y - w * Z; m[..., 0, 2] = x * Z + w * Y
    m[..., 1, 0] = x * Y + w * Z; m[..., 1, 1] = 1.0 - (x * X + z * Z); m[..., 1, 2] = y * Z - w * X
    m[..., 2, 0] = x * Z - w * Y; m[..., 2, 1] = y * Z + w * X; m[..., 2, 2] = 1.0 - (x * X + y * Y)
    return np.where((Nq > np.finfo(np.float64).eps)[..., None, None], m, np.eye(3))


def mat2euler(mat):
    """quat_math.py:96-115: angles (x, y, z) with mat = Rx Ry Rz convention of the reference (its euler2mat inverse)."""
    mat = np.asarray(mat, float)
    cy = np.sqrt(mat[..., 2, 2] ** 2 + mat[..., 1, 2] ** 2)
    cond = cy > np.finfo(np.float64).eps * 4.0
    e = np.empty(mat.shape[:-2] + (3,))
    e[..., 2] = np.where(cond, -np.arctan2(mat[..., 0, 1], mat[..., 0, 0]), -np.arctan2(-mat[..., 1, 0], mat[..., 1, 1]))
    e[..., 1] = -np.arctan2(-mat[..., 0, 2], cy)
    e[..., 0] = np.where(cond, -np.arctan2(mat[..., 1, 2], mat[..., 2, 2]), 0.0)
    return e


def quat2euler(quat):
    return mat2euler(quat2mat(quat))


def euler2quat(euler):
    """quat_math.py:77-93."""
    euler = np.asarray(euler, float)
    ai, aj, ak = euler[..., 2] / 2, -euler[..., 1] / 2, euler[..., 0] / 2
    si, sj, sk, ci, cj, ck = np.sin(ai), np.sin(aj), np.sin(ak), np.cos(ai), np.cos(aj), np.cos(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.stack([cj * cc + sj * ss, cj * cs - sj * sc, -(cj * ss + sj * cc), cj * sc - sj * cs], -1)


def mat2quat(mat):
    """quat_math.py:118-152: eigenvector of the symmetric K matrix with the largest eigenvalue, w >= 0."""
    mat = np.asarray(mat, float)
    Qxx, Qyx, Qzx = mat[..., 0, 0], mat[..., 0, 1], mat[..., 0, 2]
    Qxy, Qyy, Qzy = mat[..., 1, 0], mat[..., 1, 1], mat[..., 1, 2]
    Qxz, Qyz, Qzz = mat[..., 2, 0], mat[..., 2, 1], mat[..., 2, 2]
    K = np.zeros(mat.shape[:-2] + (4, 4))
    K[..., 0, 0] = Qxx - Qyy - Qzz
    K[..., 1, 0] = Qyx + Qxy; K[..., 1, 1] = Qyy - Qxx - Qzz
    K[..., 2, 0] = Qzx + Qxz; K[..., 2, 1] = Qzy + Qyz; K[..., 2, 2] = Qzz - Qxx - Qyy
    K[..., 3, 0] = Qyz - Qzy; K[..., 3, 1] = Qzx - Qxz; K[..., 3, 2] = Qxy - Qyx; K[..., 3, 3] = Qxx + Qyy + Qzz
    K /= 3.0
    q = np.empty(K.shape[:-2] + (4,))
    it = np.nditer(q[..., 0], flags=["multi_index"])
    while not it.finished:
        vals, vecs = np.linalg.eigh(K[it.multi_index])
        v = vecs[[3, 0, 1, 2], np.argmax(vals)]
        q[it.multi_index] = -v if v[0] < 0 else v
        it.iternext()
    return q


# --------------------------------------------------------------------------------------------- body frames from exported link frames
def body_frames(m, linkx):
    """World position / rotation of every body from MYO_F_LINKX ([B, 12 * nlink]) and the lowering's body-in-link tables."""
    B = linkx.shape[0]
    L = np.asarray(linkx, float).reshape(B, -1, 12)
    link = np.asarray(m.hip_body_link)
    lpos, lquat = np.asarray(m.hip_body_lpos, float), np.asarray(m.hip_body_lquat, float)
    org = np.asarray(m.hip_origin, float)
    nb = len(link)
    xpos, xmat = np.zeros((B, nb, 3)), np.zeros((B, nb, 3, 3))
    Rl = quat2mat(lquat)
    for b in range(nb):
        if link[b] < 0:
            xpos[:, b] = lpos[b] + org
            xmat[:, b] = Rl[b]
        else:
            R = L[:, link[b], 3:].reshape(B, 3, 3)
            xpos[:, b] = L[:, link[b], :3] + R @ lpos[b]
            xmat[:, b] = R @ Rl[b]
    return xpos, xmat
