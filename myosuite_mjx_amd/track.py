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


# --------------------------------------------------------------------------------------------- reference motion
class ReferenceMotion:
    """Batched restatement of the reference lookup (mjx/reference_motion.py:7-313 == logger/reference_motion.py): FIXED (1 row), RANDOM
    (2 rows: low / high) or TRACK (a motion of N > 2 frames).  `get_reference(time[B])` answers for a whole batch at once.

    Faithful to the reference, quirks included (they are pinned by tests/golden/ref_motion.npz, generated by running the reference):
      * times are rounded to 4 decimals before any comparison (:9, :187);
      * an exact frame time returns that frame; past the last frame with motion_extrapolation the last frame is held (:189-190);
      * BETWEEN frames the reference computes  blend = time - t[i] / (t[i+1] - t[i])  (operator precedence: not a fraction of the interval)
        and  robot = (1 - blend) ** robot[i] + blend * robot[i+1]  (a power, not a product); the object row is linear in that blend
        (:271-300).  `interpolation="linear"` replaces both by the evident intent, (1 - b) x[i] + b x[i+1] with b the interval fraction.
      * RANDOM draws uniformly between the two rows at every call (the jax port re-uses PRNGKey(0), i.e. one fixed draw; the numpy
        version uses its generator: the latter is followed, parity in distribution)."""

    def __init__(self, reference, motion_extrapolation=False, rng=None, interpolation="reference"):
        if isinstance(reference, str):
            reference = {k: v for k, v in np.load(reference).items()}           # .npz motion file (allow_pickle stays False)
        ref = {k: (None if v is None else np.asarray(v, float)) for k, v in dict(reference).items()}
        assert "time" in ref, "Missing key (time) in reference"
        for k in ("robot", "robot_vel", "object"):
            ref.setdefault(k, None)
            assert ref[k] is None or ref[k].ndim == 2
        ref["time"] = np.around(ref["time"], 4)
        rs = ref["robot"].shape if ref["robot"] is not None else (0, 0)
        osz = ref["object"].shape if ref["object"] is not None else (0, 0)
        self.robot_dim, self.object_dim = rs[1], osz[1]
        self.robot_horizon, self.object_horizon, self.horizon = rs[0], osz[0], max(rs[0], osz[0])
        if rs[0] > 2 or osz[0] > 2:
            self.type = "TRACK"
        elif rs[0] == 2 or osz[0] == 2:
            self.type = "RANDOM"
        elif rs[0] == 1 or osz[0] == 1:
            self.type = "FIXED"
        else:
            raise ValueError("Reference values not per specs")
        if ref.get("robot_init") is None:
            ref["robot_init"] = ref["robot"][0] if ref["robot"] is not None else None
        if ref.get("object_init") is None:
            ref["object_init"] = ref["object"][0] if ref["object"] is not None else None
        self.reference = ref
        self.motion_extrapolation = motion_extrapolation
        self.rng = rng or np.random.default_rng()
        self.interpolation = interpolation

    def get_init(self):
        return self.reference["robot_init"], self.reference["object_init"]

    def get_reference(self, time):
        """time: scalar or [B] -> dict(robot [B, nr], robot_vel [B, nr] | None, object [B, no]) in float64 numpy."""
        R = self.reference
        t = np.around(np.atleast_1d(np.asarray(time, float)), 4)
        B = t.shape[0]
        rows = lambda a, i: None if a is None else a[i]
        if self.type == "FIXED":
            z = np.zeros(B, int)
            return dict(robot=rows(R["robot"], z), robot_vel=rows(R["robot_vel"], z), object=rows(R["object"], z))
        if self.type == "RANDOM":
            draw = lambda a: None if a is None else self.rng.uniform(a[0], a[1], (B, a.shape[1]))
            return dict(robot=draw(R["robot"]), robot_vel=draw(R["robot_vel"]), object=draw(R["object"]))
        T = R["time"]
        if not self.motion_extrapolation:
            assert (t <= T[-1]).all(), "Trying to access time beyond max reference duration"
        idx = np.clip(np.searchsorted(T, t, side="right") - 1, 0, self.horizon - 1)
        held = t >= T[-1]
        idx = np.where(held, self.horizon - 1, idx)
        exact = held | (T[idx] == t)
        nxt = np.minimum(idx + 1, self.horizon - 1)
        dt = np.where(exact, 1.0, T[nxt] - T[idx])
        if self.interpolation == "linear":
            blend = (t - T[idx]) / dt
            mixr = lambda a: (1.0 - blend)[:, None] * a[idx] + blend[:, None] * a[nxt]
            mixo = mixr
        else:
            blend = t - T[idx] / dt
            mixr = lambda a: (1.0 - blend)[:, None] ** a[idx] + blend[:, None] * a[nxt]
            mixo = lambda a: (1.0 - blend)[:, None] * a[idx] + blend[:, None] * a[nxt]

        def pick(a, mix, horizon):
            if a is None:
                return None
            if horizon <= 1:
                return a[np.zeros(B, int)]
            return np.where(exact[:, None], a[idx], mix(a))
        with np.errstate(invalid="ignore", over="ignore", divide="ignore"):      # (a zero base at an exact frame: that row is not used)
            return dict(robot=pick(R["robot"], mixr, self.robot_horizon), robot_vel=pick(R["robot_vel"], mixr, self.robot_horizon),
                        object=pick(R["object"], mixo, self.object_horizon))

    def get_reference_torch(self, time, generator=None):
        """The same lookup with torch tensors on `time`'s device (float64 inside, like the numpy path): no host round trip per env step.
        time: [B] tensor.  RANDOM draws come from `generator` (a torch.Generator on that device)."""
        import torch
        dev = time.device
        if not hasattr(self, "_tt") or self._tt["time"].device != dev:
            self._tt = {k: (None if v is None else torch.as_tensor(np.asarray(v, float), dtype=torch.float64, device=dev))
                        for k, v in self.reference.items() if k in ("time", "robot", "robot_vel", "object")}
        R = self._tt
        t = torch.round(time.to(torch.float64) * 1e4) / 1e4
        B = t.shape[0]
        if self.type == "FIXED":
            return {k: (None if R[k] is None else R[k][:1].expand(B, -1)) for k in ("robot", "robot_vel", "object")}
        if self.type == "RANDOM":
            draw = lambda a: None if a is None else a[0] + (a[1] - a[0]) * torch.rand((B, a.shape[1]), dtype=torch.float64, device=dev, generator=generator)
            return dict(robot=draw(R["robot"]), robot_vel=draw(R["robot_vel"]), object=draw(R["object"]))
        T = R["time"]
        idx = torch.clamp(torch.searchsorted(T, t, right=True) - 1, 0, self.horizon - 1)
        held = t >= T[-1]
        idx = torch.where(held, torch.full_like(idx, self.horizon - 1), idx)
        exact = held | (T[idx] == t)
        nxt = torch.clamp(idx + 1, max=self.horizon - 1)
        dt = torch.where(exact, torch.ones_like(t), T[nxt] - T[idx])
        if self.interpolation == "linear":
            blend = ((t - T[idx]) / dt)[:, None]
            mixr = mixo = lambda a: (1.0 - blend) * a[idx] + blend * a[nxt]
        else:
            blend = (t - T[idx] / dt)[:, None]
            mixr = lambda a: torch.pow(1.0 - blend, a[idx]) + blend * a[nxt]
            mixo = lambda a: (1.0 - blend) * a[idx] + blend * a[nxt]

        def pick(a, mix, horizon):
            if a is None:
                return None
            if horizon <= 1:
                return a[:1].expand(B, -1)
            return torch.where(exact[:, None], a[idx], mix(a))
        return dict(robot=pick(R["robot"], mixr, self.robot_horizon), robot_vel=pick(R["robot_vel"], mixr, self.robot_horizon),
                    object=pick(R["object"], mixo, self.object_horizon))


# --------------------------------------------------------------------------------------------- the env
MYODM_DEFAULT_REFERENCE = dict(          # mjx/myodm_v0.py:306-318 (two rows => RANDOM type)
    time=np.array([0.0, 4.0]), robot=np.zeros((2, 29)), robot_vel=np.zeros((2, 29)),
    object_init=np.array([0.0, 0.0, 0.1, 1.0, 0.0, 0.0, 0.0]),
    object=np.array([[-0.2, -0.2, 0.1, 1.0, 0.0, 0.0, -1.0], [0.2, 0.2, 0.1, 1.0, 0.0, 0.0, 1.0]]))


class TrackReward:
    """compute_reward of the reference (mjx/myodm_v0.py:185-267) for a whole batch, in torch (float32, any device).  Body frames come from
    link frames [B, 12 * nlink] (MYO_F_LINKX on the GPU; the oracle's xpos / xmat of the link head bodies in the CPU tests)."""

    DEFAULT_RWD_KEYS_AND_WEIGHTS = {"pose": 0.0, "object": 1.0, "bonus": 1.0, "penalty": -2}     # :16-21

    def __init__(self, m, object_name="airplane", device="cpu", terminate_obj_fail=True, terminate_pose_fail=False):
        import torch
        self._torch = torch
        self.TermObj, self.TermPose = terminate_obj_fail, terminate_pose_fail
        # constants of _load_reference_motion (:104-128)
        self.lift_bonus_thresh, self.obj_err_scale, self.base_err_scale, self.lift_bonus_mag = 0.02, 50, 40, 1
        self.qpos_reward_weight, self.qpos_err_scale, self.qvel_reward_weight, self.qvel_err_scale = 0.35, 5.0, 0.05, 0.1
        self.obj_fail_thresh, self.base_fail_thresh, self.qpos_fail_thresh = 0.25, 0.25, 0.75
        self.object_bid, self.wrist_bid = m.name2id("body", object_name), m.name2id("body", "lunate")
        self.lift_z = float(m.body_ipos[self.object_bid][2] + m.body_pos[self.object_bid][2]) + self.lift_bonus_thresh      # :137-139
        f32 = torch.float32
        self._bl = {}                                        # body-in-link tables for xipos / ximat of the two bodies the reward reads
        for b in (self.object_bid, self.wrist_bid):
            l = int(m.hip_body_link[b])
            R = quat2mat(np.asarray(m.hip_body_lquat[b], float))
            self._bl[b] = (l, torch.tensor(np.asarray(m.hip_body_lpos[b], float) + R @ np.asarray(m.body_ipos[b], float), dtype=f32, device=device),
                           torch.tensor(R @ quat2mat(np.asarray(m.body_iquat[b], float)), dtype=f32, device=device))

    def _body_ipose(self, linkx, b):
        l, p, R = self._bl[b]
        L = linkx[:, 12 * l: 12 * l + 12]
        Rl = L[:, 3:].reshape(-1, 3, 3)
        return L[:, :3] + Rl @ p, Rl @ R

    @staticmethod
    def _mat2quat(torch, M):
        """Rotation matrix -> unit quaternion with w >= 0 (mjx/quat_math.py:108-169, four branches by the largest diagonal term)."""
        m00, m01, m02, m10, m11, m12, m20, m21, m22 = (M[:, i, j] for i in range(3) for j in range(3))
        q = torch.empty((M.shape[0], 4), dtype=M.dtype, device=M.device)
        t4 = 1 + m00 + m11 + m22; t1 = 1 + m00 - m11 - m22; t2 = 1 - m00 + m11 - m22; t3 = 1 - m00 - m11 + m22
        c4 = (m22 >= 0) & ~(m00 < -m11); c3 = (m22 >= 0) & (m00 < -m11); c1 = (m22 < 0) & (m00 > m11); c2 = (m22 < 0) & ~(m00 > m11)
        s4 = 2 * torch.sqrt(torch.clamp(t4, min=1e-30)); s1 = 2 * torch.sqrt(torch.clamp(t1, min=1e-30))
        s2 = 2 * torch.sqrt(torch.clamp(t2, min=1e-30)); s3 = 2 * torch.sqrt(torch.clamp(t3, min=1e-30))
        w = torch.where(c4, 0.25 * s4, torch.where(c1, (m21 - m12) / s1, torch.where(c2, (m02 - m20) / s2, (m10 - m01) / s3)))
        x = torch.where(c4, (m21 - m12) / s4, torch.where(c1, 0.25 * s1, torch.where(c2, (m01 + m10) / s2, (m20 + m02) / s3)))
        y = torch.where(c4, (m02 - m20) / s4, torch.where(c1, (m01 + m10) / s1, torch.where(c2, 0.25 * s2, (m12 + m21) / s3)))
        z = torch.where(c4, (m10 - m01) / s4, torch.where(c1, (m20 + m02) / s1, torch.where(c2, (m12 + m21) / s2, 0.25 * s3)))
        q[:, 0], q[:, 1], q[:, 2], q[:, 3] = w, x, y, z
        return torch.where((q[:, :1] < 0), -q, q)

    def __call__(self, ref, qpos, qvel, linkx):
        """ref = dict of [B, .] tensors (robot, robot_vel | None, object); returns reward[B], done[B], metrics."""
        torch = self._torch
        norm2 = lambda x: torch.sum(torch.square(x), -1)
        hq, hv = qpos[:, :-6], qvel[:, :-6]
        targ_com, targ_rot = ref["object"][:, :3], ref["object"][:, 3:]
        com, Rm = self._body_ipose(linkx, self.object_bid)
        cur_rot = self._mat2quat(torch, Rm)
        obj_com_err = torch.sqrt(norm2(targ_com - com))
        # rotation_distance(curr, targ, euler=False) = |quatDiff2Vel(targ, curr, 1)[0]|  (:180-183): diff = curr * conj(targ)
        a, b = cur_rot, torch.cat((targ_rot[:, :1], -targ_rot[:, 1:]), 1)
        d = torch.stack((a[:, 0] * b[:, 0] - a[:, 1] * b[:, 1] - a[:, 2] * b[:, 2] - a[:, 3] * b[:, 3],
                         a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0] + a[:, 2] * b[:, 3] - a[:, 3] * b[:, 2],
                         a[:, 0] * b[:, 2] - a[:, 1] * b[:, 3] + a[:, 2] * b[:, 0] + a[:, 3] * b[:, 1],
                         a[:, 0] * b[:, 3] + a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1] + a[:, 3] * b[:, 0]), 1)
        obj_rot_err = torch.abs(2 * torch.atan2(torch.sqrt(norm2(d[:, 1:])), d[:, 0])) / np.pi
        obj_reward = torch.exp(-self.obj_err_scale * obj_com_err) * torch.exp(-self.obj_err_scale * obj_rot_err)
        lift_bonus = ((targ_com[:, 2] >= self.lift_z) & (com[:, 2] >= self.lift_z)).to(qpos.dtype)
        qerr = hq - ref["robot"]
        qpos_reward = torch.exp(-self.qpos_err_scale * norm2(qerr))
        if ref.get("robot_vel") is None:
            qvel_reward = torch.ones_like(qpos_reward)                 # exp(-0.1 * norm2([0])) = 1 (:208-218)
        else:
            qvel_reward = torch.exp(-self.qvel_err_scale * norm2(hv - ref["robot_vel"]))
        pose_reward, vel_reward = self.qpos_reward_weight * qpos_reward, self.qvel_reward_weight * qvel_reward
        wrist, _ = self._body_ipose(linkx, self.wrist_bid)
        base_error = torch.sqrt(norm2(com - wrist))
        base_reward = torch.exp(-self.base_err_scale * base_error)
        term = torch.zeros_like(base_error, dtype=torch.bool)
        if self.TermObj:
            # (the reference squares the already-unsquared error once more: norm2(obj_com_err) = obj_com_err ** 2, :236)
            term = term | (obj_com_err ** 2 >= self.obj_fail_thresh ** 2) | (base_error ** 2 >= self.base_fail_thresh ** 2)
        if self.TermPose:
            term = term | (norm2(qerr) >= self.qpos_fail_thresh)
        done = term.to(qpos.dtype)
        metrics = {"pose": pose_reward + vel_reward, "object": obj_reward + base_reward, "bonus": self.lift_bonus_mag * lift_bonus, "penalty": done}
        reward = sum(w * metrics[k] for k, w in self.DEFAULT_RWD_KEYS_AND_WEIGHTS.items())
        return reward, done, metrics


class TrackEnv:
    """MyoDM TrackEnv (mjx/myodm_v0.py:14-304), `num_envs` copies stepped together on one MI355X.

      reset(seed) -> obs[B, 70];   step(action[B, 45]) -> (obs, reward[B], done[B], info{metrics})
    Per env step (:269-295): ctrl = (action + 1) (hi - lo) / 2 + lo over actuator_ctrlrange, reference looked up at the time BEFORE the
    step (+ motion_start_time), n_frames = 5 physics substeps of 2 ms with the Newton solver (:37-46), obs = [qpos, qvel] (:297-304),
    reward / done from compute_reward (:185-267) on the stepped state -- whose body frames (xipos, ximat) are those of the last substep's
    position stage, as in an MJX pipeline_state (the kernel exports exactly those link frames: MYO_F_LINKX).
    Reward arithmetic runs in torch on the device (float32 like jax's default); nothing goes through the host inside step()."""

    def __init__(self, num_envs=1, object_name="airplane", reference=None, model_path=None, motion_start_time=0.0, motion_extrapolation=True,
                 terminate_obj_fail=True, terminate_pose_fail=False, n_frames=5, device=0, seed=0, autoreset=False, interpolation="reference"):
        import torch
        from . import capi, model as _model
        from .envs import _DevArray
        self._torch, self._capi, self._DevArray = torch, capi, _DevArray
        self.num_envs, self.device, self.n_frames, self.autoreset = int(num_envs), device, int(n_frames), autoreset
        self.mjmodel = m = _model.load_asset(f"myohand_object_{object_name}")
        self.model = capi.HipModel(m.blob(), device)
        self.batch = capi.HipBatch(self.model, self.num_envs)
        self.dt = m.timestep * self.n_frames
        self.ref = ReferenceMotion(MYODM_DEFAULT_REFERENCE if reference is None else reference, motion_extrapolation,
                                   np.random.default_rng(seed), interpolation)
        self.motion_start_time = float(motion_start_time)
        self.rwd = TrackReward(m, object_name, f"cuda:{device}", terminate_obj_fail, terminate_pose_fail)
        init = np.array(m.qpos0, float)                                                                                    # :144-150
        ri, oi = self.ref.get_init()
        if ri is not None:
            init[: self.ref.robot_dim] = ri
        if oi is not None:
            init[self.ref.robot_dim: self.ref.robot_dim + 3] = oi[:3]
            init[-3:] = quat2euler(oi[3:])
        self.init_qpos = init.astype(np.float32)
        dev = f"cuda:{device}"
        f32 = torch.float32
        cr = np.asarray(m.actuator_ctrlrange, float)
        self._lo, self._hi = torch.tensor(cr[:, 0], dtype=f32, device=dev), torch.tensor(cr[:, 1], dtype=f32, device=dev)
        self.obs_dim, self.act_dim = m.nq + m.nv, m.nu
        self._views = {}
        self.metrics = {}
        self._init_dev = torch.tensor(self.init_qpos, device=dev)
        self._gen = torch.Generator(device=dev).manual_seed(int(seed))

    def view(self, field):
        capi = self._capi
        if field not in self._views:
            ptr, pitch, width = self.batch.field_ptr(field)
            arr = self._DevArray(ptr, (self.num_envs, width), "<i4" if field in capi.INT_FIELDS else "<f4", self.batch)
            self._views[field] = self._torch.as_tensor(arr, device=f"cuda:{self.device}")
        return self._views[field]

    def _stream(self):
        return self._torch.cuda.current_stream(self.device).cuda_stream

    def _obs(self):
        capi = self._capi
        return self._torch.cat((self.view(capi.F_QPOS), self.view(capi.F_QVEL)), 1)

    def reset(self, seed=None, mask=None):
        """All envs (or those in `mask`) back to init_qpos, zero velocity / activation / time (:152-173; the rng argument is ignored there too)."""
        capi, torch = self._capi, self._torch
        if mask is None:
            self.view(capi.F_QPOS)[:] = self._init_dev
            for f in (capi.F_QVEL, capi.F_ACT, capi.F_CTRL, capi.F_WARMSTART, capi.F_TIME):
                self.view(f).zero_()
        else:                       # masked, without a host round trip: blend in place
            mk = mask.reshape(-1, 1)
            q = self.view(capi.F_QPOS)
            q.copy_(torch.where(mk, self._init_dev, q))
            for f in (capi.F_QVEL, capi.F_ACT, capi.F_CTRL, capi.F_WARMSTART, capi.F_TIME):
                v = self.view(f)
                v.mul_((~mk).to(v.dtype))
        return self._obs()

    def step(self, action):
        capi, torch = self._capi, self._torch
        dev = f"cuda:{self.device}"
        a = torch.as_tensor(action, dtype=torch.float32, device=dev).reshape(self.num_envs, self.act_dim)
        self.view(capi.F_CTRL)[:] = (a + 1) * (self._hi - self._lo) * 0.5 + self._lo                       # :272-275
        t0 = self.view(capi.F_TIME)[:, 0].double() + self.motion_start_time                                  # reference at the pre-step time (:278-279)
        r = self.ref.get_reference_torch(t0, self._gen)
        ref = {k: (None if v is None else v.to(torch.float32)) for k, v in r.items()}
        self.batch.step(None, capi.ACTMAP_NONE, self.n_frames, self._stream())
        obs = self._obs()
        reward, done, self.metrics = self.rwd(ref, self.view(capi.F_QPOS), self.view(capi.F_QVEL), self.view(capi.F_LINKX))
        if self.autoreset:
            obs = self.reset(mask=done > 0)
        return obs, reward, done, {"metrics": self.metrics}

    def status(self):
        return self.batch.status()
