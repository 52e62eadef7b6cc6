"""Compile-time derived constants (MuJoCo's `mj_setConst` / `set0`) in numpy.

Restates, for the compiled model of `mjcf.py`, the reference-configuration quantities
that MuJoCo stores in `mjModel` and that the constraint code needs every step:
`dof_invweight0`, `body_invweight0`, `tendon_invweight0`, `actuator_acc0`,
`stat.meaninertia`, and (finger model only) muscle `lengthrange`.  It is written in
plain numpy on purpose: it is independent from both the C oracle and the HIP kernels,
so `tests/` can also use it as a third opinion on kinematics, tendon wrapping and the
mass matrix.  [3P: formulas from MuJoCo's documentation, "Computation" chapter.]
"""
from __future__ import annotations

import math

import numpy as np

from .mjcf import (JNT_FREE, JNT_HINGE, JNT_SLIDE, WRAP_CYLINDER, WRAP_PULLEY, WRAP_SITE, WRAP_SPHERE,
                   mjMINVAL, quat2mat, quat_mul, quat_normalize, axisangle2quat)


def forward_kinematics(m, qpos):
    """Body frames, joint anchors/axes (world), for hinge/slide/free joints."""
    nb = len(m.body_parentid)
    xpos = np.zeros((nb, 3))
    xquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
    xanchor = np.zeros((len(m.jnt_type), 3))
    xaxis = np.zeros((len(m.jnt_type), 3))
    for b in range(1, nb):
        p = m.body_parentid[b]
        ja, jn = m.body_jntadr[b], m.body_jntnum[b]
        if jn == 1 and m.jnt_type[ja] == JNT_FREE:
            qa = m.jnt_qposadr[ja]
            pos = qpos[qa:qa + 3].copy()
            quat = quat_normalize(qpos[qa + 3:qa + 7])
            xanchor[ja] = pos
            xaxis[ja] = [0, 0, 1]
        else:
            Rp = quat2mat(xquat[p])
            pos = xpos[p] + Rp @ m.body_pos[b]
            quat = quat_mul(xquat[p], m.body_quat[b])
            for j in range(ja, ja + jn):
                R = quat2mat(quat)
                xanchor[j] = pos + R @ m.jnt_pos[j]
                xaxis[j] = R @ m.jnt_axis[j]
                q = qpos[m.jnt_qposadr[j]] - m.qpos0[m.jnt_qposadr[j]]
                if m.jnt_type[j] == JNT_SLIDE:
                    pos = pos + xaxis[j] * q
                elif m.jnt_type[j] == JNT_HINGE:
                    quat = quat_mul(axisangle2quat(xaxis[j], q), quat)  # world-frame axis: premultiply
                    pos = xanchor[j] - quat2mat(quat) @ m.jnt_pos[j]
                else:
                    raise NotImplementedError
        xpos[b] = pos
        xquat[b] = quat_normalize(quat)
    return xpos, xquat, xanchor, xaxis


def jac_point(m, xanchor, xaxis, body, point, xquat=None):
    """3 x nv translational and rotational Jacobians of a world point fixed to `body`."""
    nv = len(m.dof_bodyid)
    jp, jr = np.zeros((3, nv)), np.zeros((3, nv))
    b = body
    while b > 0:
        for j in range(m.body_jntadr[b], m.body_jntadr[b] + m.body_jntnum[b]) if m.body_jntnum[b] else []:
            d = m.jnt_dofadr[j]
            t = m.jnt_type[j]
            if t == JNT_HINGE:
                jr[:, d] = xaxis[j]
                jp[:, d] = np.cross(xaxis[j], point - xanchor[j])
            elif t == JNT_SLIDE:
                jp[:, d] = xaxis[j]
            elif t == JNT_FREE:
                jp[:, d:d + 3] = np.eye(3)
                # rotational dofs of a free joint are expressed in the body frame (MuJoCo convention)
                R = quat2mat(xquat[b])
                for k in range(3):
                    jr[:, d + 3 + k] = R[:, k]
                    jp[:, d + 3 + k] = np.cross(R[:, k], point - xanchor[j])
        b = m.body_parentid[b]
    return jp, jr


def mass_matrix(m, qpos):
    """Dense joint-space inertia via sum_b J_b^T diag(m, I_b) J_b + armature."""
    xpos, xquat, xanchor, xaxis = forward_kinematics(m, qpos)
    nv = len(m.dof_bodyid)
    M = np.diag(m.dof_armature.astype(float)).copy()
    for b in range(1, len(m.body_parentid)):
        if m.body_weldid[b] == 0:
            continue
        R = quat2mat(xquat[b])
        c = xpos[b] + R @ m.body_ipos[b]
        Ri = R @ quat2mat(m.body_iquat[b])
        I = Ri @ np.diag(m.body_inertia[b]) @ Ri.T
        jp, jr = jac_point(m, xanchor, xaxis, b, c, xquat)
        M += m.body_mass[b] * jp.T @ jp + jr.T @ I @ jr
    return M


# --------------------------------------------------------------------------- tendon wrapping (numpy twin of mju_wrap)
def _is_intersect(p1, p2, p3, p4):
    det = (p4[1] - p3[1]) * (p2[0] - p1[0]) - (p4[0] - p3[0]) * (p2[1] - p1[1])
    if abs(det) < mjMINVAL:
        return False
    a = ((p4[0] - p3[0]) * (p1[1] - p3[1]) - (p4[1] - p3[1]) * (p1[0] - p3[0])) / det
    b = ((p2[0] - p1[0]) * (p1[1] - p3[1]) - (p2[1] - p1[1]) * (p1[0] - p3[0])) / det
    return 0 <= a <= 1 and 0 <= b <= 1


def _wrap_circle(d, sd, r):
    e0, e1 = d[:2], d[2:]
    sq0, sq1, sqr = e0 @ e0, e1 @ e1, r * r
    if sq0 < sqr or sq1 < sqr or r < mjMINVAL:
        return -1.0, None
    dif = e1 - e0
    dd = dif @ dif
    if dd < mjMINVAL:
        return -1.0, None
    a = min(1.0, max(0.0, -(dif @ e0) / dd))
    tmp = e0 + a * dif
    if tmp @ tmp > sqr and (sd is None or sd @ tmp >= 0):
        return -1.0, None
    s0, s1 = math.sqrt(sq0 - sqr), math.sqrt(sq1 - sqr)
    sols, good = [], []
    for sgn in (1.0, -1.0):
        p0 = np.array([(e0[0] * sqr + sgn * r * e0[1] * s0) / sq0, (e0[1] * sqr - sgn * r * e0[0] * s0) / sq0])
        p1 = np.array([(e1[0] * sqr - sgn * r * e1[1] * s1) / sq1, (e1[1] * sqr + sgn * r * e1[0] * s1) / sq1])
        if sd is not None:
            t = p0 + p1
            n = np.linalg.norm(t)
            t = t / n if n > mjMINVAL else t
            g = t @ sd
        else:
            t = p0 - p1
            g = -(t @ t)
        if _is_intersect(e0, p0, e1, p1):
            g = -10000.0
        sols.append((p0, p1))
        good.append(g)
    p0, p1 = sols[0] if good[0] > good[1] else sols[1]
    if _is_intersect(e0, p0, e1, p1):
        return -1.0, None
    return r * math.acos(min(1.0, max(-1.0, (p0 @ p1) / sqr))), np.concatenate([p0, p1])


def _wrap_inside(d, r):
    e0, e1 = d[:2], d[2:]
    l0, l1 = np.linalg.norm(e0), np.linalg.norm(e1)
    if l0 <= r or l1 <= r or r < mjMINVAL or l0 < mjMINVAL or l1 < mjMINVAL:
        return -1.0, None
    dif = e1 - e0
    dd = dif @ dif
    if dd > mjMINVAL:
        a = -(dif @ e0) / dd
        if 0 < a < 1 and np.linalg.norm(e0 + a * dif) <= r:
            return -1.0, None
    p = 0.5 * (e0 + e1)
    n = np.linalg.norm(p)
    p = p / n * r if n > mjMINVAL else p
    pnt = np.concatenate([p, p])
    A, B = r / l0, r / l1
    cosG = (l0 * l0 + l1 * l1 - dd) / (2 * l0 * l1)
    if cosG < -1 + mjMINVAL:
        return -1.0, None
    if cosG > 1 - mjMINVAL:
        return 0.0, pnt
    G = math.acos(cosG)
    z = 1 - 1e-7
    f = math.asin(A * z) + math.asin(B * z) - 2 * math.asin(z) + G
    if f > 0:
        return 0.0, pnt
    it = 0
    while it < 20 and abs(f) > 1e-6:
        df = (A / max(mjMINVAL, math.sqrt(1 - z * z * A * A)) + B / max(mjMINVAL, math.sqrt(1 - z * z * B * B))
              - 2 / max(mjMINVAL, math.sqrt(1 - z * z)))
        if df > -mjMINVAL:
            return 0.0, pnt
        z1 = z - f / df
        if z1 > z:
            return 0.0, pnt
        z = z1
        f = math.asin(A * z) + math.asin(B * z) - 2 * math.asin(z) + G
        if f > 1e-6:
            return 0.0, pnt
        it += 1
    if it >= 20:
        return 0.0, pnt
    if e0[0] * e1[1] - e0[1] * e1[0] > 0:
        vec, ang = e0 / l0, math.asin(z) - math.asin(A * z)
    else:
        vec, ang = e1 / l1, math.asin(z) - math.asin(B * z)
    p = r * np.array([math.cos(ang) * vec[0] - math.sin(ang) * vec[1], math.sin(ang) * vec[0] + math.cos(ang) * vec[1]])
    return 0.0, np.concatenate([p, p])


def wrap(x0, x1, gpos, gmat, radius, wtype, side):
    """Returns (wlen, wpnt[2,3]); wlen < 0 means the tendon does not touch the geom."""
    p0 = gmat.T @ (x0 - gpos)
    p1 = gmat.T @ (x1 - gpos)
    if np.linalg.norm(p0) < mjMINVAL or np.linalg.norm(p1) < mjMINVAL:
        return -1.0, None
    s = gmat.T @ (side - gpos) if side is not None else None
    if wtype == WRAP_SPHERE:
        ax0 = p0 / np.linalg.norm(p0)
        nrm = np.cross(p0, p1)
        nn = np.linalg.norm(nrm)
        if nn < mjMINVAL:
            i = int(np.argmax(np.abs(ax0)))
            t = np.ones(3)
            t[i] = 0
            nrm = np.cross(ax0, t)
            nn = np.linalg.norm(nrm)
        nrm = nrm / nn
        ax1 = np.cross(nrm, ax0)
        ax1 /= np.linalg.norm(ax1)
        d = np.array([p0 @ ax0, p0 @ ax1, p1 @ ax0, p1 @ ax1])
        sd = np.array([s @ ax0, s @ ax1]) if s is not None else None
    else:
        d = np.array([p0[0], p0[1], p1[0], p1[1]])
        sd = s[:2].copy() if s is not None else None
    if sd is not None and np.linalg.norm(sd) < radius:
        wlen, pnt = _wrap_inside(d, radius)
    else:
        if sd is not None:
            n = np.linalg.norm(sd)
            sd = sd / n if n > mjMINVAL else sd
        wlen, pnt = _wrap_circle(d, sd, radius)
    if wlen < 0:
        return -1.0, None
    if wtype == WRAP_SPHERE:
        r0 = ax0 * pnt[0] + ax1 * pnt[1]
        r1 = ax0 * pnt[2] + ax1 * pnt[3]
    else:
        L0 = math.hypot(p0[0] - pnt[0], p0[1] - pnt[1])
        L1 = math.hypot(p1[0] - pnt[2], p1[1] - pnt[3])
        tot = L0 + wlen + L1
        z0 = p0[2] + (p1[2] - p0[2]) * L0 / tot
        z1 = p0[2] + (p1[2] - p0[2]) * (L0 + wlen) / tot
        r0 = np.array([pnt[0], pnt[1], z0])
        r1 = np.array([pnt[2], pnt[3], z1])
        wlen = math.sqrt(wlen * wlen + (z1 - z0) ** 2)
    return wlen, np.stack([gmat @ r0 + gpos, gmat @ r1 + gpos])


def tendons(m, qpos, want_jac=True):
    """Spatial tendon lengths and Jacobians (ntendon x nv) at qpos."""
    xpos, xquat, xanchor, xaxis = forward_kinematics(m, qpos)
    xmat = [quat2mat(q) for q in xquat]
    site_x = np.stack([xpos[b] + xmat[b] @ p for b, p in zip(m.site_bodyid, m.site_pos)]) if len(m.site_pos) else None
    nt, nv = len(m.tendon_adr), len(m.dof_bodyid)
    L = np.zeros(nt)
    J = np.zeros((nt, nv))
    for t in range(nt):
        adr, num = m.tendon_adr[t], m.tendon_num[t]
        div = 1.0
        j = 0
        while j < num - 1:
            t0, t1 = m.wrap_type[adr + j], m.wrap_type[adr + j + 1]
            if t0 == WRAP_PULLEY or t1 == WRAP_PULLEY:
                if t0 == WRAP_PULLEY:
                    div = m.wrap_prm[adr + j]
                j += 1
                continue
            id0 = m.wrap_objid[adr + j]
            pts = [(site_x[id0], m.site_bodyid[id0])]
            wlen = -1.0
            if t1 in (WRAP_SPHERE, WRAP_CYLINDER):
                g = m.wrap_objid[adr + j + 1]
                id1 = m.wrap_objid[adr + j + 2]
                sid = int(round(m.wrap_prm[adr + j + 1]))
                gb = m.geom_bodyid[g]
                gpos = xpos[gb] + xmat[gb] @ m.geom_pos[g]
                gmat = xmat[gb] @ quat2mat(m.geom_quat[g])
                wlen, wp = wrap(site_x[id0], site_x[id1], gpos, gmat, m.geom_size[g, 0], t1,
                                site_x[sid] if sid >= 0 else None)
                if wlen >= 0:
                    pts += [(wp[0], gb), (wp[1], gb)]
                j += 2
            else:
                id1 = m.wrap_objid[adr + j + 1]
                j += 1
            pts.append((site_x[id1], m.site_bodyid[id1]))
            for k in range(len(pts) - 1):
                (pa, ba), (pb, bb) = pts[k], pts[k + 1]
                if wlen >= 0 and k == 1:
                    L[t] += wlen / div
                    continue
                dif = pb - pa
                dist = np.linalg.norm(dif)
                L[t] += dist / div
                if want_jac and ba != bb and dist > mjMINVAL:
                    dif /= dist
                    ja, _ = jac_point(m, xanchor, xaxis, ba, pa, xquat)
                    jb, _ = jac_point(m, xanchor, xaxis, bb, pb, xquat)
                    J[t] += (dif @ (jb - ja)) / div
    return L, J


def set_constants(m, lengthrange_grid=5):
    """Fill stat.meaninertia, *_invweight0, actuator_acc0 and missing muscle lengthranges (in place)."""
    A = m.arrays
    nv = len(m.dof_bodyid)
    M = mass_matrix(m, m.qpos0)
    Minv = np.linalg.inv(M)
    A["opt"][9] = float(np.mean(np.diag(M)))
    diw = np.diag(Minv).copy()
    for j in range(len(m.jnt_type)):          # free joints: average over the 3 translational and the 3 rotational dofs
        if m.jnt_type[j] == JNT_FREE:
            d = m.jnt_dofadr[j]
            diw[d:d + 3] = diw[d:d + 3].mean()
            diw[d + 3:d + 6] = diw[d + 3:d + 6].mean()
    A["dof_invweight0"] = diw
    xpos, xquat, xanchor, xaxis = forward_kinematics(m, m.qpos0)
    nb = len(m.body_parentid)
    biw = np.zeros((nb, 2))
    for b in range(1, nb):
        if m.body_weldid[b] == 0:
            continue
        c = xpos[b] + quat2mat(xquat[b]) @ m.body_ipos[b]
        jp, jr = jac_point(m, xanchor, xaxis, b, c, xquat)
        biw[b, 0] = np.trace(jp @ Minv @ jp.T) / 3.0
        biw[b, 1] = np.trace(jr @ Minv @ jr.T) / 3.0
    A["body_invweight0"] = biw
    L0, J0 = tendons(m, m.qpos0)
    A["tendon_length0"] = L0
    A["tendon_invweight0"] = np.einsum("ti,ij,tj->t", J0, Minv, J0)
    nu = len(m.actuator_trnid)
    acc0 = np.zeros(nu)
    for i in range(nu):
        if m.actuator_trntype[i] == 1:
            mom = J0[m.actuator_trnid[i]] * m.actuator_gear[i]
        else:
            mom = np.zeros(nv)
            mom[m.jnt_dofadr[m.actuator_trnid[i]]] = m.actuator_gear[i]
        acc0[i] = np.linalg.norm(Minv @ mom)
    A["actuator_acc0"] = acc0
    # muscles without an explicit lengthrange (finger model): MuJoCo finds the range by a
    # damped simulation (mj_setLengthRange).  Here: min/max tendon length over a joint-range
    # grid, which that procedure converges towards.  Documented as "parity unpinned".
    missing = [i for i in range(nu) if not m.actuator_has_lengthrange[i] and m.actuator_kind[i] == 0 and m.actuator_trntype[i] == 1]
    if missing:
        lim = np.array([m.jnt_range[j] if m.jnt_limited[j] else (-math.pi, math.pi) for j in range(len(m.jnt_type))])
        if len(lim) > 6:
            raise NotImplementedError("lengthrange search is only meant for small models")
        grids = [np.linspace(lo, hi, lengthrange_grid) for lo, hi in lim]
        lo = np.full(len(m.tendon_adr), np.inf)
        hi = np.full(len(m.tendon_adr), -np.inf)
        for q in np.stack(np.meshgrid(*grids, indexing="ij"), -1).reshape(-1, len(lim)):
            qq = m.qpos0.copy()
            qq[m.jnt_qposadr] = q
            L, _ = tendons(m, qq, want_jac=False)
            lo, hi = np.minimum(lo, L), np.maximum(hi, L)
        for i in missing:
            t = m.actuator_trnid[i]
            A["actuator_lengthrange"][i] = (lo[t], hi[t])
    return m
