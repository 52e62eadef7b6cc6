"""Lowering: compiled model (mjModel-style arrays) -> tables for the HIP kernels.

The HIP stepper does not walk MuJoCo's generic body/geom/site arrays.  At model-compile
time this pass
  * folds every jointless body into the nearest jointed ancestor ("link"): the MyoHand's
    38 bodies become 17 links carrying the 23 hinge DoF; world-welded bodies become
    constants (their sites / wrap geoms / collision geoms get world coordinates);
  * renumbers links breadth-first so that each tree level is a contiguous range (the
    kernels process one level per phase, lanes = links of that level);
  * turns each actuated / limited spatial tendon into a list of segments
    (site, [wrap geom, side site], site) with precomputed sparse moment-arm DoF lists, so
    the tendon Jacobian is never materialised densely (SURVEY.md section 7 design notes);
  * builds the collision pair table over capsule/ellipsoid geoms with the DoF list of each
    pair, pruning static plane / cylinder geoms that the moving geoms provably cannot reach.

All `hip_*` arrays are added to the model's array dict and travel in the MYOB blob.
Element layouts are documented next to the C structs in csrc/myo_kernels.hip.
"""
from __future__ import annotations

import math

import numpy as np

from .mjcf import (GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_ELLIPSOID, GEOM_HFIELD, GEOM_MESH, GEOM_PLANE, GEOM_SPHERE, JNT_FREE, JNT_HINGE, JNT_SLIDE,
                   WRAP_CYLINDER, WRAP_PULLEY, WRAP_SITE, WRAP_SPHERE, mat2quat, quat2mat, quat_mul)
from . import setconst as sc

SEG_INTS = 12   # ints per tendon segment record
PAIR_INTS = 6   # ints per collision pair record
PAIR_FLTS = 12  # floats per collision pair record
ACT_FLTS = 16   # floats per actuator record


def _rel_transforms(m):
    """For every body: (link head body id or 0 for static, R_rel 3x3, p_rel 3) expressing the
    body frame in its link's head-body frame (or in the world frame for static bodies)."""
    nb = len(m.body_parentid)
    head = np.zeros(nb, int)
    R = [np.eye(3) for _ in range(nb)]
    p = [np.zeros(3) for _ in range(nb)]
    for b in range(1, nb):
        par = m.body_parentid[b]
        Rb = quat2mat(m.body_quat[b])
        if m.body_jntnum[b] > 0:
            head[b] = b
        else:
            head[b] = head[par]
            R[b] = R[par] @ Rb
            p[b] = p[par] + R[par] @ m.body_pos[b]
    return head, R, p



def _cube_dirs():
    """Centres of the 6 x 4 x 4 direction cells of the hull start table, in the kernel's cell order (support_shape, myo_physics.h):
    cell = ((2 * axis + (d[axis] < 0)) * 4 + iu) * 4 + iv, u / v = the two other components (cyclic order) over |d[axis]|, in [-1, 1]."""
    out = []
    for axis in range(3):
        for neg in (0, 1):
            for iu in range(4):
                for iv in range(4):
                    d = np.zeros(3)
                    d[axis] = -1.0 if neg else 1.0
                    d[(axis + 1) % 3] = (iu + 0.5) / 2 - 1
                    d[(axis + 2) % 3] = (iv + 0.5) / 2 - 1
                    out.append(d / np.linalg.norm(d))
    return out

def lower(cm):
    m = cm
    A = cm.arrays
    nb = len(m.body_parentid)
    nv = len(m.dof_bodyid)
    has_free = False
    for j in range(len(m.jnt_type)):
        if m.jnt_type[j] == JNT_FREE:
            b = m.jnt_bodyid[j]
            if m.body_parentid[b] != 0 or m.body_jntnum[b] != 1 or has_free:
                raise NotImplementedError("HIP path: a free joint must be the only joint of a single root body")
            has_free = True
        elif m.jnt_type[j] not in (JNT_HINGE, JNT_SLIDE):
            raise NotImplementedError("HIP path: ball joints")
    head, Rrel, prel = _rel_transforms(m)
    heads = [b for b in range(1, nb) if m.body_jntnum[b] > 0]
    # parent link of each head body
    def parent_head(b):
        return head[m.body_parentid[b]]
    depth = {}
    for b in heads:
        ph = parent_head(b)
        depth[b] = 1 if ph == 0 else depth[ph] + 1
    order = sorted(heads, key=lambda b: (depth[b], b))
    lid = {b: i for i, b in enumerate(order)}      # head body -> link id (BFS order)
    nl = len(order)
    nlevel = max(depth.values()) if depth else 0
    level_adr = np.zeros(nlevel + 1, np.int32)
    for b in order:
        level_adr[depth[b]] += 1
    level_adr = np.concatenate([[0], np.cumsum(level_adr[1:])]).astype(np.int32)
    link_parent = np.full(nl, -1, np.int32)
    link_pos = np.zeros((nl, 3))
    link_quat = np.zeros((nl, 4))
    link_dofadr = np.zeros(nl, np.int32)
    link_dofnum = np.zeros(nl, np.int32)
    link_mass = np.zeros(nl)
    link_com = np.zeros((nl, 3))
    link_inertia = np.zeros((nl, 6))
    # static world poses of all world-welded bodies
    xpos0, xquat0, _, _ = sc.forward_kinematics(m, m.qpos0)
    body_link = np.full(nb, -1, np.int32)
    for b in range(1, nb):
        if head[b]:
            body_link[b] = lid[head[b]]
    for b in order:
        l = lid[b]
        par = m.body_parentid[b]
        Rb = quat2mat(m.body_quat[b])
        if head[par] == 0:     # parent is static: world pose of this link's (pre-joint) frame
            Rw = quat2mat(xquat0[par])
            link_pos[l] = xpos0[par] + Rw @ m.body_pos[b]
            link_quat[l] = mat2quat(Rw @ Rb)
        else:
            link_parent[l] = lid[head[par]]
            link_pos[l] = prel[par] + Rrel[par] @ m.body_pos[b]
            link_quat[l] = mat2quat(Rrel[par] @ Rb)
        link_dofadr[l] = m.body_dofadr[b]
        link_dofnum[l] = m.body_dofnum[b]
    # merged inertias
    for l, hb in enumerate(order):
        members = [b for b in range(1, nb) if head[b] == hb]
        mass = sum(m.body_mass[b] for b in members)
        com = sum(m.body_mass[b] * (prel[b] + Rrel[b] @ m.body_ipos[b]) for b in members) / mass
        I = np.zeros((3, 3))
        for b in members:
            Ri = Rrel[b] @ quat2mat(m.body_iquat[b])
            d = prel[b] + Rrel[b] @ m.body_ipos[b] - com
            I += Ri @ np.diag(m.body_inertia[b]) @ Ri.T + m.body_mass[b] * (d @ d * np.eye(3) - np.outer(d, d))
        link_mass[l], link_com[l] = mass, com
        link_inertia[l] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
    # children CSR
    child_adr = np.zeros(nl + 1, np.int32)
    childs = []
    for l in range(nl):
        child_adr[l] = len(childs)
        childs += [c for c in range(nl) if link_parent[c] == l]
    child_adr[nl] = len(childs)
    # dofs.  A free joint's 3 translational dofs behave like slides along the world axes and its 3 rotational dofs like hinges
    # about the body axes through the body origin: the kernels get type 2 / 3 for them (axis / anchor are written by the
    # kinematics stage) plus a per-link `free` flag for the places that differ (pose from qpos, RNE, quaternion integration)
    dof_link = np.array([lid[m.dof_bodyid[d]] for d in range(nv)], np.int32)
    dof_type = np.zeros(nv, np.int32)
    dof_qposadr = np.zeros(nv, np.int32)
    dof_pos = np.zeros((nv, 3))
    dof_axis = np.zeros((nv, 3))
    link_free = np.zeros(nl, np.int32)
    for j in range(len(m.jnt_type)):
        d, qa = m.jnt_dofadr[j], m.jnt_qposadr[j]
        if m.jnt_type[j] == JNT_FREE:
            dof_type[d:d + 3] = JNT_SLIDE
            dof_type[d + 3:d + 6] = JNT_HINGE
            dof_qposadr[d:d + 3] = qa + np.arange(3)
            dof_qposadr[d + 3:d + 6] = qa + 3
            link_free[lid[m.jnt_bodyid[j]]] = 1
        else:
            dof_type[d] = m.jnt_type[j]
            dof_qposadr[d] = qa
            dof_pos[d] = m.jnt_pos[j]
            dof_axis[d] = m.jnt_axis[j]
    # ancestors (dof chains) helpers
    def dof_chain(link):  # all dofs from root to this link (inclusive), as a set
        s = []
        l = link
        while l >= 0:
            s += list(range(link_dofadr[l], link_dofadr[l] + link_dofnum[l]))
            l = link_parent[l]
        return set(s)

    def dof_list(link_a, link_b):
        """(dof, sign) for the Jacobian difference J_b - J_a restricted to non-common dofs."""
        ca = dof_chain(link_a) if link_a >= 0 else set()
        cb = dof_chain(link_b) if link_b >= 0 else set()
        out = [(d, +1) for d in sorted(cb - ca)] + [(d, -1) for d in sorted(ca - cb)]
        return out

    # ---- sites
    ns = len(m.site_bodyid)
    site_link = np.full(ns, -1, np.int32)
    site_lpos = np.zeros((ns, 3))
    for s in range(ns):
        b = m.site_bodyid[s]
        if b == 0 or head[b] == 0:
            site_lpos[s] = xpos0[b] + quat2mat(xquat0[b]) @ m.site_pos[s]
        else:
            site_link[s] = lid[head[b]]
            site_lpos[s] = prel[b] + Rrel[b] @ m.site_pos[s]
    # ---- geoms in link frames
    ng = len(m.geom_type)
    geom_link = np.full(ng, -1, np.int32)
    geom_lpos = np.zeros((ng, 3))
    geom_lmat = np.zeros((ng, 9))
    for g in range(ng):
        b = m.geom_bodyid[g]
        Rg = quat2mat(m.geom_quat[g])
        if b == 0 or head[b] == 0:
            Rw = quat2mat(xquat0[b])
            geom_lpos[g] = xpos0[b] + Rw @ m.geom_pos[g]
            geom_lmat[g] = (Rw @ Rg).ravel()
        else:
            geom_link[g] = lid[head[b]]
            geom_lpos[g] = prel[b] + Rrel[b] @ m.geom_pos[g]
            geom_lmat[g] = (Rrel[b] @ Rg).ravel()
    # ---- tendons driven by actuators (gt index == actuator index), then limited-only tendons
    nu = len(m.actuator_trnid)
    gt_tendon = []
    kind = np.asarray(m.arrays.get("actuator_kind", np.zeros(nu, np.int32)), np.int32)
    for i in range(nu):
        # a joint-transmission actuator becomes a pseudo tendon (-1): no segments, a constant unit moment arm on its dof
        gt_tendon.append(int(m.actuator_trnid[i]) if m.actuator_trntype[i] == 1 else -1)
        if m.actuator_trntype[i] != 1 and (kind[i] == 0 or m.jnt_type[m.actuator_trnid[i]] not in (JNT_HINGE, JNT_SLIDE)):
            raise NotImplementedError("HIP path: joint transmission is for stateless actuators on hinge / slide joints")
    real_t = [t for t in gt_tendon if t >= 0]
    if len(set(real_t)) != len(real_t):
        raise NotImplementedError("HIP path: two actuators on one tendon")
    for t in range(len(m.tendon_adr)):
        if t not in gt_tendon and (m.tendon_limited[t] or m.tendon_stiffness[t] or m.tendon_damping[t]):
            gt_tendon.append(t)
    segs, seg_div, dls = [], [], []
    gt_seg_adr, gt_seg_num, gt_dofs = [], [], []
    gt_len0 = []    # per tendon: summed length of its constant (same-link) straight segments
    wg_ids = {}     # geom id -> wrap geom index

    def wrap_index(g):
        if g not in wg_ids:
            wg_ids[g] = len(wg_ids)
        return wg_ids[g]

    gt_j0 = []      # per tendon: constant moment arms (joint transmission), by slot of its Jacobian row
    for ti, t in enumerate(gt_tendon):
        gt_seg_adr.append(len(segs))
        gt_len0.append(0.0)
        if t < 0:
            gt_seg_num.append(0)
            gt_dofs.append([int(m.jnt_dofadr[m.actuator_trnid[ti]])])
            gt_j0.append([1.0])
            continue
        gt_j0.append([])
        adr, num = m.tendon_adr[t], m.tendon_num[t]
        row = []       # dofs of this tendon's sparse Jacobian row

        def add_list(la, lb):
            lst = dof_list(la, lb)
            a0 = len(dls)
            for d, sgn in lst:
                if d not in row:
                    row.append(d)
                dls.append((d, sgn, row.index(d)))
            return a0, len(lst)

        div = 1.0
        j = 0
        while j < num - 1:
            t0, t1 = m.wrap_type[adr + j], m.wrap_type[adr + j + 1]
            if t0 == WRAP_PULLEY or t1 == WRAP_PULLEY:
                if t0 == WRAP_PULLEY:
                    div = float(m.wrap_prm[adr + j])
                j += 1
                continue
            s0 = int(m.wrap_objid[adr + j])
            if t1 in (WRAP_SPHERE, WRAP_CYLINDER):
                g = int(m.wrap_objid[adr + j + 1])
                s1 = int(m.wrap_objid[adr + j + 2])
                side = int(round(m.wrap_prm[adr + j + 1]))
                d_adr, d_n = add_list(site_link[s0], site_link[s1])
                a_adr, a_n = add_list(site_link[s0], geom_link[g])
                b_adr, b_n = add_list(geom_link[g], site_link[s1])
                segs.append([s0, s1, wrap_index(g), side, d_adr, d_n, a_adr, a_n, b_adr, b_n,
                             1 if t1 == WRAP_CYLINDER else 0, 0])
                j += 2
            else:
                s1 = int(m.wrap_objid[adr + j + 1])
                j += 1
                if site_link[s0] == site_link[s1]:
                    # both sites ride on the same link (or are both world-fixed): constant length, no moment arm.  Folded into a
                    # per-tendon offset (97 of MyoHand's 150 straight segments are of this kind)
                    gt_len0[-1] += float(np.linalg.norm(site_lpos[s1] - site_lpos[s0])) / div
                    continue
                d_adr, d_n = add_list(site_link[s0], site_link[s1])
                segs.append([s0, s1, -1, -1, d_adr, d_n, 0, 0, 0, 0, 0, 0])
            seg_div.append(div)
        gt_seg_num.append(len(segs) - gt_seg_adr[-1])
        gt_dofs.append(row)
    # per-tendon contiguous range of moment-arm entries, and a segment order with the wrapping segments first
    # (the wave-per-env kernel runs lane = segment: one full round of wraps, then straight segments only)
    gt_dl_adr, gt_dl_num = [], []
    for i in range(len(gt_tendon)):
        sa, sn = gt_seg_adr[i], gt_seg_num[i]
        es = [x for sg in segs[sa:sa + sn] for x in ((sg[4], sg[5]), (sg[6], sg[7]), (sg[8], sg[9])) if x[1] > 0]
        lo_e = min([a for a, n in es], default=0)
        hi_e = max([a + n for a, n in es], default=0)
        gt_dl_adr.append(lo_e)
        gt_dl_num.append(hi_e - lo_e)
    seg_tendon = []
    for i in range(len(gt_tendon)):
        seg_tendon += [i] * gt_seg_num[i]
    seg_order = [k for k in range(len(segs)) if segs[k][2] >= 0] + [k for k in range(len(segs)) if segs[k][2] < 0]
    nwrapseg = sum(1 for sg in segs if sg[2] >= 0)
    maxnnz = max([len(r) for r in gt_dofs] + [1])
    ngt = len(gt_tendon)
    gt_dof_tab = np.full((ngt, maxnnz), -1, np.int32)
    gt_j0_tab = np.zeros((ngt, maxnnz))
    for i, r in enumerate(gt_dofs):
        gt_dof_tab[i, :len(r)] = r
        gt_j0_tab[i, :len(gt_j0[i])] = gt_j0[i]
    # column CSR (per dof: which (tendon, slot) touch it), actuated tendons only
    col_adr = np.zeros(nv + 1, np.int32)
    cols = []
    for d in range(nv):
        col_adr[d] = len(cols)
        for i in range(nu):
            if d in gt_dofs[i]:
                cols.append((i, gt_dofs[i].index(d)))
    col_adr[nv] = len(cols)
    wgs = sorted(wg_ids, key=lambda g: wg_ids[g])
    # ---- actuator records
    act = np.zeros((nu, ACT_FLTS))
    for i in range(nu):
        gp, bp = m.actuator_gainprm[i], m.actuator_biasprm[i]
        if kind[i] == 1:
            # stateless affine actuator: force = gp[0] * clip(ctrl) + bp[0] + bp[1] * length + bp[2] * velocity.  Record: slots 0-3
            # the four coefficients, slot 10 (activation time constant of a muscle) negative as the type mark
            if m.actuator_forcelimited[i]:
                raise NotImplementedError("HIP path: forcelimited actuators")
            cr = m.actuator_ctrlrange[i] if m.actuator_ctrllimited[i] else (-1e30, 1e30)
            act[i, :4] = [gp[0], bp[0], bp[1], bp[2]]
            act[i, 10], act[i, 11] = -1.0, 1.0
            cr_raw = m.actuator_ctrlrange[i]
            act[i, 5:7] = [1.0, 0.0] if (kind == 0).any() else [(cr_raw[1] - cr_raw[0]) / 2.0, (cr_raw[1] + cr_raw[0]) / 2.0]
            act[i, 12:15] = [cr[0], cr[1], m.actuator_gear[i]]
            continue
        if not np.allclose(np.delete(gp, 2), np.delete(bp, 2)):
            raise NotImplementedError("HIP path: muscle gainprm != biasprm (other than the peak force)")
        force = gp[2] if gp[2] >= 0 else gp[3] / max(1e-15, m.actuator_acc0[i])
        bforce = bp[2] if bp[2] >= 0 else bp[3] / max(1e-15, m.actuator_acc0[i])     # sarcopenia halves the gain's force only
        lr = m.actuator_lengthrange[i]
        cr = m.actuator_ctrlrange[i] if m.actuator_ctrllimited[i] else (-1e30, 1e30)
        if m.actuator_forcelimited[i]:
            raise NotImplementedError("HIP path: forcelimited muscles")
        if m.actuator_dynprm[i, 2] != 0:
            raise NotImplementedError("HIP path: muscle tausmooth")
        act[i] = [gp[0], gp[1], force, gp[4], gp[5], gp[6], gp[7], gp[8], lr[0], lr[1],
                  m.actuator_dynprm[i, 0], m.actuator_dynprm[i, 1], cr[0], cr[1], m.actuator_gear[i], bforce]
    # ---- collision geoms + pair table
    # reach bound: every point of link l stays within reach[l] + |p - anchor_l| of the (static) anchor of its root link,
    # where anchor_l is the link's first joint position; distances between consecutive anchors are pose invariant
    def first_anchor(l):
        return dof_pos[link_dofadr[l]]

    def link_extra(l):   # several joints with different anchors on one link, and slide travel
        a0 = first_anchor(l)
        ex = 0.0
        for d in range(link_dofadr[l], link_dofadr[l] + link_dofnum[l]):
            ex += 2 * np.linalg.norm(dof_pos[d] - a0)
            if dof_type[d] == JNT_SLIDE:
                j = m.dof_jntid[d]
                ex += max(abs(m.jnt_range[j, 0]), abs(m.jnt_range[j, 1])) if m.jnt_limited[j] else 1e9
        return ex

    link_mat0 = [quat2mat(q) for q in link_quat]
    reach = np.zeros(nl)
    root_anchor = {}
    for l in range(nl):
        par = link_parent[l]
        if par < 0:
            reach[l] = np.inf if link_free[l] else link_extra(l)
            root_anchor[l] = link_pos[l] + link_mat0[l] @ first_anchor(l)
        else:
            a_in_parent = link_pos[l] + link_mat0[l] @ first_anchor(l)
            reach[l] = reach[par] + np.linalg.norm(a_in_parent - first_anchor(par)) + link_extra(l)
    cg_ids = {}
    pairs_i, pairs_f, pair_dl = [], [], []

    def cg_index(g):
        if g not in cg_ids:
            cg_ids[g] = len(cg_ids)
        return cg_ids[g]

    def geom_reach(g):   # (centre, radius) of a world sphere containing geom g in every pose
        l = geom_link[g]
        root = l
        while link_parent[root] >= 0:
            root = link_parent[root]
        return root_anchor[root], reach[l] + np.linalg.norm(geom_lpos[g] - first_anchor(l)) + m.geom_rbound[g]

    pruned = 0
    for g1, g2 in m.pair_geom:
        t1, t2 = m.geom_type[g1], m.geom_type[g2]
        margin = max(m.geom_margin[g1], m.geom_margin[g2])
        if geom_link[g1] < 0 and geom_link[g2] < 0:
            continue
        stat, mov = (g1, g2) if geom_link[g1] < 0 else ((g2, g1) if geom_link[g2] < 0 else (None, None))
        if stat is not None and m.geom_type[stat] in (GEOM_PLANE, GEOM_CYLINDER):
            c, r = geom_reach(mov)
            R = geom_lmat[stat].reshape(3, 3)
            axis = R[:, 2]
            top = geom_lpos[stat] + (axis * m.geom_size[stat, 1] if m.geom_type[stat] == GEOM_CYLINDER else 0)
            if np.isfinite(r) and (c - top) @ axis - r > margin:
                pruned += 1
                continue
            # not provably out of reach (e.g. a free object over the scene's floor / pedestal): a plane goes to the analytic plane
            # narrow phases, a static cylinder to the generic convex one (the kernel's geom frames accept world-fixed geoms)
            plane_ok = m.geom_type[stat] == GEOM_PLANE and stat == g1 and m.geom_type[mov] in (GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_MESH)
            cyl_ok = m.geom_type[stat] == GEOM_CYLINDER and m.geom_type[mov] in (GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_SPHERE, GEOM_CYLINDER)
            if not (plane_ok or cyl_ok):
                raise NotImplementedError(f"HIP path: cannot prune static geom {stat} against moving geom {mov}")
        has_hull = "mesh_vert" in A and len(A["mesh_vert"]) > 0
        ok = (GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_SPHERE, GEOM_CYLINDER) + ((GEOM_BOX, GEOM_MESH) if has_hull else ())
        plane_pair = t1 == GEOM_PLANE
        hfield_pair = t1 == GEOM_HFIELD and t2 in ok and geom_link[g1] < 0      # world-fixed height field first (the compiler orders it so)
        if not plane_pair and not hfield_pair and (t1 not in ok or t2 not in ok):
            raise NotImplementedError(f"HIP path: geom pair types {t1},{t2}")
        lst = dof_list(geom_link[g1], geom_link[g2])
        # contact parameter mixing (mj_contactParam), equal priorities
        if m.geom_priority[g1] != m.geom_priority[g2]:
            gsel = g1 if m.geom_priority[g1] > m.geom_priority[g2] else g2
            solref, solimp, fric, condim = m.geom_solref[gsel], m.geom_solimp[gsel], m.geom_friction[gsel], m.geom_condim[gsel]
        else:
            s1, s2 = m.geom_solmix[g1], m.geom_solmix[g2]
            mix = s1 / (s1 + s2) if (s1 >= 1e-15 and s2 >= 1e-15) else (0.5 if (s1 < 1e-15 and s2 < 1e-15) else (0.0 if s1 < 1e-15 else 1.0))
            r1, r2 = m.geom_solref[g1], m.geom_solref[g2]
            solref = mix * r1 + (1 - mix) * r2 if (r1[0] > 0 and r2[0] > 0) else np.minimum(r1, r2)
            solimp = mix * m.geom_solimp[g1] + (1 - mix) * m.geom_solimp[g2]
            fric = np.maximum(m.geom_friction[g1], m.geom_friction[g2])
            condim = max(m.geom_condim[g1], m.geom_condim[g2])
        pidx = [k for k in range(len(m.pair_geom)) if m.pair_geom[k, 0] == g1 and m.pair_geom[k, 1] == g2][-1]
        if "pair_condim" in A and A["pair_condim"][pidx] > 0:
            condim = int(A["pair_condim"][pidx])
        if condim not in (1, 3, 4):
            raise NotImplementedError("HIP path: condim must be 1, 3 or 4")
        if plane_pair and t2 in (GEOM_BOX, GEOM_SPHERE, GEOM_CYLINDER):
            raise NotImplementedError("HIP path: plane against a moving box / sphere / cylinder")
        b1, b2 = m.geom_bodyid[g1], m.geom_bodyid[g2]
        invw = m.body_invweight0[b1, 0] + m.body_invweight0[b2, 0]
        # narrow-phase type: 1 capsule-capsule (analytic), 2 plane-capsule, 3 plane-ellipsoid, 0 generic convex (MPR)
        # 5 plane - convex hull (deepest vertex)
        ptype = 1 if (t1 == GEOM_CAPSULE and t2 == GEOM_CAPSULE) else (2 if (plane_pair and t2 == GEOM_CAPSULE) else (
            5 if (plane_pair and t2 == GEOM_MESH) else (3 if plane_pair else (4 if hfield_pair else 0))))
        pairs_i.append([cg_index(g1), cg_index(g2), len(pair_dl), len(lst), ptype, condim])
        pairs_f.append([margin, max(m.geom_gap[g1], m.geom_gap[g2]), fric[0], invw, solref[0], solref[1],
                        solimp[0], solimp[1], solimp[2], solimp[3], solimp[4], fric[1] if condim >= 4 else 0.0])   # last: torsional coefficient
        pair_dl += lst
    cgs = sorted(cg_ids, key=lambda g: cg_ids[g])
    maxkc = max([p[3] for p in pairs_i] + [1])
    # ---- joint limits
    jl = np.zeros((nv, 12))
    for d in range(nv):
        j = m.dof_jntid[d]
        jl[d] = [m.jnt_limited[j], m.jnt_range[j, 0], m.jnt_range[j, 1], m.jnt_margin[j], m.jnt_solref[j, 0],
                 m.jnt_solref[j, 1], *m.jnt_solimp[j], m.dof_invweight0[d]]
    tl = np.zeros((ngt, 12))
    for i, t in enumerate(gt_tendon):
        if t < 0:
            continue
        tl[i] = [m.tendon_limited[t], m.tendon_range[t, 0], m.tendon_range[t, 1], m.tendon_margin[t],
                 m.tendon_solref[t, 0], m.tendon_solref[t, 1], *m.tendon_solimp[t], m.tendon_invweight0[t]]
        if m.tendon_stiffness[t] or m.tendon_damping[t]:
            raise NotImplementedError("HIP path: tendon spring/damper")
    # ---- reference point for spatial quantities: COM of the moving bodies at qpos0
    mv = [b for b in range(1, nb) if head[b]]
    c0 = sum(m.body_mass[b] * (xpos0[b] + quat2mat(xquat0[b]) @ m.body_ipos[b]) for b in mv) / sum(m.body_mass[b] for b in mv)
    # shift the world origin to c0: float coordinates stay < ~0.3 m instead of ~1.4 m (free precision);
    # everything is translation invariant, outputs that are world positions add hip_origin back
    if has_free:
        c0 = np.zeros(3)       # free-floating model: no origin shift; the spatial reference point follows the root link
    for l in range(nl):
        if link_parent[l] < 0 and not link_free[l]:
            link_pos[l] = link_pos[l] - c0
    site_lpos[site_link < 0] -= c0
    geom_lpos[geom_link < 0] -= c0
    A["hip_origin"] = np.asarray(c0).copy()
    c0 = np.zeros(3)
    # equality rows (joint couplings): ints [dof1, dof2, qadr1, qadr2], floats [a0..a4, q0_1, q0_2, solref0, solref1, solimp0..4, invweight]
    neq = int(A["sizes"][12])
    eq_i = np.zeros((neq, 4), np.int32)
    eq_f = np.zeros((neq, 16))
    for e in range(neq):
        j1, j2 = int(m.eq_obj1id[e]), int(m.eq_obj2id[e])
        if j2 < 0 or m.jnt_type[j1] == JNT_FREE or m.jnt_type[j2] == JNT_FREE:
            raise NotImplementedError("HIP path: equality must couple two scalar joints")
        d1, d2 = int(m.jnt_dofadr[j1]), int(m.jnt_dofadr[j2])
        eq_i[e] = [d1, d2, m.jnt_qposadr[j1], m.jnt_qposadr[j2]]
        eq_f[e] = [*m.eq_data[e], m.qpos0[m.jnt_qposadr[j1]], m.qpos0[m.jnt_qposadr[j2]], *m.eq_solref[e], *m.eq_solimp[e],
                   m.dof_invweight0[d1] + m.dof_invweight0[d2], 0.0]
    A["hip_eq_i"], A["hip_eq_f"] = eq_i, eq_f
    # every body's pose inside its link frame (walk-task observations read body positions / orientations): the quaternion is the
    # product of the body_quat chain from the link's head body, as mj_kinematics accumulates it (no sign canonicalisation)
    body_lquat = np.zeros((nb, 4))
    body_lquat[:, 0] = 1.0
    for b in range(1, nb):
        if m.body_jntnum[b] == 0 and head[b]:
            body_lquat[b] = quat_mul(body_lquat[m.body_parentid[b]], m.body_quat[b])
    A["hip_body_link"] = body_link
    A["hip_body_lpos"] = np.array([prel[b] if head[b] else (xpos0[b] - A["hip_origin"]) for b in range(nb)])
    A["hip_body_lquat"] = np.array([body_lquat[b] if head[b] else xquat0[b] for b in range(nb)])
    # total mass and the (constant) mass-weighted COM of the world-welded bodies, for whole-model COM observations
    xipos0 = [xpos0[b] + quat2mat(xquat0[b]) @ m.body_ipos[b] for b in range(nb)]
    st = [b for b in range(nb) if not head[b]]
    A["hip_mass"] = np.array([float(np.sum(m.body_mass)), *(sum((m.body_mass[b] * (xipos0[b] - A["hip_origin"]) for b in st), np.zeros(3)))])
    A["hip_dof_qposadr"] = dof_qposadr
    A["hip_link_free"] = link_free
    # per link: the dofs of its whole ancestor chain, root first (the wave kernel's velocity / acceleration pass accumulates them in
    # one sweep per lane instead of level by level).  entry = dof | index-within-a-free-joint << 8 | is-free-joint << 12
    chain_adr = np.zeros(nl + 1, np.int32)
    chain = []
    for l in range(nl):
        path = []
        k = l
        while k >= 0:
            path.append(k)
            k = int(link_parent[k])
        for k in reversed(path):
            for j in range(int(link_dofnum[k])):
                chain.append(int(link_dofadr[k]) + j + ((j << 8) | (1 << 12) if link_free[k] else 0))
        chain_adr[l + 1] = len(chain)
    A["hip_link_chain_adr"] = chain_adr
    A["hip_link_chain"] = np.array(chain, np.int32)
    # kinematics of the wave kernel in two phases: (1) lane = link evaluates the link's own joint chain in its PARENT's frame (independent of
    # every other link): the columns of R_loc, the origin p_loc and, per dof, axis and anchor, 4 + 2 * dofnum vectors written to an LDS scratch
    # at kin_base[l] + 3 j; (2) level by level, lane = (link of the level, vector) maps one vector to the world with the parent's frame.
    # kin_vec entry = [link | kind << 8 | index << 16, scratch offset]: kind 0 = column `index` of the link's rotation, 1 = link origin,
    # 2 = axis of dof `index`, 3 = anchor of dof `index` (kinds 1 and 3 are points: the parent's origin is added).  Free-joint links take
    # their pose from qpos directly and have no entries.
    kin_base = np.zeros(nl, np.int32)
    o = 0
    for l in range(nl):
        kin_base[l] = o
        if not link_free[l]:
            o += 3 * (4 + 2 * int(link_dofnum[l]))
    kin_adr, kin_vec = [0], []
    for L in range(nlevel):
        for l in range(int(level_adr[L]), int(level_adr[L + 1])):
            if link_free[l]:
                continue
            j = 0
            for c in range(3):
                kin_vec.append([l | (0 << 8) | (c << 16), int(kin_base[l]) + 3 * j]); j += 1
            kin_vec.append([l | (1 << 8), int(kin_base[l]) + 3 * j]); j += 1
            for k in range(int(link_dofnum[l])):
                d = int(link_dofadr[l]) + k
                kin_vec.append([l | (2 << 8) | (d << 16), int(kin_base[l]) + 3 * j]); j += 1
                kin_vec.append([l | (3 << 8) | (d << 16), int(kin_base[l]) + 3 * j]); j += 1
        kin_adr.append(len(kin_vec))
    A["hip_kin_base"] = kin_base
    A["hip_kin_adr"] = np.array(kin_adr, np.int32)
    A["hip_kin_vec"] = np.array(kin_vec if kin_vec else [[0, 0]], np.int32)
    A["hip_kin_size"] = np.array([o, max([int(link_dofnum[l]) for l in range(nl) if not link_free[l]] + [0])], np.int32)   # scratch floats, longest non-free joint chain
    act_obs = np.full(nu, -1, np.int32)      # slot of each actuator's activation in the observation's "act" block (sim.data.act order)
    act_obs[kind == 0] = np.arange(int((kind == 0).sum()))
    # height field (terrain models): [on, nrow, ncol, collision-geom index] and [x, y half extents, z scale, base depth, position]
    hfg = int(A["hfield_dims"][2]) if "hfield_dims" in A else -1
    A["hip_hf_i"] = np.array([int(hfg >= 0), int(A["hfield_dims"][0]), int(A["hfield_dims"][1]), cgs.index(hfg) if hfg >= 0 else -1] if "hfield_dims" in A else [0, 0, 0, -1], np.int32)
    A["hip_hf_f"] = np.array([*A["hfield_size"], *(geom_lpos[hfg] if hfg >= 0 else np.zeros(3))] if "hfield_dims" in A else np.zeros(7))
    A["hip_act_obs"] = act_obs
    A["hip_gt_j0"] = gt_j0_tab
    A["hip_flags"] = np.array([int(has_free), int(A["sizes"][0]), neq, int(gt_j0_tab.any()), int((kind == 0).sum()), int((kind == 1).any())], np.int32)
    A["hip_sizes"] = np.array([nl, nlevel, nv, nu, ngt, len(segs), len(dls), maxnnz, len(wgs), len(cgs), len(pairs_i),
                               maxkc, ns, len(cols), len(childs), pruned], np.int32)
    A["hip_level_adr"] = level_adr
    A["hip_link_parent"] = link_parent
    A["hip_link_pos"] = link_pos
    A["hip_link_quat"] = link_quat
    A["hip_link_dofadr"] = link_dofadr
    A["hip_link_dofnum"] = link_dofnum
    A["hip_link_mass"] = link_mass
    A["hip_link_com"] = link_com
    A["hip_link_inertia"] = link_inertia
    A["hip_child_adr"] = child_adr
    A["hip_child"] = np.array(childs, np.int32)
    A["hip_dof_link"] = dof_link
    A["hip_dof_type"] = dof_type
    A["hip_dof_pos"] = dof_pos
    A["hip_dof_axis"] = dof_axis
    A["hip_site_link"] = site_link
    A["hip_site_lpos"] = site_lpos
    A["hip_wg_link"] = geom_link[wgs] if wgs else np.zeros(0, np.int32)
    A["hip_wg_lpos"] = geom_lpos[wgs] if wgs else np.zeros((0, 3))
    A["hip_wg_lmat"] = geom_lmat[wgs] if wgs else np.zeros((0, 9))
    A["hip_wg_radius"] = m.geom_size[wgs, 0] if wgs else np.zeros(0)
    A["hip_gt_tendon"] = np.array(gt_tendon, np.int32)
    A["hip_gt_seg_adr"] = np.array(gt_seg_adr, np.int32)
    A["hip_gt_len0"] = np.array(gt_len0)
    A["hip_gt_seg_num"] = np.array(gt_seg_num, np.int32)
    A["hip_gt_dofs"] = gt_dof_tab
    A["hip_gt_dl"] = np.stack([np.array(gt_dl_adr, np.int32), np.array(gt_dl_num, np.int32)], 1).reshape(-1, 2)
    A["hip_seg_order"] = np.array(seg_order, np.int32)
    A["hip_seg_tendon"] = np.array(seg_tendon, np.int32)
    A["hip_nwrapseg"] = np.array([nwrapseg], np.int32)
    A["hip_link_mat0"] = np.stack([quat2mat(q).ravel() for q in link_quat]) if nl else np.zeros((0, 9))
    A["hip_seg"] = np.array(segs, np.int32).reshape(-1, SEG_INTS)
    A["hip_seg_div"] = np.array(seg_div)
    A["hip_dl"] = np.array(dls, np.int32).reshape(-1, 3)
    A["hip_col_adr"] = col_adr
    A["hip_col"] = np.array(cols, np.int32).reshape(-1, 2)
    A["hip_act"] = act
    A["hip_cg_link"] = geom_link[cgs] if cgs else np.zeros(0, np.int32)
    A["hip_cg_type"] = m.geom_type[cgs] if cgs else np.zeros(0, np.int32)
    A["hip_cg_lpos"] = geom_lpos[cgs] if cgs else np.zeros((0, 3))
    A["hip_cg_lmat"] = geom_lmat[cgs] if cgs else np.zeros((0, 9))
    cg_size = np.array(m.geom_size[cgs], float) if cgs else np.zeros((0, 3))
    for k, g in enumerate(cgs):
        if m.geom_type[g] == GEOM_MESH and "geom_meshadr" in A and A["geom_meshadr"][g] >= 0:
            cg_size[k] = [float(A["geom_meshadr"][g]), float(A["geom_meshnum"][g]), 0.0]     # hull: first vertex, vertex count (exact in float32)
    A["hip_cg_size"] = cg_size
    A["hip_mesh_vert"] = np.asarray(A["mesh_vert"], float).reshape(-1, 3) if "mesh_vert" in A else np.zeros((0, 3))
    # hull vertex graphs: on a convex polytope a vertex that beats all its edge neighbours along a direction is the support point, so the
    # kernels climb the graph (from the best of six axis-extreme start vertices) instead of scanning all vertices (the airplane's visual hull
    # has 500).  nbr_adr is indexed by the GLOBAL vertex number, nbr holds mesh-local neighbour numbers, start[6 per mesh] likewise local.
    nbr_adr, nbr, starts, mesh_of = [0], [], [], {}
    if "geom_meshadr" in A and len(A["hip_mesh_vert"]):
        from scipy.spatial import ConvexHull
        V = A["hip_mesh_vert"]
        for g in cgs:
            adr, num = int(A["geom_meshadr"][g]), int(A["geom_meshnum"][g])
            if m.geom_type[g] != GEOM_MESH or adr < 0 or adr in mesh_of:
                continue
            mesh_of[adr] = len(mesh_of)
        for adr in sorted(mesh_of, key=lambda a: mesh_of[a]):
            num = int([A["geom_meshnum"][g] for g in cgs if int(A["geom_meshadr"][g]) == adr][0])
            P = V[adr:adr + num]
            hull = ConvexHull(P)
            adj = [set() for _ in range(num)]
            for tri in hull.simplices:
                for i in range(3):
                    a, b = int(tri[i]), int(tri[(i + 1) % 3])
                    adj[a].add(b); adj[b].add(a)
            assert all(adj), "every stored vertex is a hull vertex"
            assert len(nbr_adr) - 1 == adr, "meshes are stored back to back in first-use order"
            for i in range(num):
                nbr += sorted(adj[i]); nbr_adr.append(len(nbr))
            starts += [int(np.argmax(P[:, 0])), int(np.argmin(P[:, 0])), int(np.argmax(P[:, 1])), int(np.argmin(P[:, 1])), int(np.argmax(P[:, 2])), int(np.argmin(P[:, 2]))]
        for k, g in enumerate(cgs):
            if m.geom_type[g] == GEOM_MESH and int(A["geom_meshadr"][g]) in mesh_of:
                cg_size[k, 2] = float(mesh_of[int(A["geom_meshadr"][g])])
        A["hip_cg_size"] = cg_size
    A["hip_mesh_nbr_adr"] = np.array(nbr_adr, np.int32)
    A["hip_mesh_nbr"] = np.array(nbr if nbr else [0], np.int32)
    A["hip_mesh_start"] = np.array(starts if starts else [0] * 6, np.int32)
    # the same graph as self-contained records, one 16-byte load per neighbour and no index chasing: entry e of the adjacency list holds the
    # neighbour's position AND where the neighbour's own (padded) adjacency list sits: [x, y, z, float(256 * first_entry + padded_degree)] (exact in float32).
    # A climb step is then one level of independent loads; hip_mesh_startrec carries, per mesh, a 96-cell direction table of start vertices
    # (_cube_dirs) in the same form.
    rec, srec = [], []
    if len(nbr):
        V = A["hip_mesh_vert"]
        adrs = sorted(mesh_of, key=lambda a: mesh_of[a])
        nums = [int([A["geom_meshnum"][g] for g in cgs if int(A["geom_meshadr"][g]) == a][0]) for a in adrs]
        # each adjacency list is padded to a multiple of EIGHT entries (last neighbour repeated) so that the kernel loads eight records at a
        # time without a bounds test; first_rec[v] = where global vertex v's padded list starts
        pad4 = lambda n: (n + 7) // 8 * 8
        first_rec = np.concatenate([[0], np.cumsum([pad4(nbr_adr[v + 1] - nbr_adr[v]) for v in range(len(nbr_adr) - 1)])]).astype(int)
        assert first_rec[-1] < (1 << 16), "adjacency entry numbers are stored exactly in a float32"
        for a, num in zip(adrs, nums):
            word = lambda v: float(256 * first_rec[a + v] + pad4(nbr_adr[a + v + 1] - nbr_adr[a + v]))
            assert max(pad4(nbr_adr[a + v + 1] - nbr_adr[a + v]) for v in range(num)) < 256      # (the poles of the sphere meshes have ~100 neighbours)
            for v in range(num):
                lst = nbr[nbr_adr[a + v]:nbr_adr[a + v + 1]]
                lst = lst + [lst[-1]] * (pad4(len(lst)) - len(lst))
                assert len(rec) == first_rec[a + v]
                for w in lst:
                    rec.append([V[a + w, 0], V[a + w, 1], V[a + w, 2], word(w)])
            # start table of the climb: 6 cube faces x 4 x 4 cells of directions, each holding the support vertex of the cell centre
            for d in _cube_dirs():
                w = int(np.argmax(V[a:a + num] @ d))
                srec.append([V[a + w, 0], V[a + w, 1], V[a + w, 2], word(w)])
    # vertex bounding box of each mesh in the mesh frame (broad phase of the TRK kernels): centre | half sizes
    aabb = []
    if len(nbr):
        for a, num in zip(adrs, nums):
            lo, hi = V[a:a + num].min(0), V[a:a + num].max(0)
            aabb.append(np.concatenate([(lo + hi) / 2, (hi - lo) / 2]))
    A["hip_mesh_aabb"] = np.array(aabb if aabb else [[0.0] * 6], float)
    A["hip_mesh_rec"] = np.array(rec if rec else [[0.0] * 4], float)
    A["hip_mesh_startrec"] = np.array(srec if srec else [[0.0] * 4] * 96, float)
    # joint friction loss (mj_instantiateFriction): per dof [frictionloss, D = 1 / R, B] with R = (1 - d) / d * invweight at the row's
    # constant position 0 (impedance = solimp[0]) and aref = -B * qvel
    fl = np.zeros((nv, 4))
    if "dof_frictionloss" in A:
        for d in range(nv):
            f = float(A["dof_frictionloss"][d])
            if f <= 0:
                continue
            sr, si = A["dof_solref_fri"][d], A["dof_solimp_fri"][d]
            imp = min(max(si[0], 1e-4), 0.9999)
            dmax = min(max(si[1], 1e-4), 0.9999)
            R = max(1e-15, (1 - imp) / imp * m.dof_invweight0[d])
            if sr[0] <= 0:
                raise NotImplementedError("HIP path: direct solref on friction loss")
            tc = max(sr[0], 2 * float(A["opt"][0]))
            fl[d] = [f, 1.0 / R, 2.0 / max(1e-15, dmax * tc), 0.0]
    A["hip_fl"] = fl
    A["hip_trk"] = np.array([int(any(p[5] >= 4 for p in pairs_i)), int((fl[:, 0] > 0).any()),
                             int(any(m.geom_type[g] in (GEOM_BOX, GEOM_MESH) for g in cgs))], np.int32)
    A["hip_cg_rbound"] = m.geom_rbound[cgs] if cgs else np.zeros(0)
    A["hip_cg_geom"] = np.array(cgs, np.int32)
    # pair order = candidate order = lane order of the narrow phase, whose 64-lane rounds each cost their slowest lane: the pairs that go
    # through MPR (generic convex, height-field prisms) come first, so that an env with more than 64 candidates (the usual case for MyoHand:
    # ~87 per substep, ~20 of them MPR) runs ONE round with MPR lanes and then rounds of analytic pairs only, instead of paying the MPR
    # latency in every round
    # (among the MPR pairs: convex-hull pairs first, then boxes, then the smooth shapes -- a hull support is a vertex-graph climb, ~8x the
    # cost of an ellipsoid support, so the round that carries them should carry all of them)
    def _cost_class(i):
        if pairs_i[i][4] not in (0, 4):
            return 3
        ty = {int(m.geom_type[cgs[pairs_i[i][0]]]), int(m.geom_type[cgs[pairs_i[i][1]]])}
        return 0 if GEOM_MESH in ty else (1 if GEOM_BOX in ty else 2)
    order = sorted(range(len(pairs_i)), key=_cost_class)
    pairs_i, pairs_f = [pairs_i[i] for i in order], [pairs_f[i] for i in order]
    A["hip_pair_i"] = np.array(pairs_i, np.int32).reshape(-1, PAIR_INTS)
    A["hip_pair_f"] = np.array(pairs_f).reshape(-1, PAIR_FLTS)
    A["hip_pair_dl"] = np.array(pair_dl, np.int32).reshape(-1, 2)
    A["hip_jl"] = jl
    A["hip_tl"] = tl
    A["hip_c0"] = np.asarray(c0)
    return cm
